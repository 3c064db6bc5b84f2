#!/bin/bash
# round 6, session 3, call o (re-used for two A/Bs): k_bwd_dx_dw variants -- the owner split in the dX GEMM shadows against the rounds (osold),
# and two sets of piece registers requested a round ahead (default) against one set requested late (libharl_dl1.so): unit tests, goldens, A/B
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r06s3
mkdir -p $O
cd $R
( time timeout 900 python -m pytest tests/ -x -q -m gpu -p no:cacheprovider -k "whole_layer_backward or backward_variants or fused_update_kernels_many" ) > $O/t_dl2.txt 2>&1
tail -3 $O/t_dl2.txt
for rep in 1 2 3; do for v in hip dl1; do
for cfg in mpe:20 cheetah6:8; do c=${cfg%%:*}; n=${cfg##*:}
HARL_LIB=$v timeout 600 python bench.py --config $c --steps $n --warmup 3 --cpu-cols 0 --instr-steps 0 --no-kernel-timing --no-other-configs 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$c $v', round(d['ms_per_step'],3))"
done; done; done | sort -s -k1,1 | tee $O/ab_dl2.txt
for v in hip dl1; do
HARL_LIB=$v timeout 600 python bench.py --steps 10 --warmup 3 --cpu-cols 0 --instr-steps 2 --no-other-configs 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', round(d['ms_per_step'],3), {k:round(x['avg_ms'],4) for k,x in d['kernels'].items() if k.startswith('bwd') or k.startswith('update_fwd')})"
done | tee -a $O/ab_dl2.txt
