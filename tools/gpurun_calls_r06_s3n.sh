#!/bin/bash
# round 6, session 3, call n: workgroups of k_adam_fold: 64 (default build) against 128 (aw128) and 256 (aw256)
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r06s3
mkdir -p $O
cd $R
for v in hip aw32 aw16 hip aw32 aw16; do
HARL_LIB=$v timeout 600 python bench.py --steps 10 --warmup 3 --cpu-cols 0 --instr-steps 2 --no-other-configs 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('mpe $v', round(d['ms_per_step'],3), {k:round(x['avg_ms'],4) for k,x in d['kernels'].items() if k in ('adam_fold','reduce_partials')})"
done | tee $O/ab_aw2.txt
for rep in 1 2 3; do for v in hip aw32 aw16; do
for cfg in smac3s5z:10; do c=${cfg%%:*}; n=${cfg##*:}
HARL_LIB=$v timeout 600 python bench.py --config $c --steps $n --warmup 2 --cpu-cols 0 --instr-steps 0 --no-kernel-timing --no-other-configs 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$c $v', round(d['ms_per_step'],3))"
done
HARL_LIB=$v timeout 300 python bench.py --threads-per-gpu 512 --steps 20 --cpu-cols 0 --no-other-configs --instr-steps 0 --no-kernel-timing 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('share512 $v', round(d['ms_per_step'],3))"
done; done | sort -s -k1,1 | tee -a $O/ab_aw2.txt
