#!/bin/bash
# round 6, session 3, call d: the lighter grid barrier (one release, relaxed polls, one acquire) against the old one (libharl_oldbar.so)
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r06s3
mkdir -p $O
cd $R
( time timeout 900 python -m pytest tests/ -x -q -m gpu -p no:cacheprovider -k "adam or hatrpo_gradient or hatrpo_train or golden or sharded" ) > $O/t_barrier.txt 2>&1
tail -4 $O/t_barrier.txt
for rep in 1 2 3; do for v in hip oldbar; do
for cfg in mpe:20 smac3s5z:10 humanoid17:4; do c=${cfg%%:*}; n=${cfg##*:}
HARL_LIB=$v timeout 600 python bench.py --config $c --steps $n --warmup 3 --cpu-cols 0 --instr-steps 0 --no-kernel-timing --no-other-configs 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$c $v', round(d['ms_per_step'],3))"
done; done; done | sort -s -k1,1 | tee $O/ab_barrier.txt
for v in hip oldbar; do
HARL_LIB=$v timeout 600 python bench.py --steps 10 --warmup 3 --cpu-cols 0 --instr-steps 2 --no-other-configs 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', round(d['ms_per_step'],3), {k:round(x['avg_ms'],4) for k,x in d['kernels'].items() if k in ('adam_fold','reduce_partials')})"
done | tee -a $O/ab_barrier.txt
for n in 512; do for v in hip oldbar hip oldbar; do
HARL_LIB=$v timeout 300 python bench.py --threads-per-gpu $n --steps 20 --cpu-cols 0 --no-other-configs --instr-steps 0 --no-kernel-timing 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('share$n $v', round(d['ms_per_step'],3))"
done; done | tee -a $O/ab_barrier.txt
