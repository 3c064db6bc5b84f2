#!/bin/bash
# round 6, session 3, call l: reduce_partials with four elements per lane (HARL_REDUCE_VEC=1, default) against one (0): tests + A/B
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r06s3
mkdir -p $O
cd $R
( time timeout 900 python -m pytest tests/ -x -q -m gpu -p no:cacheprovider -k "adam or gradient or golden or trunk or sharded or multidiscrete or md_" ) > $O/t_reduce.txt 2>&1
tail -3 $O/t_reduce.txt
for rep in 1 2 3; do for v in 1 0; do
for cfg in mpe:20 smac3s5z:10 humanoid17:3 cheetah6:6; do c=${cfg%%:*}; n=${cfg##*:}
HARL_REDUCE_VEC=$v timeout 600 python bench.py --config $c --steps $n --warmup 2 --cpu-cols 0 --instr-steps 0 --no-kernel-timing --no-other-configs 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$c vec=$v', round(d['ms_per_step'],3))"
done; done; done | sort -s -k1,1 | tee $O/ab_reduce.txt
for v in 1 0; do
HARL_REDUCE_VEC=$v timeout 600 python bench.py --steps 10 --warmup 3 --cpu-cols 0 --instr-steps 2 --no-other-configs 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('vec=$v', round(d['ms_per_step'],3), {k:round(x['avg_ms'],4) for k,x in d['kernels'].items() if k in ('adam_fold','reduce_partials')})"
done | tee -a $O/ab_reduce.txt
