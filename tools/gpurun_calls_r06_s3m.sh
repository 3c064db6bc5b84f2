#!/bin/bash
# round 6, session 3, call m: loads in flight per lane in k_reduce_partials_multi: 16 (default build) against 8 (rd8) and 32 (rd32)
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r06s3
mkdir -p $O
cd $R
for rep in 1 2 3; do for v in hip rd8 rd32; do
for cfg in mpe:20 smac3s5z:10 humanoid17:3; do c=${cfg%%:*}; n=${cfg##*:}
HARL_LIB=$v timeout 600 python bench.py --config $c --steps $n --warmup 2 --cpu-cols 0 --instr-steps 0 --no-kernel-timing --no-other-configs 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$c $v', round(d['ms_per_step'],3))"
done; done; done | sort -s -k1,1 | tee $O/ab_rd.txt
for v in hip rd8 rd32; do
HARL_LIB=$v timeout 600 python bench.py --steps 10 --warmup 3 --cpu-cols 0 --instr-steps 2 --no-other-configs 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', round(d['ms_per_step'],3), {k:round(x['avg_ms'],4) for k,x in d['kernels'].items() if k in ('adam_fold','reduce_partials')})"
done | tee -a $O/ab_rd.txt
( time timeout 600 python -m pytest tests/ -x -q -m gpu -p no:cacheprovider -k "adam or trunk_in_one or train_matches_reference_golden" ) > $O/t_rd.txt 2>&1
tail -3 $O/t_rd.txt
