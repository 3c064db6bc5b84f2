#!/bin/bash
# round 6, session 3, call k: the new default (NO_PK also on gru_cell.hip + elementwise.hip) against the previous one on the feed-forward workloads
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r06s3
mkdir -p $O
cd $R
for rep in 1 2 3; do for v in hip prev; do
for cfg in mpe:20 cheetah6:8 humanoid17:3; do c=${cfg%%:*}; n=${cfg##*:}
HARL_LIB=$v timeout 600 python bench.py --config $c --steps $n --warmup 2 --cpu-cols 0 --instr-steps 0 --no-kernel-timing --no-other-configs 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$c $v', round(d['ms_per_step'],3))"
done; done; done | sort -s -k1,1 | tee $O/ab_pk4.txt
( time timeout 900 python -m pytest tests/ -x -q -m gpu -p no:cacheprovider -k "adam or hatrpo or gru or rnn or golden or activation" ) > $O/t_pk4.txt 2>&1
tail -3 $O/t_pk4.txt
