#!/bin/bash
# round 6, session 3, call j: NO_PK on the remaining files (gru_cell.hip + elementwise.hip = pkE; gru.hip = pkG) on the workloads that use them
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r06s3
mkdir -p $O
cd $R
for rep in 1 2 3; do for v in hip pkE pkG; do
for cfg in hatrpo_gru128:3 smac3s5z:10 smac3s5z_n4096:4; do c=${cfg%%:*}; n=${cfg##*:}
HARL_LIB=$v timeout 600 python bench.py --config $c --steps $n --warmup 2 --cpu-cols 0 --instr-steps 0 --no-kernel-timing --no-other-configs 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$c $v', round(d['ms_per_step'],3))"
done; done; done | sort -s -k1,1 | tee $O/ab_pk3.txt
