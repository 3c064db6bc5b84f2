#!/bin/bash
# round 6, session 3, call g: re-check of round-4/5 routing choices on the new build (mpe, 20 steps, interleaved)
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r06s3
mkdir -p $O
cd $R
for rep in 1 2 3; do
for e in "X=1" "HARL_FUSED_UPDATE=1" "HARL_FUSED_UPDATE=logp" "HARL_BWD_FUSED=0" "HARL_NWG=512" "HARL_CRITIC_FIRST=0"; do
env $e timeout 600 python bench.py --steps 20 --warmup 3 --cpu-cols 0 --instr-steps 0 --no-kernel-timing --no-other-configs 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('mpe $e', round(d['ms_per_step'],3))"
done; done | sort -s -k2,2 | tee $O/ab_routing.txt
