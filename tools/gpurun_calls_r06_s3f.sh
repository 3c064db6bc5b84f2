#!/bin/bash
# round 6, session 3, call f: critic stream on / off at the strong-scaling shares and on the other workloads
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r06s3
mkdir -p $O
cd $R
for rep in 1 2 3; do for cs in 1 0; do
for n in 512 1024 2048; do
HARL_CRITIC_STREAM=$cs timeout 600 python bench.py --threads-per-gpu $n --steps 20 --warmup 3 --cpu-cols 0 --instr-steps 0 --no-kernel-timing --no-other-configs 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('share$n critic_stream=$cs', round(d['ms_per_step'],3))"
done
for cfg in humanoid17:3 smac3s5z:10; do c=${cfg%%:*}; n=${cfg##*:}
HARL_CRITIC_STREAM=$cs timeout 600 python bench.py --config $c --steps $n --warmup 2 --cpu-cols 0 --instr-steps 0 --no-kernel-timing --no-other-configs 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$c critic_stream=$cs', round(d['ms_per_step'],3))"
done; done; done | sort -s -k1,1 | tee $O/ab_critic_stream2.txt
