#!/bin/bash
# round 6, session 3, call e: critic chain on its own stream (default) against one stream, new build
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r06s3
mkdir -p $O
cd $R
for rep in 1 2 3; do for cs in 1 0; do
for cfg in mpe:20 cheetah6:8; do c=${cfg%%:*}; n=${cfg##*:}
HARL_CRITIC_STREAM=$cs timeout 600 python bench.py --config $c --steps $n --warmup 3 --cpu-cols 0 --instr-steps 0 --no-other-configs 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d.get('roofline') or {}; print('$c critic_stream=$cs', round(d['ms_per_step'],3), r.get('kernel'), round(r.get('frac',0),4), round(r.get('avg_ms',0),4))"
done; done; done | sort -s -k1,1 | tee $O/ab_critic_stream.txt
