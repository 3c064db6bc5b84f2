#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r04c28
mkdir -p $O
export TMPDIR=/tmp
timeout 200 python bench.py --steps 3 --warmup 1 --other-steps 2 --cpu-cols 0 > $O/bench_default_short.json 2> $O/bench_default_short.err
python - <<P
import json
d=json.loads(open("$O/bench_default_short.json").read().strip().split("\n")[-1])
print(round(d["ms_per_step"],3), {k:(round(v["ms_per_step"],2) if "ms_per_step" in v else v) for k,v in d.get("other_configs",{}).items()})
P
tail -3 $O/bench_default_short.err
