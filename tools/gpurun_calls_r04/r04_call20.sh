#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r04c20
mkdir -p $O
export TMPDIR=/tmp
timeout 400 python bench.py > $O/bench_default.json 2> $O/bench_default.err
timeout 400 python bench.py --cpu-cols 4096 --cpu-reps 1 --steps 2 --warmup 1 --no-other-configs --instr-steps 0 > $O/bench_cpu4096.json 2> $O/bench_cpu4096.err
python - <<P
import json
d=json.loads(open("$O/bench_default.json").read().strip().split("\n")[-1])
print(round(d["ms_per_step"],3), d.get("git_sha"), {k:round(v["ms_per_step"],2) for k,v in d.get("other_configs",{}).items()}, d["cpu_baseline"]["value"])
d=json.loads(open("$O/bench_cpu4096.json").read().strip().split("\n")[-1])
print(d["cpu_baseline"])
P
