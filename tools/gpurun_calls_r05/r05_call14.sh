#!/bin/bash
# round 5, call 14: the driver's default bench command at the final commit (second box of the round for the headline), cheetah6 trace
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
R=$GRAFT_REPO_ROOT
O=gpurun_out/r05c14
mkdir -p $O
export TMPDIR=/tmp
(time timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_default_20.json 2> $O/bench_default_20.err) 2> $O/bench_default_20.time
cd /tmp; rm -rf /tmp/kt
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt -- python $R/bench.py --config cheetah6 --steps 3 --warmup 1 --cpu-cols 0 --instr-steps 0 --no-kernel-timing --no-other-configs > /dev/null 2>&1
python $R/tools/prof_summary.py $(ls /tmp/kt/*/*kernel_trace.csv | head -1) --gaps 60 > $R/$O/kernel_trace_cheetah6.md 2>&1
cd $R
python - <<P
import json
d=json.loads(open("$O/bench_default_20.json").read().strip().split("\n")[-1])
print(round(d["ms_per_step"],3), d["value"], d["roofline"]["kernel"], round(d["roofline"]["frac"],3), d["roofline"].get("traffic"))
for k,v in (d.get("other_configs") or {}).items(): print("   ", k, v.get("ms_per_step"), (v.get("cpu_baseline") or {}).get("value"), v.get("error"))
P
cat $O/bench_default_20.time
