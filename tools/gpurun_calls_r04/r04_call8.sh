#!/bin/bash
set -x
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r04c8
mkdir -p $O
export TMPDIR=/tmp
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1
timeout 600 python bench.py --steps 10 --warmup 3 --cpu-cols 0 --no-other-configs > $O/bench.json 2> $O/bench.err
timeout 600 python bench.py --steps 10 --warmup 3 --cpu-cols 0 --no-other-configs --dist-single > $O/bench_dist1.json 2> $O/bench_dist1.err
HARL_CRITIC_GROUP=0 timeout 600 python bench.py --steps 10 --warmup 3 --cpu-cols 0 --no-other-configs --dist-single > $O/bench_dist1_onegroup.json 2> $O/bench_dist1_onegroup.err
timeout 2400 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.txt 2>&1
tail -2 $O/smoke.txt | cut -c1-200
for f in bench bench_dist1 bench_dist1_onegroup; do python - <<P
import json
try:
    d=json.loads(open("$O/$f.json").read().strip().split("\n")[-1])
    print("$f", d["ms_per_step"], d["config"]["collective"], {k:(x["avg_ms"],x["n"]) for k,x in d["kernels"].items() if x["total_ms"]>0.3})
except Exception as e:
    print("$f ERR", e); print(open("$O/$f.err").read()[-1500:])
P
done
tail -6 $O/pytest_gpu.txt
