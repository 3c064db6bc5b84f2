#!/bin/bash
# GPU call 1 of round 4: split microbenchmark, A/B of the split variants in the real step, BENCH-configuration parity test
set -x
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r04c1
mkdir -p $O
export TMPDIR=/tmp
timeout 120 tools/_bin/split_cost > $O/split_cost.txt 2>&1
for v in hip sv2 sv3 sv4; do
  HARL_LIB=$v timeout 300 python bench.py --steps 10 --warmup 3 --cpu-cols 0 --no-other-configs > $O/bench_$v.json 2> $O/bench_$v.err
  HARL_LIB=$v timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke_$v.txt 2>&1
done
timeout 900 python -m pytest tests/test_gpu_parity.py -k bench_configuration -x -q -s > $O/bench_parity.txt 2>&1
tail -5 $O/bench_parity.txt
cat $O/split_cost.txt
for v in hip sv2 sv3 sv4; do python - <<P
import json
try:
    d=json.loads(open("$O/bench_$v.json").read().strip().split("\n")[-1])
    print("$v", d["ms_per_step"], {k:(x["avg_ms"],x["n"]) for k,x in d["kernels"].items() if x["total_ms"]>0.5})
except Exception as e: print("$v", "ERR", e)
P
tail -2 $O/smoke_$v.txt | cut -c1-300
done
