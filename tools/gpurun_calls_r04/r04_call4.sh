#!/bin/bash
# GPU call 4 of round 4: per-action-width instantiations of the fused forward + loss launch: full GPU suite, bench, diagnostics
set -x
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r04c4
mkdir -p $O
export TMPDIR=/tmp
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1
timeout 600 python bench.py --steps 10 --warmup 3 --cpu-cols 0 --no-other-configs > $O/bench.json 2> $O/bench.err
timeout 900 python tools/diag_bench_parity.py > $O/diag_bench_parity.txt 2>&1
timeout 1500 python -m pytest tests -m gpu -x -q -k "not bench_configuration" > $O/pytest_gpu.txt 2>&1
tail -3 $O/smoke.txt | cut -c1-300
tail -5 $O/pytest_gpu.txt
python - <<P
import json
d=json.loads(open("$O/bench.json").read().strip().split("\n")[-1])
print(d["ms_per_step"], {k:(x["avg_ms"],x["n"]) for k,x in d["kernels"].items() if x["total_ms"]>0.3})
P
grep -A50 "where the hip - f32" $O/diag_bench_parity.txt | head -40
grep "first update, agent 0\|ratio quantiles" $O/diag_bench_parity.txt
