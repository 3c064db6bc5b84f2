#!/bin/bash
# GPU call 3 of round 4: BENCH-shape parity diagnostics on a sound library, new bench.py line (other_configs), quick GPU suite slice
set -x
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r04c3
mkdir -p $O
export TMPDIR=/tmp
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1
timeout 900 python tools/diag_bench_parity.py > $O/diag_bench_parity.txt 2>&1
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -k "forward or full_size" > $O/pytest_slice.txt 2>&1
tail -3 $O/smoke.txt | cut -c1-300
tail -3 $O/pytest_slice.txt
python - <<P
import json
d=json.loads(open("$O/bench_default.json").read().strip().split("\n")[-1])
print(d["ms_per_step"], d.get("end_to_end"), json.dumps(d.get("other_configs"))[:3000])
print({k:(x["avg_ms"],x["n"]) for k,x in d["kernels"].items() if x["total_ms"]>0.3})
P
tail -5 $O/bench_default.err
cat $O/diag_bench_parity.txt | tail -120
