#!/bin/bash
# GPU call 2 of round 4: safe variants of the cheaper pack (VGPR selector / v_pack away from MFMAs), BENCH-shape parity diagnostics
set -x
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r04c2
mkdir -p $O
export TMPDIR=/tmp
timeout 120 tools/_bin/split_cost > $O/split_cost.txt 2>&1
for v in hip sv5 sv6; do
  HARL_LIB=$v timeout 300 python bench.py --steps 10 --warmup 3 --cpu-cols 0 --no-other-configs > $O/bench_$v.json 2> $O/bench_$v.err
  HARL_LIB=$v timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke_$v.txt 2>&1
done
timeout 900 python tools/diag_bench_parity.py > $O/diag_bench_parity.txt 2>&1
grep "V[0-9]" $O/split_cost.txt
for v in hip sv5 sv6; do python - <<P
import json
try:
    d=json.loads(open("$O/bench_$v.json").read().strip().split("\n")[-1])
    print("$v", d["ms_per_step"], {k:(x["avg_ms"],x["n"]) for k,x in d["kernels"].items() if x["total_ms"]>0.5})
except Exception as e: print("$v", "ERR", e)
P
tail -2 $O/smoke_$v.txt | cut -c1-200
done
tail -100 $O/diag_bench_parity.txt
