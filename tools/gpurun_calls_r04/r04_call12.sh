#!/bin/bash
set -x
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r04c12
mkdir -p $O
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -k "gru128_train" > $O/pytest_rn2.txt 2>&1
for v in hip ex ht both; do
  HARL_LIB=$v timeout 300 python bench.py --steps 10 --warmup 3 --cpu-cols 0 --no-other-configs > $O/bench_$v.json 2> $O/bench_$v.err
  HARL_LIB=$v timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke_$v.txt 2>&1
done
tail -4 $O/pytest_rn2.txt
for v in hip ex ht both; do python - <<P
import json
try:
    d=json.loads(open("$O/bench_$v.json").read().strip().split("\n")[-1])
    ks=sorted(d["kernels"].items(), key=lambda kv:-kv[1]["total_ms"])[:8]
    print("$v", round(d["ms_per_step"],3), {k:(x["avg_ms"],x["n"]) for k,x in ks})
except Exception as e:
    print("$v ERR", e); print(open("$O/bench_$v.err").read()[-800:])
P
tail -1 $O/smoke_$v.txt | cut -c1-160
done
