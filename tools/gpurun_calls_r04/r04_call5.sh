#!/bin/bash
set -x
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r04c5
mkdir -p $O
export TMPDIR=/tmp
timeout 300 python tools/diag_fused_last.py > $O/diag_fused_last.txt 2>&1
timeout 300 python tools/diag_fused_last.py 8192 > $O/diag_fused_last_small.txt 2>&1
for ws in 0 1; do
  HARL_DW_WS=$ws timeout 600 python bench.py --steps 10 --warmup 3 --cpu-cols 0 --no-other-configs > $O/bench_ws$ws.json 2> $O/bench_ws$ws.err
  HARL_DW_WS=$ws timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke_ws$ws.txt 2>&1
done
HARL_DW_WS=1 timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "gradients or golden or fused" > $O/pytest_ws1.txt 2>&1
cat $O/diag_fused_last.txt | tail -14
cat $O/diag_fused_last_small.txt | tail -14
for ws in 0 1; do python - <<P
import json
d=json.loads(open("$O/bench_ws$ws.json").read().strip().split("\n")[-1])
print("ws$ws", d["ms_per_step"], {k:(x["avg_ms"],x["n"]) for k,x in d["kernels"].items() if x["total_ms"]>0.3})
P
tail -1 $O/smoke_ws$ws.txt | cut -c1-150
done
tail -5 $O/pytest_ws1.txt
