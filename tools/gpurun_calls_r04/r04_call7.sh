#!/bin/bash
set -x
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r04c7
mkdir -p $O
export TMPDIR=/tmp
for ws in 0 1; do
  HARL_DW_WS=$ws timeout 600 python bench.py --steps 10 --warmup 3 --cpu-cols 0 --no-other-configs > $O/bench_ws$ws.json 2> $O/bench_ws$ws.err
done
HARL_DW_WS=1 timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke_ws1.txt 2>&1
HARL_DW_WS=0 HARL_DEBUG_BLOCKS=1 timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -s -k "last_layer_in_loss" > $O/pytest_last.txt 2>&1
HARL_DW_WS=0 timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -s -k "bench_configuration" > $O/pytest_bench_cfg.txt 2>&1
for ws in 0 1; do python - <<P
import json
d=json.loads(open("$O/bench_ws$ws.json").read().strip().split("\n")[-1])
print("ws$ws", d["ms_per_step"], {k:(x["avg_ms"],x["n"]) for k,x in d["kernels"].items() if x["total_ms"]>0.3})
P
done
tail -1 $O/smoke_ws1.txt | cut -c1-150
grep -E "^block|^scalars|passed|failed" $O/pytest_last.txt | cut -c1-400
grep "bench-config parity" $O/pytest_bench_cfg.txt | cut -c1-4000
tail -4 $O/pytest_bench_cfg.txt | cut -c1-400
