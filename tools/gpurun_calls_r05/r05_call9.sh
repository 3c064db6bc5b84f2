#!/bin/bash
# round 5, call 9: four waves per slab in the GRU forward (k_gru_fwd_q): recurrent parity tests through it, SMAC A/B, and the new
# coverage workload (HATRPO on the composed 128-wide GRU)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r05c9
mkdir -p $O gpurun_out/parity
export TMPDIR=/tmp
(timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_sharded.py -q -x -k "recurrent or gru or rnn or post_update" 2>&1 | tail -12) > $O/t_rnn.txt 2>&1
for v in 1 0; do
  HARL_GRU_QUAD=$v timeout 300 python bench.py --config smac3s5z --steps 5 --warmup 2 --cpu-cols 0 --no-other-configs > $O/bench_smac_quad$v.json 2> $O/bench_smac_quad$v.err
done
timeout 400 python bench.py --config hatrpo_gru128 --steps 2 --warmup 1 --no-other-configs > $O/bench_hatrpo_gru128.json 2> $O/bench_hatrpo_gru128.err
for f in $O/bench_*.json; do python - <<P
import json
try:
    d=json.loads(open("$f").read().strip().split("\n")[-1])
    print("$f".split("/")[-1], round(d["ms_per_step"],3), {k:(round(x["avg_ms"],4),x["n"]) for k,x in d["kernels"].items() if k in ("gru_fwd","gru_bwd")}, (d.get("cpu_baseline") or {}).get("value"))
except Exception as e: print("$f", "ERR", e)
P
done
tail -6 $O/t_rnn.txt; tail -3 $O/bench_hatrpo_gru128.err
