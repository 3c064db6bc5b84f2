#!/bin/bash
# round 5, call 15: the input half of the GRU gates on the bf16 pipe (k_gru_gates_xs): recurrent parity tests, SMAC A/B
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r05c15
mkdir -p $O gpurun_out/parity
export TMPDIR=/tmp
(timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_sharded.py -q -x -k "recurrent_train or rnn_update or post_update or trpo_rnn_disc_h64 or (mappo and rnn) or (sharded and rnn) or generator_api or get_actions or smac3s5z_full" 2>&1 | tail -6) > $O/t_rnn.txt 2>&1
for v in 0 1; do
  HARL_GRU_GATES_F32=$v timeout 300 python bench.py --config smac3s5z --steps 5 --warmup 2 --cpu-cols 0 --no-other-configs > $O/bench_smac_gates_f32_$v.json 2> $O/bench_smac_gates_f32_$v.err
done
for f in $O/bench_*.json; do python - <<P
import json
try:
    d=json.loads(open("$f").read().strip().split("\n")[-1])
    print("$f".split("/")[-1], round(d["ms_per_step"],3), {k:(round(x["avg_ms"],4),x["n"]) for k,x in d["kernels"].items() if k in ("gru_fwd","gru_bwd")})
except Exception as e: print("$f", "ERR", e)
P
done
tail -4 $O/t_rnn.txt
