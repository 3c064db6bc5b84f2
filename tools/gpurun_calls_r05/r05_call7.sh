#!/bin/bash
# round 5, call 7 (calls 5 and 6 again: their results were lost with the container): recurrent + wide-GEMM tests, SMAC / Humanoid A/B of
# the two-wave training forward and the LDS-shared weight panels, the default bench line
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r05c7
mkdir -p $O gpurun_out/parity
export TMPDIR=/tmp
(timeout 700 python -m pytest tests/test_gpu_parity.py -q -x -k "rnn or recurrent or gru or generator_api or smac or shared_weight_panels or wide_input_first_layer or wide_obs" 2>&1 | tail -8) > $O/t_rnn_wide.txt 2>&1
for v in 1 0; do
  HARL_GRU_TP_SAVE=$v timeout 300 python bench.py --config smac3s5z --steps 5 --warmup 2 --cpu-cols 0 --no-other-configs > $O/bench_smac_tp$v.json 2> $O/bench_smac_tp$v.err
  HARL_WIDE_SHARED=$v timeout 400 python bench.py --config humanoid17 --steps 3 --warmup 2 --cpu-cols 0 --no-other-configs > $O/bench_humanoid_sh$v.json 2> $O/bench_humanoid_sh$v.err
done
(timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -k "hatrpo or (parity_at_baseline_shapes and humanoid)" 2>&1 | tail -8) > $O/t_hatrpo.txt 2>&1
(time timeout 600 python bench.py > $O/bench_default.json 2> $O/bench_default.err) 2> $O/bench_default.time
for f in $O/bench_*.json; do python - <<P
import json
try:
    d=json.loads(open("$f").read().strip().split("\n")[-1])
    print("$f".split("/")[-1], round(d["ms_per_step"],3), {k:(x["avg_ms"],x["n"]) for k,x in d["kernels"].items() if k in ("gru_fwd","gru_bwd","adam_fold","reduce_partials","tangent_hidden","tangent_wide","fwd_wide","dw_input","update_fwd","bwd_dx_dw1","dw_hidden","bwd_fused")})
    for k,v in (d.get("other_configs") or {}).items(): print("   ", k, v.get("ms_per_step"), (v.get("cpu_baseline") or {}).get("value"), v.get("error"))
except Exception as e: print("$f", "ERR", e)
P
done
tail -4 $O/t_rnn_wide.txt; tail -4 $O/t_hatrpo.txt; cat $O/bench_default.time
