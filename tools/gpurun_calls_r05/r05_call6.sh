#!/bin/bash
# round 5, call 6: the wide GEMMs with LDS-shared weight panels (k_fwd_wide_sh): bit-for-bit against the streaming kernel, the wide /
# HATRPO parity tests through it, Humanoid-17x1 A/B
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r05c6
mkdir -p $O
export TMPDIR=/tmp
(timeout 600 python -m pytest tests/test_gpu_parity.py -q -x -k "shared_weight_panels or wide_input_first_layer" 2>&1 | tail -8) > $O/t_wide.txt 2>&1
for v in 1 0; do
  HARL_WIDE_SHARED=$v timeout 500 python bench.py --config humanoid17 --steps 3 --warmup 2 --cpu-cols 0 --no-other-configs > $O/bench_humanoid_sh$v.json 2> $O/bench_humanoid_sh$v.err
done
(timeout 1200 python -m pytest tests/test_gpu_parity.py -q -x -k "hatrpo or (parity_at_baseline_shapes and humanoid) or wide_obs" 2>&1 | tail -8) > $O/t_hatrpo.txt 2>&1
for f in $O/bench_*.json; do python - <<P
import json
try:
    d=json.loads(open("$f").read().strip().split("\n")[-1])
    print("$f".split("/")[-1], round(d["ms_per_step"],3), {k:(x["avg_ms"],x["n"]) for k,x in d["kernels"].items() if k in ("tangent_hidden","tangent_wide","fwd_wide","dw_input","reduce_partials")})
except Exception as e: print("$f", "ERR", e)
P
done
tail -4 $O/t_wide.txt; tail -4 $O/t_hatrpo.txt
