#!/bin/bash
# round 6, session 2, call 1: the one-launch 64-wide trunk (csrc/trunk.hip) -- bitwise A/B, the recurrent goldens, SMAC bench A/B
set -x
O=gpurun_out/r06h
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "trunk_in_one_launch or recurrent_train_matches or gru_policy_forward or hatrpo_gru_gradient" -p no:cacheprovider > $O/t1.txt 2>&1
tail -15 $O/t1.txt
for cfg in smac3s5z smac3s5z_n4096; do
  for f in 1 0; do
    HARL_TRUNK_FUSED=$f timeout 600 python bench.py --config $cfg --steps 10 --warmup 3 --cpu-cols 0 > $O/bench_${cfg}_fused$f.json 2> $O/bench_${cfg}_fused$f.err
    python - <<PY
import json
try:
    d = json.loads(open("$O/bench_${cfg}_fused$f.json").read().strip().splitlines()[-1])
    print("$cfg fused=$f ms_per_step", d["ms_per_step"], "value", d["value"])
except Exception as e:
    print("$cfg fused=$f failed", e)
PY
  done
done
