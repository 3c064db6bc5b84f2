#!/bin/bash
# round 6, session 2, call 3: gate weight gradients on a second stream next to the trunk's backward -- A/B on the SMAC shapes
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r06h
mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "trunk_in_one_launch or recurrent_train" -p no:cacheprovider > $O/t3.txt 2>&1
tail -3 $O/t3.txt
for cfg in smac3s5z smac3s5z_n4096; do
  for f in 1 0 1 0; do
    HARL_TRUNK_DW_STREAM=$f timeout 600 python bench.py --config $cfg --steps 10 --warmup 3 --cpu-cols 0 --instr-steps 0 --no-kernel-timing > $O/bench_${cfg}_dws$f.json 2> $O/bench_${cfg}_dws$f.err
    python - <<PY
import json
try:
    d = json.loads(open("$O/bench_${cfg}_dws$f.json").read().strip().splitlines()[-1])
    print("$cfg dw_stream=$f ms_per_step", round(d["ms_per_step"], 3))
except Exception as e:
    print("$cfg dw_stream=$f failed", e)
PY
  done
done
