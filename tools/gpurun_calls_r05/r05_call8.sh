#!/bin/bash
# round 5, call 8: one-shot all-reduce (2 and 4 processes on this GPU, bit for bit; the two-rank sharded update through it), the
# post-update stream of recurrent policies (goldens, bit-identity, SMAC A/B), SMAC kernel trace
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r05c8
mkdir -p $O gpurun_out/parity
export TMPDIR=/tmp
(timeout 600 python -m pytest tests/test_gpu_sharded.py -q -x 2>&1 | tail -15) > $O/t_sharded.txt 2>&1
(timeout 500 python -m pytest tests/test_gpu_parity.py -q -x -k "recurrent_train_matches or post_update_stream or (mappo_train and rnn) or gru128_train" 2>&1 | tail -8) > $O/t_rnn.txt 2>&1
for v in 1 0; do
  HARL_POST_STREAM=$v timeout 300 python bench.py --config smac3s5z --steps 5 --warmup 2 --cpu-cols 0 --no-other-configs > $O/bench_smac_post$v.json 2> $O/bench_smac_post$v.err
done
R=$GRAFT_REPO_ROOT
cd /tmp; rm -rf /tmp/kt
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt -- python $R/bench.py --config smac3s5z --steps 3 --warmup 1 --cpu-cols 0 --instr-steps 0 --no-kernel-timing --no-other-configs > /dev/null 2>&1
python $R/tools/prof_summary.py $(ls /tmp/kt/*/*kernel_trace.csv | head -1) --gaps 60 > $R/$O/kernel_trace_smac3s5z.md 2>&1
cd $R
for f in $O/bench_*.json; do python - <<P
import json
try:
    d=json.loads(open("$f").read().strip().split("\n")[-1])
    print("$f".split("/")[-1], round(d["ms_per_step"],3))
except Exception as e: print("$f", "ERR", e)
P
done
tail -6 $O/t_sharded.txt; tail -4 $O/t_rnn.txt; cat gpurun_out/parity/oneshot_allreduce_world*.json 2>/dev/null | grep -E "us_per|kind|world\""; grep -n "timeline" -A8 $O/kernel_trace_smac3s5z.md | cut -c1-150
