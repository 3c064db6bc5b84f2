#!/bin/bash
# round 6, session 2, call 2: bitwise A/B of the one-launch trunk (all four shapes), the recurrent / wide-input tests, SMAC trace
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r06h
mkdir -p $O
cd $R
timeout 1200 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "trunk_in_one_launch or recurrent_train or gru_policy or hatrpo_gru or mappo_train or wide_input or rollout_loop or composed_gru" -p no:cacheprovider > $O/t2.txt 2>&1
tail -8 $O/t2.txt
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/kt
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt -- python $R/bench.py --config smac3s5z --steps 3 --warmup 1 --cpu-cols 0 --instr-steps 0 --no-kernel-timing --no-other-configs > /dev/null 2>&1
python $R/tools/prof_summary.py $(ls /tmp/kt/*/*kernel_trace.csv | head -1) --gaps 60 > $O/kernel_trace_smac3s5z.md 2>&1
head -45 $O/kernel_trace_smac3s5z.md
