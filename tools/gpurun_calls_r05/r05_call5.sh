#!/bin/bash
# round 5, call 5: the recurrent workload after the two-wave training forward (k_gru_fwd_tp with saves), the pinned upload ring and the
# adaptive partial-row count; recurrent goldens; on-policy parity (three one-ulp twins)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r05c5
mkdir -p $O gpurun_out/parity
export TMPDIR=/tmp
(timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -k "rnn or recurrent or gru or generator_api or smac" 2>&1 | tail -8) > $O/t_rnn.txt 2>&1
for v in 1 0; do
  HARL_GRU_TP_SAVE=$v timeout 400 python bench.py --config smac3s5z --steps 5 --warmup 2 --cpu-cols 0 --no-other-configs > $O/bench_smac_tp$v.json 2> $O/bench_smac_tp$v.err
done
HARL_NWG=512 timeout 400 python bench.py --config smac3s5z --steps 5 --warmup 2 --cpu-cols 0 --no-other-configs > $O/bench_smac_nwg512.json 2> $O/bench_smac_nwg512.err
R=$GRAFT_REPO_ROOT
cd /tmp; rm -rf /tmp/kt
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt -- python $R/bench.py --config smac3s5z --steps 3 --warmup 1 --cpu-cols 0 --instr-steps 0 --no-kernel-timing --no-other-configs > /dev/null 2>&1
python $R/tools/prof_summary.py $(ls /tmp/kt/*/*kernel_trace.csv | head -1) --gaps 60 > $R/$O/kernel_trace_smac3s5z.md 2>&1
cd $R
(timeout 1200 python -m pytest tests/test_gpu_parity.py -q -s -k "bench_configuration_onpolicy" > $O/t_onpolicy.txt 2>&1)
for f in $O/bench_*.json; do python - <<P
import json
try:
    d=json.loads(open("$f").read().strip().split("\n")[-1])
    print("$f".split("/")[-1], round(d["ms_per_step"],3), {k:(x["avg_ms"],x["n"]) for k,x in d["kernels"].items() if k in ("gru_fwd","gru_bwd","adam_fold","reduce_partials")})
except Exception as e: print("$f", "ERR", e)
P
done
tail -3 $O/t_rnn.txt; tail -2 $O/t_onpolicy.txt | cut -c1-300; sed -n 5,14p $O/kernel_trace_smac3s5z.md | cut -c1-140
