#!/bin/bash
# round 5, call 2: two-stream backward (weight gradient next to the register-lean dx launch) A/B, variants' goldens, on-policy
# parity with the corrected assertions
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r05c2
mkdir -p $O gpurun_out/parity
export TMPDIR=/tmp
(timeout 600 python -m pytest tests/test_gpu_parity.py -q -k "backward_variants_train_golden or whole_layer_backward" 2>&1 | tail -8) > $O/t_variants.txt 2>&1
run() { # name, env...
  n=$1; shift
  env "$@" timeout 300 python bench.py --steps 10 --warmup 3 --cpu-cols 0 --no-other-configs > $O/bench_$n.json 2> $O/bench_$n.err
}
run base HARL_BWD_STREAMS=0
run streams HARL_BWD_STREAMS=1
run streams_nwg256 HARL_BWD_STREAMS=1 HARL_NWG=256
run streams_nocritic HARL_BWD_STREAMS=1 HARL_CRITIC_STREAM=0
run base_nocritic HARL_BWD_STREAMS=0 HARL_CRITIC_STREAM=0
for c in cheetah6 humanoid17; do
  for st in 0 1; do
    HARL_BWD_STREAMS=$st timeout 400 python bench.py --config $c --steps 3 --warmup 2 --cpu-cols 0 --no-other-configs > $O/bench_${c}_st$st.json 2> $O/bench_${c}_st$st.err
  done
done
HARL_BWD_FUSED=1 timeout 400 python bench.py --config humanoid17 --steps 3 --warmup 2 --cpu-cols 0 --no-other-configs > $O/bench_humanoid17_fused.json 2> $O/bench_humanoid17_fused.err
HARL_BWD_FUSED=1 timeout 400 python bench.py --config cheetah6 --steps 3 --warmup 2 --cpu-cols 0 --no-other-configs > $O/bench_cheetah6_fused.json 2> $O/bench_cheetah6_fused.err
# kernel trace of the two-stream arrangement: do the two kernels overlap?
cd /tmp; rm -rf /tmp/kt
HARL_BWD_STREAMS=1 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --cpu-cols 0 --instr-steps 0 --no-kernel-timing --no-other-configs > /dev/null 2>&1
python $GRAFT_REPO_ROOT/tools/prof_summary.py $(ls /tmp/kt/*/*kernel_trace.csv | head -1) --gaps 40 > $GRAFT_REPO_ROOT/$O/kernel_trace_streams.md 2>&1
cp $(ls /tmp/kt/*/*kernel_trace.csv | head -1) $GRAFT_REPO_ROOT/$O/kernel_trace_streams.csv 2>/dev/null
cd $GRAFT_REPO_ROOT
(timeout 1200 python -m pytest tests/test_gpu_parity.py -q -s -k "bench_configuration_onpolicy" > $O/t_onpolicy.txt 2>&1)
for f in $O/bench_*.json; do python - <<P
import json
try:
    d=json.loads(open("$f").read().strip().split("\n")[-1])
    print("$f".split("/")[-1], round(d["ms_per_step"],3), {k:(x["avg_ms"],x["n"]) for k,x in d["kernels"].items() if x["total_ms"]>0.6 and k in ("bwd_dx_dw1","dw_hidden","bwd_dx","update_fwd","bwd_full","bwd_full_dw1")})
except Exception as e: print("$f", "ERR", e)
P
done
tail -3 $O/t_variants.txt; tail -2 $O/t_onpolicy.txt | cut -c1-300
