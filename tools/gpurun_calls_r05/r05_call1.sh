#!/bin/bash
# round 5, call 1: first hardware run of harl_mlp_bwd_dx_dw (unit check, goldens through it, A/B bench, phase timers),
# then the long parity checks (on-policy bench configuration, cheetah6 at full size) and the GPU leg of the drop-in
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r05c1
mkdir -p $O gpurun_out/parity
export TMPDIR=/tmp
nproc > $O/nproc.txt
(timeout 300 python -m pytest tests/test_gpu_parity.py -x -q -k "whole_layer_backward" 2>&1 | tail -15) > $O/t_unit.txt 2>&1
(timeout 600 python -m pytest tests/test_gpu_parity.py -q -k "test_train_matches_reference_golden or test_single_update_gradients or (many_slabs_per_wave and not width256)" 2>&1 | tail -15) > $O/t_golden.txt 2>&1
timeout 200 python tools/kbench.py --reps 20 bwd_ dw_hidden > $O/kbench.txt 2>&1
for m in 0 1 nofill; do
  HARL_BWD_FUSED=$m timeout 300 python bench.py --steps 10 --warmup 3 --cpu-cols 0 --no-other-configs > $O/bench_bwd$m.json 2> $O/bench_bwd$m.err
done
HARL_LIB=phase timeout 300 python tools/phase_cycles.py --wg > $O/phase_cycles.txt 2>&1
# long CPU-side checks in parallel (the oracle runs are worker processes on the host cores)
(timeout 1200 python -m pytest tests/test_gpu_parity.py -q -s -k "bench_configuration_onpolicy" > $O/t_onpolicy.txt 2>&1) &
(timeout 1500 python -m pytest tests/test_gpu_parity.py -q -s -k "cheetah6_full_size" > $O/t_cheetah6.txt 2>&1) &
(HARL_REFERENCE=$PWD/.refcopy timeout 900 python -m pytest tests/test_gpu_parity.py -q -s -k "dropin_under_reference_launcher" > $O/t_dropin.txt 2>&1) &
wait
for m in 0 1 nofill; do python - <<P
import json
try:
    d=json.loads(open("$O/bench_bwd$m.json").read().strip().split("\n")[-1])
    print("bwd$m", round(d["ms_per_step"],3), d["roofline"].get("clock_ghz"), {k:(x["avg_ms"],x["n"]) for k,x in d["kernels"].items() if x["total_ms"]>0.25})
except Exception as e: print("bwd$m", "ERR", e)
P
done
tail -3 $O/t_unit.txt; tail -3 $O/t_golden.txt; cat $O/kbench.txt | tail -12
grep -A12 "k_bwd_dx_dw" $O/phase_cycles.txt | head -40
tail -2 $O/t_onpolicy.txt | cut -c1-300; tail -2 $O/t_cheetah6.txt | cut -c1-300; tail -2 $O/t_dropin.txt | cut -c1-300
