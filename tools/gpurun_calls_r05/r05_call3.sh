#!/bin/bash
# round 5, call 3: version 2 of harl_mlp_bwd_dx_dw (pipelined rounds, A operand on the matrix-pipe transposes, one barrier per round),
# harl_build_seq through the recurrent goldens, on-policy parity with corrected assertions
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r05c3
mkdir -p $O gpurun_out/parity
export TMPDIR=/tmp
(timeout 300 python -m pytest tests/test_gpu_parity.py -x -q -k "whole_layer_backward" 2>&1 | tail -15) > $O/t_unit.txt 2>&1
(timeout 600 python -m pytest tests/test_gpu_parity.py -q -k "backward_variants_train_golden" 2>&1 | tail -8) > $O/t_variants.txt 2>&1
timeout 200 python tools/kbench.py --reps 20 bwd_ dw_hidden > $O/kbench.txt 2>&1
for m in 0 1 nofill; do
  HARL_BWD_FUSED=$m timeout 300 python bench.py --steps 10 --warmup 3 --cpu-cols 0 --no-other-configs > $O/bench_bwd$m.json 2> $O/bench_bwd$m.err
done
HARL_BWD_FUSED=1 HARL_BWD_K64=1 timeout 300 python bench.py --steps 10 --warmup 3 --cpu-cols 0 --no-other-configs > $O/bench_bwd1_k64.json 2> $O/bench_bwd1_k64.err
HARL_BWD_FUSED=1 HARL_LIB=phase timeout 300 python tools/phase_cycles.py --wg > $O/phase_cycles.txt 2>&1
for c in cheetah6 humanoid17 smac3s5z; do
  for f in 0 1; do
    HARL_BWD_FUSED=$f timeout 400 python bench.py --config $c --steps 3 --warmup 2 --cpu-cols 0 --no-other-configs > $O/bench_${c}_f$f.json 2> $O/bench_${c}_f$f.err
  done
done
(timeout 900 python -m pytest tests/test_gpu_parity.py -q -k "rnn or recurrent or gru or generator_api" 2>&1 | tail -8) > $O/t_rnn.txt 2>&1 &
(timeout 1200 python -m pytest tests/test_gpu_parity.py -q -s -k "bench_configuration_onpolicy" > $O/t_onpolicy.txt 2>&1) &
wait
for f in $O/bench_*.json; do python - <<P
import json
try:
    d=json.loads(open("$f").read().strip().split("\n")[-1])
    print("$f".split("/")[-1], round(d["ms_per_step"],3), {k:(x["avg_ms"],x["n"]) for k,x in d["kernels"].items() if x["total_ms"]>0.6 and k in ("bwd_dx_dw1","dw_hidden","bwd_dx","update_fwd","bwd_full","bwd_full_dw1","dw_input")})
except Exception as e: print("$f", "ERR", e)
P
done
tail -3 $O/t_unit.txt; tail -3 $O/t_variants.txt; tail -3 $O/t_rnn.txt; tail -2 $O/t_onpolicy.txt | cut -c1-300; tail -9 $O/kbench.txt
grep -A12 "k_bwd_dx_dw" $O/phase_cycles.txt | head -34
