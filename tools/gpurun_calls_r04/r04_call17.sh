#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r04c17
mkdir -p $O
export TMPDIR=/tmp
timeout 300 python bench.py --steps 10 --warmup 3 --cpu-cols 0 --no-other-configs > $O/bench_mpe.json 2> $O/bench_mpe.err
HARL_LIB=phase timeout 300 python tools/phase_cycles.py --wg > $O/phase_wg.txt 2>&1
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "fused_update or last_layer or single_update or mlp_forward or train_matches_reference_golden or full_size" > $O/pytest_fused.txt 2>&1
python - <<P
import json
d=json.loads(open("$O/bench_mpe.json").read().strip().split("\n")[-1])
ks=sorted(d["kernels"].items(), key=lambda kv:-kv[1]["total_ms"])[:8]
print(round(d["ms_per_step"],3), {k:(x["avg_ms"],x["n"]) for k,x in ks})
P
grep -A14 "k_upd_fwd actor TRAIN:" $O/phase_wg.txt | head -34
tail -3 $O/pytest_fused.txt
