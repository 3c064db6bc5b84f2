#!/bin/bash
set -x
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r04c15
mkdir -p $O
export TMPDIR=/tmp
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1
timeout 300 python bench.py --steps 10 --warmup 3 --cpu-cols 0 --no-other-configs > $O/bench_mpe.json 2> $O/bench_mpe.err
HARL_LIB=phase timeout 300 python tools/phase_cycles.py > $O/phase_cycles.txt 2>&1
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "fused_update or last_layer or single_update or mlp_forward or train_matches_reference_golden or full_size" > $O/pytest_fused.txt 2>&1
tail -1 $O/smoke.txt | cut -c1-200
python - <<P
import json
d=json.loads(open("$O/bench_mpe.json").read().strip().split("\n")[-1])
ks=sorted(d["kernels"].items(), key=lambda kv:-kv[1]["total_ms"])[:8]
print(round(d["ms_per_step"],3), {k:(x["avg_ms"],x["n"]) for k,x in ks})
P
grep -A12 "k_upd_fwd actor TRAIN\|k_upd_fwd actor logp" $O/phase_cycles.txt
tail -4 $O/pytest_fused.txt
