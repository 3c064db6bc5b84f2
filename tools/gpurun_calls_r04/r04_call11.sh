#!/bin/bash
set -x
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r04c11
mkdir -p $O
export TMPDIR=/tmp
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1
run() { name=$1; shift; env "$@" timeout 600 python bench.py --cpu-cols 0 --no-other-configs $EXTRA > $O/$name.json 2> $O/$name.err; }
EXTRA="--steps 10 --warmup 3" run mpe_wg64_fused X=1
EXTRA="--steps 10 --warmup 3" run mpe_wg64_sep HARL_FUSED_REDUCE=0
EXTRA="--config humanoid17 --steps 3 --warmup 2" run hum_one HARL_FUSED_REDUCE=0
EXTRA="--config humanoid17 --steps 3 --warmup 2" run hum_two HARL_FUSED_REDUCE=0 HARL_TANGENT_ONE_LAUNCH=0
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "trpo or hatrpo" > $O/pytest_trpo.txt 2>&1
tail -1 $O/smoke.txt | cut -c1-200
for f in mpe_wg64_fused mpe_wg64_sep hum_one hum_two; do python - <<P
import json
try:
    d=json.loads(open("$O/$f.json").read().strip().split("\n")[-1])
    ks=sorted(d["kernels"].items(), key=lambda kv:-kv[1]["total_ms"])[:10]
    print("$f", round(d["ms_per_step"],3), {k:(x["avg_ms"],x["n"]) for k,x in ks})
except Exception as e:
    print("$f ERR", e); print(open("$O/$f.err").read()[-800:])
P
done
tail -4 $O/pytest_trpo.txt
