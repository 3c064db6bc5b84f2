#!/bin/bash
set -x
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r04c10
mkdir -p $O
export TMPDIR=/tmp
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1
run() { name=$1; shift; env "$@" timeout 600 python bench.py --cpu-cols 0 --no-other-configs $EXTRA > $O/$name.json 2> $O/$name.err; }
EXTRA="--steps 10 --warmup 3" run mpe_new X=1
EXTRA="--steps 10 --warmup 3" run mpe_sepreduce HARL_FUSED_REDUCE=0
EXTRA="--config smac3s5z --steps 5 --warmup 2" run smac_new X=1
EXTRA="--config cheetah6 --steps 5 --warmup 2" run cheetah_new X=1
timeout 2400 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.txt 2>&1
tail -2 $O/smoke.txt | cut -c1-300
for f in mpe_new mpe_sepreduce smac_new cheetah_new; do python - <<P
import json
try:
    d=json.loads(open("$O/$f.json").read().strip().split("\n")[-1])
    ks=sorted(d["kernels"].items(), key=lambda kv:-kv[1]["total_ms"])[:10]
    print("$f", round(d["ms_per_step"],3), {k:(x["avg_ms"],x["n"]) for k,x in ks})
except Exception as e:
    print("$f ERR", e); print(open("$O/$f.err").read()[-800:])
P
done
tail -6 $O/pytest_gpu.txt
