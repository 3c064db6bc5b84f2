#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r04_final
mkdir -p $O
export TMPDIR=/tmp
(cat .git_sha; timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -15; timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | cut -c1-400) > $O/gpu_tests.txt 2>&1
timeout 300 python bench.py --cpu-cols 0 --no-other-configs > $O/bench_mpe.json 2> $O/bench_mpe.err
tail -8 $O/gpu_tests.txt
python - <<P
import json
d=json.loads(open("$O/bench_mpe.json").read().strip().split("\n")[-1])
print(round(d["ms_per_step"],3), d.get("git_sha"))
P
