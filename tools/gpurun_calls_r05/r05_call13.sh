#!/bin/bash
# round 5, call 13: the full GPU suite (the driver's command) + smoke at the final commit
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r05c13
mkdir -p $O gpurun_out/parity
export TMPDIR=/tmp
(cat .git_sha 2>/dev/null; time timeout 1500 python -m pytest tests/ -x -q -m gpu --durations=15 2>&1 | tail -40) > $O/gpu_tests.txt 2>&1
(timeout 200 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3 | cut -c1-600) >> $O/gpu_tests.txt 2>&1
tail -32 $O/gpu_tests.txt | cut -c1-300
