#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r04c27
mkdir -p $O
export TMPDIR=/tmp
(cat .git_sha; timeout 200 python -m pytest tests/test_gpu_parity.py -q -k "hatrpo_train_matches or (hatrpo_gradient_fvp and 0) or (test_train_matches_reference_golden and hands)" 2>&1 | tail -4; timeout 100 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | cut -c1-200) > $O/final_regression.txt 2>&1
cat $O/final_regression.txt
