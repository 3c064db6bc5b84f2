#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r04c26
mkdir -p $O
export TMPDIR=/tmp
timeout 500 python -m pytest tests/test_gpu_parity.py -q -k "hatrpo_width256 or (hatrpo_train_matches and h256x2)" > $O/pytest.txt 2>&1
tail -25 $O/pytest.txt | cut -c1-400
