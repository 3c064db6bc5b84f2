#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r04c23
mkdir -p $O
export TMPDIR=/tmp
timeout 500 python -m pytest tests/test_gpu_parity.py -q -k "hatrpo_composed_gru" > $O/pytest.txt 2>&1
tail -25 $O/pytest.txt | cut -c1-400
