#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r04c24
mkdir -p $O
export TMPDIR=/tmp
timeout 500 python -m pytest tests/test_gpu_parity.py -q -k "hatrpo_train_matches and (rnn_box_h128 or rnn2_disc)" > $O/pytest.txt 2>&1
tail -25 $O/pytest.txt | cut -c1-400
