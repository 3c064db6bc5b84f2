#!/bin/bash
# round 5, call 21: the feed-forward selection that exposed the order-dependent failure (an ORACLE module global inherited from the previous test), after the fix in tests/conftest.py
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r05c21
mkdir -p $O
export TMPDIR=/tmp
(timeout 100 python -m pytest tests/ -q -x -m gpu -k "not (full_size or bench_configuration or hatrpo or humanoid or recurrent or rnn or gru or trpo or baseline_shapes or smac or rollout or get_actions or post_update)" 2>&1 | grep -E "AssertionError|passed|failed" | head -6) > $O/t_ff.txt 2>&1
cat $O/t_ff.txt | cut -c1-300
