#!/bin/bash
# round 5, call 19: call 17's selection again with the assertion text kept; then the failing test alone after the ones in front of it
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r05c19
mkdir -p $O
export TMPDIR=/tmp
(timeout 150 python -m pytest tests/ -q -x -m gpu -k "not (full_size or bench_configuration or hatrpo or humanoid or recurrent or rnn or gru or trpo or baseline_shapes or smac or rollout or get_actions or post_update)" 2>&1 | grep -E "AssertionError|assert |passed|failed|values_after" | head -12) > $O/t_ff.txt 2>&1
cat $O/t_ff.txt | cut -c1-300
