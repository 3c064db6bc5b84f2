#!/bin/bash
# round 5, call 17: the fast feed-forward GPU tests at the final commit (the library was rebuilt after the full suite of call 13)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r05c17
mkdir -p $O
export TMPDIR=/tmp
(cat .git_sha; timeout 200 python -m pytest tests/ -q -x -m gpu -k "not (full_size or bench_configuration or hatrpo or humanoid or recurrent or rnn or gru or trpo or baseline_shapes or smac or rollout or get_actions or post_update)" 2>&1 | tail -5) > $O/t_ff.txt 2>&1
tail -4 $O/t_ff.txt
