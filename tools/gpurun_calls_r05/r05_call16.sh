#!/bin/bash
# round 5, call 16: every recurrent GPU test through the final GRU kernels (four waves per slab, bf16 input gates) + the default bench line
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r05c16
mkdir -p $O gpurun_out/parity
export TMPDIR=/tmp
(cat .git_sha; timeout 700 python -m pytest tests/ -q -x -m gpu -k "recurrent or gru or rnn or post_update or smac or get_actions or rollout" 2>&1 | tail -6) > $O/t_rnn_all.txt 2>&1
(time timeout 600 python bench.py > $O/bench_default.json 2> $O/bench_default.err) 2> $O/bench_default.time
python - <<P
import json
d=json.loads(open("$O/bench_default.json").read().strip().split("\n")[-1])
print(round(d["ms_per_step"],3), d["value"], d["roofline"]["kernel"], round(d["roofline"]["frac"],3), d["roofline"].get("traffic"))
for k,v in (d.get("other_configs") or {}).items(): print("   ", k, v.get("ms_per_step"), (v.get("cpu_baseline") or {}).get("value"), v.get("error"))
P
tail -4 $O/t_rnn_all.txt
