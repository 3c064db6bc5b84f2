#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r04c18
mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -q -k "hatrpo_gradient_fvp or hatrpo_gru_gradient or gru_policy_forward" > $O/pytest_bars.txt 2>&1
tail -30 $O/pytest_bars.txt
timeout 300 python - > $O/bars_values.txt 2>&1 <<'P'
from tests import gpu_checks as G
for i in range(4):
    r = G.check_rnn_update(G.RNN_SHAPES[i]); print("rnn_update", i, {k: (f"{v:.2e}" if isinstance(v, float) else v) for k, v in r.items()})
for i in range(3):
    r = G.check_trpo_rnn(G.RNN_SHAPES[i]); print("trpo_rnn", i, {k: (f"{v:.2e}" if isinstance(v, float) else v) for k, v in r.items() if "cg" in k or "surrogate" in k or "fvp" in k})
for i in (0, 1, 2, 4):
    r = G.check_trpo(G.FWD_SHAPES[i]); print("trpo", i, {k: (f"{v:.2e}" if isinstance(v, float) else v) for k, v in r.items() if "cg" in k})
P
cat $O/bars_values.txt | cut -c1-1500
