#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
timeout 200 python - 2>&1 <<'P' | tail -3 | cut -c1-900
from tests import gpu_checks as G
r = G.check_rnn_update(G.RNN_SHAPES[0]); print({k: (f"{v:.2e}" if isinstance(v, float) else v) for k, v in r.items() if "grad" in k})
P
