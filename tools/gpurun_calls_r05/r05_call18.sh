#!/bin/bash
# round 5, call 18: isolate the checkpoint-compat failure of call 17 (recurrent critic, 8 rows, one GRU step)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r05c18
mkdir -p $O
export TMPDIR=/tmp
for cfg in "0 1" "1 1" "0 0" "1 0"; do
  set -- $cfg
  echo "== HARL_GRU_GATES_F32=$1 HARL_GRU_QUAD=$2" >> $O/t.txt
  HARL_GRU_GATES_F32=$1 HARL_GRU_QUAD=$2 timeout 100 python - >> $O/t.txt 2>&1 <<P
import tempfile
from tests import gpu_checks as G
for k in range(3):
    res = G.check_checkpoint_compat(tempfile.mkdtemp())
    print({k_: v for k_, v in res.items() if v != 0.0})
P
done
cat $O/t.txt | cut -c1-400
