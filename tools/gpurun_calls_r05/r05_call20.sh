#!/bin/bash
# round 5, call 20: the checkpoint-compat check on POISONED memory (the caching allocator's free blocks filled with NaN bit
# patterns first), per GRU kernel variant: which launch reads memory it has not written?
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r05c20
mkdir -p $O
export TMPDIR=/tmp
for cfg in "0 1" "1 1" "1 0"; do
  set -- $cfg
  echo "== HARL_GRU_GATES_F32=$1 HARL_GRU_QUAD=$2" >> $O/t.txt
  HARL_GRU_GATES_F32=$1 HARL_GRU_QUAD=$2 timeout 100 python - >> $O/t.txt 2>&1 <<P
import tempfile, torch
from tests import gpu_checks as G
blocks = [torch.full((n,), float("nan"), device="cuda") for n in (1 << 26, 1 << 22, 1 << 18, 1 << 14, 1 << 10) for _ in range(4)]
torch.cuda.synchronize(); del blocks
res = G.check_checkpoint_compat(tempfile.mkdtemp())
print({k_: v for k_, v in res.items() if v != 0.0})
P
done
cat $O/t.txt | grep -v amdgpu.ids | cut -c1-300
