#!/bin/bash
# round 5, call 12: the recurrent BASELINE workload at full size against the oracle
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r05c12
mkdir -p $O gpurun_out/parity
export TMPDIR=/tmp
(timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -s -k "smac3s5z_full_size" 2>&1 | tail -30) > $O/t_smac_full.txt 2>&1
tail -12 $O/t_smac_full.txt | cut -c1-2500
