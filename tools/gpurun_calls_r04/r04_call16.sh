#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r04c16
mkdir -p $O
export TMPDIR=/tmp
HARL_LIB=phase timeout 300 python tools/phase_cycles.py --wg > $O/phase_wg.txt 2>&1
grep -A5 "workgroups (last launch)" $O/phase_wg.txt
