#!/bin/bash
# round 6, session 2, call 7: the whole GPU suite at the current commit (the driver's command), timed
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r06h
mkdir -p $O
cd $R
( time timeout 1500 python -m pytest tests/ -x -q -m gpu -p no:cacheprovider ) > $O/gpu_tests_full.txt 2>&1
tail -12 $O/gpu_tests_full.txt
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
