#!/bin/bash
# round 6, session 3, final call: the driver's three steps at the final commit -- pytest -m gpu (timed), smoke(), python bench.py
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r06s3
mkdir -p $O
cd $R
( time timeout 1500 python -m pytest tests/ -x -q -m gpu -p no:cacheprovider ) > $O/gpu_tests_final4.txt 2>&1
tail -8 $O/gpu_tests_final4.txt
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1 >> $O/gpu_tests_final4.txt
cd /tmp
( time timeout 1500 python $R/bench.py ) > $O/bench_default_final4.json 2> $O/bench_default_final4.err
tail -3 $O/bench_default_final4.err
( time timeout 600 python $R/bench.py --gpus 1 --steps 20 --warmup 5 --no-other-configs ) > $O/bench_steps20_final4.json 2> $O/bench_steps20_final4.err
