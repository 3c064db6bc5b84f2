#!/bin/bash
# round 6, session 3, call c: the library without packed fp32 VALU in the MFMA kernels as the DEFAULT build: whole GPU suite (timed),
# A/B of the static-priority variant (libharl_prio.so), the driver's default bench
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r06s3
mkdir -p $O
cd $R
for rep in 1 2 3; do for v in hip prio; do
HARL_LIB=$v timeout 600 python bench.py --steps 20 --warmup 3 --cpu-cols 0 --instr-steps 0 --no-kernel-timing --no-other-configs 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('mpe $v', round(d['ms_per_step'],3))"
done; done | tee $O/ab_prio_mpe.txt
( time timeout 1500 python -m pytest tests/ -x -q -m gpu -p no:cacheprovider ) > $O/gpu_tests_nopk.txt 2>&1
tail -8 $O/gpu_tests_nopk.txt
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1 >> $O/gpu_tests_nopk.txt
cd /tmp
( time timeout 1500 python $R/bench.py ) > $O/bench_default_nopk.json 2> $O/bench_default_nopk.err
tail -3 $O/bench_default_nopk.err
