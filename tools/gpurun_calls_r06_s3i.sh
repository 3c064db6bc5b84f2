#!/bin/bash
# round 6, session 3, call i: double-buffered GAE scan -- bit-exactness tests, smoke, timing at 4096 / 512 columns
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r06s3
mkdir -p $O
cd $R
( time timeout 900 python -m pytest tests/ -x -q -m gpu -p no:cacheprovider -k "gae or returns or smoke or rollout or golden" ) > $O/t_gae.txt 2>&1
tail -4 $O/t_gae.txt
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
for n in 4096 512; do
timeout 600 python bench.py --threads-per-gpu $n --steps 10 --warmup 3 --cpu-cols 0 --instr-steps 2 --no-other-configs 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('N=$n', round(d['ms_per_step'],3), {k:round(x['avg_ms'],4) for k,x in d['kernels'].items() if k in ('gae_returns','update_values')})"
done | tee $O/gae_timing.txt
