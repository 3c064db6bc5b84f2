# HBM-traffic PMC passes over bench.py itself, one workload at a time (run on the GPU box through gpurun):
#   bash tools/pmc_traffic.sh [config ...]
R=$GRAFT_REPO_ROOT
CFGS=${@:-mpe cheetah6 smac3s5z humanoid17}
OUT=$R/gpurun_out/pmc_traffic
mkdir -p $OUT
(cd $R && git rev-parse --short HEAD 2>/dev/null || cat $R/.git_sha 2>/dev/null) > $OUT/git_sha.txt
cd /tmp && export TMPDIR=/tmp
for c in $CFGS; do
  mkdir -p $OUT/$c
  ARGS="--config $c --steps 1 --warmup 0 --cpu-cols 0 --no-other-configs"
  timeout 600 python $R/bench.py $ARGS --instr-steps 0 --time-all-tags > $OUT/$c/plain.json 2> $OUT/$c/plain.err
  rm -rf /tmp/pf /tmp/pw
  timeout 900 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/pf -- python $R/bench.py $ARGS --instr-steps 0 --no-kernel-timing > /dev/null 2>&1
  timeout 900 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/pw -- python $R/bench.py $ARGS --instr-steps 0 --no-kernel-timing > /dev/null 2>&1
  mkdir -p $OUT/$c/fetch $OUT/$c/write
  # keep only our kernels' rows (the CSVs of a 17-agent run are large)
  for d in fetch:pf write:pw; do
    for f in /tmp/${d#*:}/*/*counter_collection.csv; do
      (head -1 $f; grep -E '"(void )?k_|\(anonymous namespace\)::k_' $f) > $OUT/$c/${d%:*}/counter_collection.csv
    done
  done
done
python $R/tools/pmc_traffic.py report $OUT $OUT/${HARL_TRAFFIC_TAG:-r04}_hbm_traffic > $OUT/report.txt 2>&1
tail -40 $OUT/report.txt
