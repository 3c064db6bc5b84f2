# Round evidence run (GPU box, through gpurun):  bash tools/profile_round3.sh [tag]   (tag defaults to r03)
#   1. default bench line (mpe) + the three other BASELINE configurations
#   2. rocprofv3 --kernel-trace --stats summary per configuration
#   3. HBM-traffic PMC passes over bench.py itself (tools/pmc_traffic.sh)
R=$GRAFT_REPO_ROOT
TAG=${1:-r03}
mkdir -p $R/gpurun_out/$TAG
cd /tmp && export TMPDIR=/tmp
for c in mpe cheetah6 smac3s5z humanoid17; do
  timeout 600 python $R/bench.py --config $c > $R/gpurun_out/$TAG/bench_$c.json 2> $R/gpurun_out/$TAG/bench_$c.err
  rm -rf /tmp/kt
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt -- python $R/bench.py --config $c --steps 3 --warmup 1 --cpu-cols 0 --instr-steps 0 --no-kernel-timing > /dev/null 2>&1
  python $R/tools/prof_summary.py $(ls /tmp/kt/*/*kernel_trace.csv | head -1) --gaps 60 > $R/gpurun_out/$TAG/kernel_trace_$c.md 2>&1
done
timeout 300 python $R/bench.py --dist-single --cpu-cols 0 > $R/gpurun_out/$TAG/bench_mpe_rccl_single.json 2> $R/gpurun_out/$TAG/bench_mpe_rccl_single.err
HARL_TRAFFIC_TAG=$TAG bash $R/tools/pmc_traffic.sh
cp $R/gpurun_out/pmc_traffic/${TAG}_hbm_traffic.json $R/gpurun_out/pmc_traffic/${TAG}_hbm_traffic.md $R/gpurun_out/$TAG/ 2>/dev/null
ls -la $R/gpurun_out/$TAG
