# Round evidence run (GPU box, through gpurun):  bash tools/profile_round5.sh [tag]   (tag defaults to r05)
#   1. default bench line (mpe) + the three other BASELINE configurations
#   2. rocprofv3 --kernel-trace --stats summary per configuration
#   3. HBM-traffic PMC passes over bench.py itself (tools/pmc_traffic.sh)
R=$GRAFT_REPO_ROOT
TAG=${1:-r05}
mkdir -p $R/gpurun_out/$TAG
cd /tmp && export TMPDIR=/tmp
for c in mpe cheetah6 smac3s5z humanoid17; do
  timeout 900 python $R/bench.py --config $c $([ $c = mpe ] || echo --no-other-configs) > $R/gpurun_out/$TAG/bench_$c.json 2> $R/gpurun_out/$TAG/bench_$c.err
  rm -rf /tmp/kt
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt -- python $R/bench.py --config $c --steps 3 --warmup 1 --cpu-cols 0 --instr-steps 0 --no-kernel-timing --no-other-configs > /dev/null 2>&1
  python $R/tools/prof_summary.py $(ls /tmp/kt/*/*kernel_trace.csv | head -1) --gaps 60 > $R/gpurun_out/$TAG/kernel_trace_$c.md 2>&1
done
# the headline workload once more with the critic chain on the main stream (per-kernel durations without a neighbour kernel)
rm -rf /tmp/kt
HARL_CRITIC_STREAM=0 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt -- python $R/bench.py --steps 3 --warmup 1 --cpu-cols 0 --instr-steps 0 --no-kernel-timing --no-other-configs > /dev/null 2>&1
python $R/tools/prof_summary.py $(ls /tmp/kt/*/*kernel_trace.csv | head -1) --gaps 60 > $R/gpurun_out/$TAG/kernel_trace_mpe_single_stream.md 2>&1
HARL_CRITIC_STREAM=0 timeout 300 python $R/bench.py --cpu-cols 0 --no-other-configs > $R/gpurun_out/$TAG/bench_mpe_single_stream.json 2> /dev/null
# per-phase shader cycles of the persistent kernels (debug library with -DHARL_PHASE_TIMING, if it was built)
if [ -f $R/harl_amd/lib/libharl_phase.so ]; then (cd $R && HARL_LIB=phase timeout 300 python tools/phase_cycles.py > $R/gpurun_out/$TAG/phase_cycles.txt 2>&1); fi
[ -x $R/tools/_bin/valu_cost ] && $R/tools/_bin/valu_cost > $R/gpurun_out/$TAG/valu_cost.txt 2>&1
[ -x $R/tools/_bin/split_cost ] && $R/tools/_bin/split_cost > $R/gpurun_out/$TAG/split_cost.txt 2>&1
timeout 300 python $R/bench.py --dist-single --cpu-cols 0 --no-other-configs > $R/gpurun_out/$TAG/bench_mpe_rccl_single.json 2> $R/gpurun_out/$TAG/bench_mpe_rccl_single.err
HARL_TRAFFIC_TAG=$TAG bash $R/tools/pmc_traffic.sh
cp $R/gpurun_out/pmc_traffic/${TAG}_hbm_traffic.json $R/gpurun_out/pmc_traffic/${TAG}_hbm_traffic.md $R/gpurun_out/$TAG/ 2>/dev/null
ls -la $R/gpurun_out/$TAG
