# Round evidence run (GPU box, through gpurun):  bash tools/profile_round6.sh [tag]   (tag defaults to r06)
#   1. the driver's command: default bench line (mpe + every other workload incl. the N = 4096 shapes, CPU twins)
#   2. rocprofv3 --kernel-trace --stats summary per BASELINE configuration
#   3. HBM-traffic PMC passes over bench.py itself (tools/pmc_traffic.sh)
#   4. the RCCL branch with one rank, strong-scaling shares of the headline workload on one GPU (512 / 1024 / 2048 threads)
R=$GRAFT_REPO_ROOT
TAG=${1:-r06}
mkdir -p $R/gpurun_out/$TAG
cd /tmp && export TMPDIR=/tmp
( time timeout 1500 python $R/bench.py --steps 20 --warmup 3 ) > $R/gpurun_out/$TAG/bench_default.json 2> $R/gpurun_out/$TAG/bench_default.err
for c in mpe cheetah6 smac3s5z smac3s5z_n4096 humanoid17; do
  rm -rf /tmp/kt
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt -- python $R/bench.py --config $c --steps 3 --warmup 1 --cpu-cols 0 --instr-steps 0 --no-kernel-timing --no-other-configs > /dev/null 2>&1
  python $R/tools/prof_summary.py $(ls /tmp/kt/*/*kernel_trace.csv | head -1) --gaps 60 > $R/gpurun_out/$TAG/kernel_trace_$c.md 2>&1
done
timeout 300 python $R/bench.py --dist-single --cpu-cols 0 --no-other-configs > $R/gpurun_out/$TAG/bench_mpe_rccl_single.json 2> $R/gpurun_out/$TAG/bench_mpe_rccl_single.err
# what one rank of a strong-scaling run at the BASELINE's global 4096 threads does on 2 / 4 / 8 GPUs (no exchange: an upper bound
# on the speed-up an N-GPU run can show)
for n in 2048 1024 512; do
  timeout 300 python $R/bench.py --threads-per-gpu $n --steps 10 --cpu-cols 0 --no-other-configs --instr-steps 0 > $R/gpurun_out/$TAG/bench_mpe_share_$n.json 2> /dev/null
  timeout 300 python $R/bench.py --threads-per-gpu $n --steps 10 --cpu-cols 0 --no-other-configs --instr-steps 0 --dist-single > $R/gpurun_out/$TAG/bench_mpe_share_${n}_rccl.json 2> /dev/null
done
HARL_TRAFFIC_TAG=$TAG bash $R/tools/pmc_traffic.sh
cp $R/gpurun_out/pmc_traffic/${TAG}_hbm_traffic.json $R/gpurun_out/pmc_traffic/${TAG}_hbm_traffic.md $R/gpurun_out/$TAG/ 2>/dev/null
ls -la $R/gpurun_out/$TAG
