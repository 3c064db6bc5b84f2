set -x
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
python $R/bench.py > $R/gpurun_out/bench_r01.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt -- python $R/bench.py --steps 3 --warmup 1 --cpu-cols 0 --instr-steps 0 > /dev/null 2>&1
python $R/tools/prof_summary.py $(ls /tmp/kt/*/*kernel_trace.csv | head -1) --gaps 60 > $R/gpurun_out/kt_summary.md 2>&1
python $R/tools/kbench.py --reps 20 > $R/gpurun_out/kb_r01.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/pf -- python $R/tools/kbench.py --reps 3 > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/pw -- python $R/tools/kbench.py --reps 3 > /dev/null 2>&1
python $R/tools/hbm_traffic.py raw /tmp/pf /tmp/pw > $R/gpurun_out/hbm_traffic_raw.txt
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE SQ_INSTS_VALU --kernel-trace --output-format csv -d /tmp/pmc -- python $R/tools/kbench.py --reps 6 > /dev/null 2>&1
python $R/tools/pmc_summary.py /tmp/pmc/*/*_counter_collection.csv > $R/gpurun_out/pmc_r01.md
hipcc --offload-arch=gfx950 -O3 -Wno-unused-value $R/tools/mfma_bf16x3.hip -o /tmp/x && /tmp/x > $R/gpurun_out/bf16x3.txt 2>&1
tail -1 $R/gpurun_out/bench_r01.log | cut -c1-600
