#!/bin/bash
# round 6, session 3, call a: where the 1/8 share of the BASELINE shape (512 rollout threads per rank = the strong-scaling share at
# 8 GPUs) spends its 4.1 ms: kernels table, raw kernel trace queue by queue, host profile
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r06s3
mkdir -p $O
cd $R
for n in 512 1024; do
timeout 600 python bench.py --threads-per-gpu $n --steps 20 --warmup 5 --cpu-cols 0 --instr-steps 2 --no-other-configs > $O/bench_share_$n.json 2> $O/bench_share_$n.err
python - <<P
import json
d=json.loads(open("$O/bench_share_$n.json").read().strip().splitlines()[-1])
print("share $n", round(d["ms_per_step"],3))
for k,v in d.get("kernels",{}).items(): print("   %-20s n %3d avg %.4f total %.3f"%(k,v["n"],v["avg_ms"],v["total_ms"]))
P
done
cd /tmp; export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $O/trace512 -o share512 -- python $R/bench.py --threads-per-gpu 512 --steps 6 --warmup 3 --cpu-cols 0 --instr-steps 0 --no-kernel-timing --no-other-configs > $O/trace512.log 2>&1
f=$(find $O/trace512 -name "*kernel_trace.csv" | head -1)
python $R/tools/trace_queues.py $f 15 > $O/share512_queues.md 2>&1
cat $O/share512_queues.md
rm -rf $O/trace512
cd $R
python - > $O/prof_host_share512.txt 2>&1 <<P
import cProfile, pstats, sys, io, os
sys.argv = ["bench.py", "--threads-per-gpu", "512", "--cpu-cols", "0", "--instr-steps", "0", "--steps", "20", "--no-kernel-timing", "--no-other-configs"]
import bench
pr = cProfile.Profile()
pr.enable()
bench.main()
pr.disable()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(30)
print(s.getvalue()[:7000])
P
head -60 $O/prof_host_share512.txt
