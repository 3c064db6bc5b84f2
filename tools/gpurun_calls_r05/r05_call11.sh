#!/bin/bash
# round 5, call 11 (evidence): GRU kernels in isolation (two vs four waves per slab), the default bench line, kernel traces and
# HBM-traffic PMC passes of the headline and the recurrent workload, one-rank collective branch through RCCL and through the
# one-shot exchange, Humanoid with 256 partial rows
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
R=$GRAFT_REPO_ROOT
O=gpurun_out/r05c11
mkdir -p $O
export TMPDIR=/tmp
for v in 0 1; do (echo "HARL_GRU_QUAD=$v"; HARL_GRU_QUAD=$v timeout 200 python tools/gru_tp_check.py 2>&1 | grep "^L=") >> $O/gru_quad_micro.txt; done
(time timeout 600 python bench.py > $O/bench_default.json 2> $O/bench_default.err) 2> $O/bench_default.time
cd /tmp
for c in mpe smac3s5z humanoid17; do
  rm -rf /tmp/kt
  timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt -- python $R/bench.py --config $c --steps 3 --warmup 1 --cpu-cols 0 --instr-steps 0 --no-kernel-timing --no-other-configs > /dev/null 2>&1
  python $R/tools/prof_summary.py $(ls /tmp/kt/*/*kernel_trace.csv | head -1) --gaps 60 > $R/$O/kernel_trace_$c.md 2>&1
done
cd $R
timeout 300 python bench.py --dist-single --cpu-cols 0 --no-other-configs > $O/bench_mpe_rccl_single.json 2> $O/bench_mpe_rccl_single.err
HARL_ALLREDUCE=oneshot timeout 300 python bench.py --dist-single --cpu-cols 0 --no-other-configs > $O/bench_mpe_oneshot_single.json 2> $O/bench_mpe_oneshot_single.err
HARL_NWG=256 timeout 300 python bench.py --config humanoid17 --steps 3 --warmup 2 --cpu-cols 0 --no-other-configs > $O/bench_humanoid_nwg256.json 2> $O/bench_humanoid_nwg256.err
HARL_TRAFFIC_TAG=r05 bash tools/pmc_traffic.sh mpe smac3s5z > $O/pmc_report.txt 2>&1
cp gpurun_out/pmc_traffic/r05_hbm_traffic.json gpurun_out/pmc_traffic/r05_hbm_traffic.md $O/ 2>/dev/null
for f in $O/bench_*.json; do python - <<P
import json
try:
    d=json.loads(open("$f").read().strip().split("\n")[-1])
    print("$f".split("/")[-1], round(d["ms_per_step"],3), d["roofline"].get("kernel"), round(d["roofline"].get("frac") or 0,3))
    for k,v in (d.get("other_configs") or {}).items(): print("   ", k, v.get("ms_per_step"), (v.get("cpu_baseline") or {}).get("value"), v.get("error"))
except Exception as e: print("$f", "ERR", e)
P
done
cat $O/gru_quad_micro.txt; cat $O/bench_default.time; tail -12 $O/pmc_report.txt | cut -c1-200
