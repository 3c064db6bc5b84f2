#!/bin/bash
# round 5, call 4: the default bench line (new fields: clock_ghz, bf16x6, CPU twins of the attached workloads), kernel traces of the
# two launch-bound workloads after harl_build_seq, the on-policy parity check
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r05c4
mkdir -p $O gpurun_out/parity
export TMPDIR=/tmp
(time timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err) 2> $O/bench_default.time
R=$GRAFT_REPO_ROOT
cd /tmp
for c in smac3s5z humanoid17; do
  rm -rf /tmp/kt
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt -- python $R/bench.py --config $c --steps 3 --warmup 1 --cpu-cols 0 --instr-steps 0 --no-kernel-timing --no-other-configs > /dev/null 2>&1
  python $R/tools/prof_summary.py $(ls /tmp/kt/*/*kernel_trace.csv | head -1) --gaps 60 > $R/$O/kernel_trace_$c.md 2>&1
done
cd $R
timeout 300 python tools/prof_host.py > $O/prof_host.txt 2>&1
(timeout 1200 python -m pytest tests/test_gpu_parity.py -q -s -k "bench_configuration_onpolicy" > $O/t_onpolicy.txt 2>&1)
python - <<P
import json
d=json.loads(open("$O/bench_default.json").read().strip().split("\n")[-1])
print(round(d["ms_per_step"],3), d["value"], d["roofline"]["kernel"], d["roofline"]["frac"], d["roofline"].get("clock_ghz"), d["roofline"].get("matrix_pipe_frac"), d["roofline"].get("bf16x6_end_to_end"))
print(d["cpu_baseline"]["value"], d["cpu_baseline"]["sample"][:80])
for k,v in d["other_configs"].items(): print(k, v.get("ms_per_step"), (v.get("cpu_baseline") or {}).get("value"), (v.get("cpu_baseline") or {}).get("runs_s"), v.get("error"))
P
cat $O/bench_default.time; tail -2 $O/t_onpolicy.txt | cut -c1-300
