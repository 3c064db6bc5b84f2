#!/bin/bash
set -x
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r04c9
mkdir -p $O
export TMPDIR=/tmp
run() { # name, env..., -- args
  name=$1; shift
  env "$@" timeout 600 python bench.py --cpu-cols 0 --no-other-configs $EXTRA > $O/$name.json 2> $O/$name.err
}
EXTRA="--config humanoid17 --steps 3 --warmup 2" run hum_base X=1
EXTRA="--config humanoid17 --steps 3 --warmup 2" run hum_notable HARL_UNFOLD_TABLE=0
EXTRA="--config humanoid17 --steps 3 --warmup 2" run hum_nwg256 HARL_NWG=256
EXTRA="--config smac3s5z --steps 5 --warmup 2" run smac_base X=1
EXTRA="--config smac3s5z --steps 5 --warmup 2" run smac_nwg256 HARL_NWG=256
EXTRA="--config cheetah6 --steps 5 --warmup 2" run cheetah_base X=1
EXTRA="--config cheetah6 --steps 5 --warmup 2" run cheetah_nwg256 HARL_NWG=256
EXTRA="--steps 10 --warmup 3" run mpe_base X=1
EXTRA="--steps 10 --warmup 3" run mpe_nwg256 HARL_NWG=256
timeout 2400 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.txt 2>&1
for f in hum_base hum_notable hum_nwg256 smac_base smac_nwg256 cheetah_base cheetah_nwg256 mpe_base mpe_nwg256; do python - <<P
import json
try:
    d=json.loads(open("$O/$f.json").read().strip().split("\n")[-1])
    ks=sorted(d["kernels"].items(), key=lambda kv:-kv[1]["total_ms"])[:9]
    print("$f", round(d["ms_per_step"],3), {k:(x["avg_ms"],x["n"]) for k,x in ks})
except Exception as e:
    print("$f ERR", e); print(open("$O/$f.err").read()[-800:])
P
done
tail -6 $O/pytest_gpu.txt
