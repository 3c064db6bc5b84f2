# A/B of library variants on ONE box: bash tools/ab_lib.sh base nt nth ntm   (harl_amd/lib/libharl_<v>.so; base = libharl_hip.so)
cp harl_amd/lib/libharl_hip.so /tmp/base.so
for rep in 1 2 3; do
for v in "$@"; do
  if [ $v = base ]; then cp /tmp/base.so harl_amd/lib/libharl_hip.so; else cp harl_amd/lib/libharl_$v.so harl_amd/lib/libharl_hip.so; fi
  python bench.py --cpu-cols 0 --instr-steps 0 --steps 8 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$v', round(d['value']/1e6,3), round(d['ms_per_step'],3))"
done; done
cp /tmp/base.so harl_amd/lib/libharl_hip.so
