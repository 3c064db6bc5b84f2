#!/bin/bash
# round 6, session 3, call b: packed fp32 VALU (v_pk_*) beside the MFMAs -- A/B of library variants on one box
#   base | nopk (no packed fp32 anywhere + no SLP) | nopksplit (only the operand splits unpacked) | noslp
VARIANTS=${VARIANTS:-"hip nopk nopksplit noslp"}
KVARIANTS=${KVARIANTS:-"hip nopk"}
TAG=${TAG:-pk}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r06s3
mkdir -p $O
cd $R
ab() {  # config steps
for rep in 1 2 3; do for v in $VARIANTS; do
HARL_LIB=$v timeout 600 python bench.py --config $1 --steps $2 --warmup 3 --cpu-cols 0 --instr-steps 0 --no-kernel-timing --no-other-configs 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1 $v', round(d['ms_per_step'],3))"
done; done
}
ab mpe 20 | tee $O/ab_${TAG}_mpe.txt
ab cheetah6 8 | tee $O/ab_${TAG}_cheetah6.txt
ab smac3s5z 10 | tee $O/ab_${TAG}_smac.txt
for v in $KVARIANTS; do
HARL_LIB=$v timeout 600 python bench.py --steps 10 --warmup 3 --cpu-cols 0 --instr-steps 2 --no-other-configs 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', round(d['ms_per_step'],3)); [print('   %-20s n %3d avg %.4f'%(k,x['n'],x['avg_ms'])) for k,x in d['kernels'].items()]"
done | tee $O/ab_${TAG}_kernels.txt
