set -x
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
CMD="python $R/tools/fused_ab.py --skip-oracle --only-full --no-mid --reps 2"
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_WAIT_ANY --kernel-trace --output-format csv -d /tmp/pmcA -- $CMD > /dev/null 2>&1
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_INSTS_MFMA SQ_WAIT_INST_LDS SQ_ACTIVE_INST_MISC --kernel-trace --output-format csv -d /tmp/pmcB -- $CMD > /dev/null 2>&1
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_FLAT SQ_INST_CYCLES_VALU SQ_VALU_MFMA_COEXEC_CYCLES --kernel-trace --output-format csv -d /tmp/pmcC -- $CMD > /dev/null 2>&1
python $R/tools/pmc_raw.py /tmp/pmcA/*/*_counter_collection.csv /tmp/pmcB/*/*_counter_collection.csv /tmp/pmcC/*/*_counter_collection.csv k_upd k_fwd_fused2x k_actor_head k_bwd_dx k_dw_tr > $R/gpurun_out/pmc_fused2.txt 2>&1
