cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for M in fwd wgrad; do
 for C in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS" "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VALU" "SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_VMEM"; do
  T=$(echo $C | tr ' ' '_' | cut -c1-40)
  timeout -k 10 120 rocprofv3 --pmc $C --kernel-trace -d $R/gpurun_out/pmc2/$M/$T -o out -- python3 $R/tools/run_one_conv.py l1 $M 5 > $R/gpurun_out/pmc2/$M.$T.log 2>&1 || exit 1
 done
done
