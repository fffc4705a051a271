# PMC passes for the loss kernels (census pair kernels, image_warp, forward_warp): instruction mix and busy cycles
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
           "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_WAVES SQ_INST_CYCLES_VMEM" \
           "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout -k 5 150 rocprofv3 --pmc $set --output-format csv -d $R/gpurun_out/pmc_loss_$i -- python $R/tools/loss_only.py 32 2 > $R/gpurun_out/pmc_loss_$i.log 2>&1; echo "pass $i done"
done
cd $R && python tools/pmc_summary.py gpurun_out/pmc_loss_*/ > gpurun_out/pmc_loss_summary.txt; tail -80 gpurun_out/pmc_loss_summary.txt
