# PMC passes (separate runs, counters only) for the loss kernels of the bench step -> gpurun_out/pmc_loss_step_*/
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
           "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_WAVES SQ_WAIT_ANY" \
           "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout -k 5 150 rocprofv3 --pmc $set --output-format csv -d $R/gpurun_out/pmc_loss_step_$i -- python $R/tools/loss_step_only.py 32 2 > $R/gpurun_out/pmc_loss_step_$i.log 2>&1; echo "pass $i done"
done
cd $R && python tools/pmc_loss_report.py gpurun_out/pmc_loss_step_*/ > gpurun_out/r03_pmc_loss.txt; cat gpurun_out/r03_pmc_loss.txt
