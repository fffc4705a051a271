# PMC passes (separate runs per counter set) for the image gradient of image_warp and the image-pyramid kernel
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/pmc_iwim_*
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
           "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_WAVES SQ_INST_CYCLES_VMEM" \
           "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout -k 5 150 rocprofv3 --pmc $set --output-format csv -d $R/gpurun_out/pmc_iwim_$i -- python $R/tools/iwim_only.py 2 > $R/gpurun_out/pmc_iwim_$i.log 2>&1; echo "pass $i done"
done
cd $R && python tools/pmc_iwim_report.py gpurun_out/pmc_iwim_*/ > gpurun_out/r04_pmc_iwim.txt; cut -c1-260 gpurun_out/r04_pmc_iwim.txt
