# usage: bash tools/pmc_variant.sh <name> [<so>] : PMC passes of the correlation backward of one build (diagnostic variants allowed)
name=$1; so=$2
R=$GRAFT_REPO_ROOT
export UNFLOW_ALLOW_DIAGNOSTIC_BUILD=1
if [ -n "$so" ]; then export UNFLOW_HIP_LIB=$R/$so; fi
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
           "SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC" \
           "SQ_LDS_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_VALU_MFMA_COEXEC_CYCLES SQ_INST_CYCLES_VALU SQ_INST_CYCLES_SALU SQ_IFETCH" \
           "GRBM_GUI_ACTIVE SQ_INSTS_BRANCH SQ_INST_LEVEL_LDS SQ_IFETCH_LEVEL SQ_INSTS_MFMA"; do
  i=$((i+1))
  timeout -k 5 120 rocprofv3 --pmc $set --output-format csv -d $R/gpurun_out/pmcv_${name}_$i -- python $R/tools/corr_only.py 32 bwd 2 > $R/gpurun_out/pmcv_${name}_$i.log 2>&1; echo "$name pass $i done"
done
cd $R && python tools/pmc_summary.py gpurun_out/pmcv_${name}_*/ > gpurun_out/pmcv_${name}.txt
