# usage: bash tools/pmc_fwd2.sh [tag] ; kernel-trace stats + PMC passes of the correlation forward alone (tools/corr_only.py)
tag=${1:-fwd2}
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout -k 5 120 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/kt_$tag -- python $R/tools/corr_only.py 32 fwd 20 > $R/gpurun_out/kt_$tag.log 2>&1
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
           "SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM" \
           "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum" \
           "GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout -k 5 120 rocprofv3 --pmc $set --output-format csv -d $R/gpurun_out/pmc_${tag}_$i -- python $R/tools/corr_only.py 32 fwd 3 > $R/gpurun_out/pmc_${tag}_$i.log 2>&1; echo "pass $i done"
done
cd $R && python - <<PY
import csv, glob
for f in glob.glob('gpurun_out/kt_$tag/**/*kernel_stats.csv', recursive=True):
    for r in list(csv.DictReader(open(f)))[:4]:
        print('%-70s calls %5s avg_us %10.1f  pct %5s' % (r['Name'][:70], r['Calls'], float(r['AverageNs'])/1e3, r['Percentage']))
PY
python tools/pmc_summary.py gpurun_out/pmc_${tag}_*/
