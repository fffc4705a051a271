# Collects every profile artefact of a round on the GPU box into gpurun_out/profiles_<tag>/ :
#   bench line, rocprofv3 kernel stats of the same command, per-op breakdown, PMC counters of the
#   correlation kernels (SQ / TCC / GRBM in separate passes), HBM traffic (FETCH_SIZE and WRITE_SIZE in
#   separate passes), PMC counters of the loss kernels.   usage: bash tools/profile_round.sh r02
tag=${1:-rXX}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/profiles_$tag
mkdir -p $O
cd $R
timeout -k 10 500 python bench.py --steps 20 --warmup 5 > $O/${tag}_bench_line.json 2> $O/bench_line.err || { echo "bench failed"; tail -5 $O/bench_line.err; exit 1; }
echo "bench line done"
timeout -k 10 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --breakdown > /dev/null 2> $O/${tag}_breakdown.txt; echo "breakdown done"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_bench -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras > $O/prof_bench.log 2>&1
cp $(ls $O/prof_bench/*/*kernel_stats.csv | head -1) $O/${tag}_bench_kernel_stats.csv 2>/dev/null; echo "kernel stats done"
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
           "SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM" \
           "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum" \
           "GRBM_GUI_ACTIVE" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout -k 5 150 rocprofv3 --pmc $set --output-format csv -d $O/pmc_corr_$i -- python $R/tools/corr_only.py 32 "fwd bwd" 2 > $O/pmc_corr_$i.log 2>&1; echo "pmc corr pass $i done"
done
cd $R && python tools/pmc_summary.py $O/pmc_corr_*/ > $O/${tag}_pmc_corr_raw.txt
python tools/traffic_json.py $O $tag > $O/${tag}_traffic.json
python tools/pmc_corr_report.py $O/${tag}_pmc_corr_raw.txt > $O/${tag}_pmc_corr.txt 2>/dev/null
# the other single-GPU configs (informational lines), the f16 matrix-core correlation and the loss kernels
timeout -k 10 200 python bench.py --config cfg2 --steps 20 --warmup 5 > $O/${tag}_bench_line_cfg2.json 2> $O/cfg2.err; echo "cfg2 done"
timeout -k 10 200 python bench.py --config cfg5 --steps 20 --warmup 5 > $O/${tag}_bench_line_cfg5.json 2> $O/cfg5.err; echo "cfg5 done"
timeout -k 10 200 python tools/corr16_time.py > $O/${tag}_corr16_time.txt 2>&1; echo "corr16 timing done"
if [ -z "$SKIP_UNCHANGED" ]; then     # (SKIP_UNCHANGED=1: a round that did not touch the f16 correlation and the loss kernels)
  bash tools/pmc_corr16.sh fwd > $O/${tag}_pmc_corr16_fwd.txt 2>&1
  bash tools/pmc_corr16.sh bwd > $O/${tag}_pmc_corr16_bwd.txt 2>&1; echo "corr16 pmc done"
  bash tools/pmc_loss_step.sh > $O/pmc_loss_step.log 2>&1
  cp gpurun_out/r03_pmc_loss.txt $O/${tag}_pmc_loss.txt 2>/dev/null; cp gpurun_out/r03_pmc_loss.json $O/${tag}_pmc_loss.json 2>/dev/null; echo "loss pmc done"
fi
cat $O/${tag}_traffic.json | head -12
