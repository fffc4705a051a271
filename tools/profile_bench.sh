# Full bench line + rocprofv3 kernel-trace stats of the same command (for profiles/).
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd $R && timeout -k 10 500 python bench.py > gpurun_out/bench_line.json 2> gpurun_out/bench_line.err; tail -c 2500 gpurun_out/bench_line.json
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_bench -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $R/gpurun_out/prof_bench.log 2>&1
cd $R && python - <<'PY'
import csv, glob
for f in glob.glob('gpurun_out/prof_bench/**/*kernel_stats.csv', recursive=True):
    rows = list(csv.DictReader(open(f)))
    print(f)
    for r in rows[:22]:
        print('%-70s calls %5s avg_us %10.1f  pct %5s' % (r['Name'][:70], r['Calls'], float(r['AverageNs'])/1e3, r['Percentage']))
PY
