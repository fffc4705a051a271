R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_bd -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --breakdown > $R/gpurun_out/prof_bd.log 2>&1
cd $R && python - <<'PY'
import csv, glob
for f in glob.glob('gpurun_out/prof_bd/**/*kernel_stats.csv', recursive=True):
    rows = list(csv.DictReader(open(f)))
    for r in rows[:40]:
        if 'unflow' in r['Name']:
            print('%-64s calls %5s avg_us %10.1f  pct %5s' % (r['Name'].replace('unflow::(anonymous namespace)::','')[:64], r['Calls'], float(r['AverageNs'])/1e3, r['Percentage']))
PY
grep -E "GB/s" $R/gpurun_out/prof_bd.log
