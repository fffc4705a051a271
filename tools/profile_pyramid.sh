# rocprofv3 kernel stats of the 5-level KITTI pyramid loss (eager: the same kernels a graph replays)
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_pyr -- python $R/tools/graph_pyramid.py > $R/gpurun_out/prof_pyr.log 2>&1
cd $R && python - <<'PY'
import csv, glob
for f in glob.glob('gpurun_out/prof_pyr/**/*kernel_stats.csv', recursive=True):
    rows = list(csv.DictReader(open(f)))
    tot = sum(float(r['TotalDurationNs']) for r in rows)
    calls = sum(int(r['Calls']) for r in rows)
    print(f, 'kernels', len(rows), 'calls', calls, 'total ms', tot / 1e6)
    for r in rows[:45]:
        print('%-90s calls %5s avg_us %8.1f  pct %5s' % (r['Name'][:90], r['Calls'], float(r['AverageNs'])/1e3, r['Percentage']))
PY
tail -5 $R/gpurun_out/prof_pyr.log
