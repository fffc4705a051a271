# HBM traffic of the correlation kernels from PMC counters (separate passes, MI355X_MICROARCH.md "HBM")
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 5 120 rocprofv3 --pmc $c --output-format csv -d $R/gpurun_out/pmc_traffic_$c -- python $R/tools/corr_only.py 32 "fwd bwd" 2 > $R/gpurun_out/pmc_traffic_$c.log 2>&1; echo "pass $c done"
done
cd $R && python - <<'PY'
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/pmc_traffic_*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"]
        if "corr" not in k: continue
        key = "fwd" if "fwd" in k else ("bwd_wrt1" if "Lb1E" in k else "bwd_wrt0")
        acc[key][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k, cs in sorted(acc.items()):
    print(k, {c: sum(v)/len(v) for c, v in cs.items()})
PY
