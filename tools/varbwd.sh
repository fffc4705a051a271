mkdir -p gpurun_out
for v in ${VARS:-0 1 2 3 4 5 6 7}; do
UNFLOW_BWD_VARIANT=$v timeout -k 10 200 python bench.py --steps 3 --warmup 1 --batch 32 --no-cpu-baseline > gpurun_out/bench_tmp.log 2>&1
python - <<PY
import json
for l in open('gpurun_out/bench_tmp.log'):
    if l.startswith('{'):
        j=json.loads(l); r=j['roofline']
        print('variant=$v corr_bwd %.3f ms %.1f TF' % (r['avg_launch_ms'], r['achieved']))
PY
done
