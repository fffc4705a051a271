mkdir -p gpurun_out
for d in ${DBGS:-0 1 2 3}; do
UNFLOW_CORR_DBG=$d timeout -k 10 200 python bench.py --steps 3 --warmup 1 --batch 32 --no-cpu-baseline > gpurun_out/bench_tmp.log 2>&1
python - <<PY
import json
for l in open('gpurun_out/bench_tmp.log'):
    if l.startswith('{'):
        j=json.loads(l); r=j['roofline']
        print('dbg=$d corr_fwd %.3f ms | corr_bwd %.3f ms %.1f TF' % (r['corr_fwd']['avg_launch_ms'], r['avg_launch_ms'], r['achieved']))
PY
done
