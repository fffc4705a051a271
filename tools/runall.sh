mkdir -p gpurun_out
timeout -k 10 400 python -m pytest tests -m gpu -q > gpurun_out/pytest_gpu.log 2>&1; tail -4 gpurun_out/pytest_gpu.log | grep -E "passed|failed"
grep -E "^FAILED" gpurun_out/pytest_gpu.log | head
timeout -k 10 300 python bench.py --steps 3 --warmup 1 --batch ${BATCH:-32} --no-cpu-baseline --breakdown > gpurun_out/bench_tmp.log 2>&1
python - <<'PY'
import json
for l in open('gpurun_out/bench_tmp.log'):
    if l.startswith('{'):
        j=json.loads(l); r=j['roofline']
        print('pairs/s %.1f  ms/step %.2f | corr_fwd %.3f ms %.1f TF | corr_bwd %.3f ms %.1f TF' % (j['value'], j['ms_per_step'], r['corr_fwd']['avg_launch_ms'], r['corr_fwd']['tflops'], r['avg_launch_ms'], r['achieved']))
    elif 'GB/s' in l or 'rror' in l: print(l.rstrip())
PY
