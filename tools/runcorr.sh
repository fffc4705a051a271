mkdir -p gpurun_out
timeout -k 10 300 python -m pytest tests -m gpu -q -k "correlation" > gpurun_out/pytest_corr.log 2>&1; tail -3 gpurun_out/pytest_corr.log
timeout -k 10 200 python bench.py --steps 3 --warmup 1 --batch 32 --no-cpu-baseline > gpurun_out/bench_tmp.log 2>&1
python - <<'PY'
import json
for l in open('gpurun_out/bench_tmp.log'):
    if l.startswith('{'):
        j=json.loads(l); r=j['roofline']
        print('pairs/s %.1f  ms/step %.2f | corr_fwd %.3f ms %.1f TF | corr_bwd %.3f ms %.1f TF' % (j['value'], j['ms_per_step'], r['corr_fwd']['avg_launch_ms'], r['corr_fwd']['tflops'], r['avg_launch_ms'], r['achieved']))
    elif 'Error' in l or 'error' in l: print(l.strip())
PY
