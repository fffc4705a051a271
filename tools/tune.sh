# Runs bench.py --breakdown once per setting of the experiment knobs and prints one line each.
# usage: tools/tune.sh "FWD=1" "BWD=80 CENSUS=1" ...   ("base" = no knob)
mkdir -p gpurun_out
for v in "$@"; do
  envs=""
  for kv in $v; do [ "$kv" != base ] && envs="$envs UNFLOW_TUNE_$kv"; done
  env $envs timeout -k 10 200 python bench.py --steps 5 --warmup 2 --batch 32 --no-cpu-baseline --breakdown > gpurun_out/tune_tmp.log 2>&1
  python - "$v" <<'PY'
import json, sys, re
tag = sys.argv[1]
ops = {}
line = None
for l in open('gpurun_out/tune_tmp.log'):
    if l.startswith('{'):
        line = json.loads(l)
    m = re.match(r'\s*(\S.*?)\s+([0-9.]+) ms', l)
    if m: ops[m.group(1).strip()] = float(m.group(2))
if line is None:
    print(tag, 'FAILED'); print(open('gpurun_out/tune_tmp.log').read()[-1500:]); sys.exit(0)
r = line['roofline']
print('%-22s step %.3f ms | corr_fwd %.3f | corr_bwd %.3f | tern fwd %s bwd %s' % (
    tag, line['ms_per_step'], r['corr_fwd']['avg_launch_ms'], r['avg_launch_ms'],
    ops.get('ternary fwd 7x7'), ops.get('ternary bwd 7x7')))
PY
done
