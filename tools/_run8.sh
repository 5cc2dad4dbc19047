mkdir -p gpurun_out/r8 && export TMPDIR=/tmp
for c in 1 2 4 5; do
  timeout -k 10 420 python bench.py --config $c --steps 3 --warmup 1 > gpurun_out/r8/b_cfg$c.json 2> gpurun_out/r8/b_cfg$c.err; echo cfg$c rc=$?
  python - <<PY
import json
try:
    j=json.loads([l for l in open('gpurun_out/r8/b_cfg$c.json') if l.startswith('{')][-1])
    print($c, j['value'], j['ms_per_step'], j['single_system_ms'], j['max_rel_err_uscat'], j['stage_ms_per_step'], j['cpu_baseline'])
except Exception as e: print('parse fail', e)
PY
done
