mkdir -p gpurun_out/r14 && export TMPDIR=/tmp
python -m pytest tests -m gpu -x -q -k "symmetric_fill or end_to_end or cfg2 or cfg5 or order_ceilings or golden" > gpurun_out/r14/t.log 2>&1; echo tests_rc=$? >> gpurun_out/r14/t.log; tail -12 gpurun_out/r14/t.log
python bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r14/b.json 2> gpurun_out/r14/b.err; echo bench_rc=$?; python - <<'PY'
import json
j=json.loads([l for l in open('gpurun_out/r14/b.json') if l.startswith('{')][-1])
print(j['value'], j['ms_per_step'], j['stage_ms_per_step'], j['roofline']['achieved'], j['roofline']['frac'])
PY
