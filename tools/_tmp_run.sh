mkdir -p gpurun_out/r15 && export TMPDIR=/tmp
python -m pytest tests -m gpu -x -q -k "symmetric_fill or sym_factor or end_to_end or cfg2 or cfg4 or golden or fill_reference or factor_once or ldlt or batched or complex_wave" > gpurun_out/r15/t.log 2>&1; echo tests_rc=$? >> gpurun_out/r15/t.log; tail -12 gpurun_out/r15/t.log
python bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r15/b.json 2> gpurun_out/r15/b.err; echo bench_rc=$?; python - <<'PY'
import json
j=json.loads([l for l in open('gpurun_out/r15/b.json') if l.startswith('{')][-1])
print(j['value'], j['ms_per_step'], j['stage_ms_per_step'], j['roofline']['achieved'], j['roofline']['frac'])
PY
