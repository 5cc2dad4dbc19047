mkdir -p gpurun_out/r7 && export TMPDIR=/tmp
python -m pytest tests -m gpu -x -q -k "sym_factor or symmetric_fill or end_to_end or factor_once or ldlt or cfg2 or cfg4 or golden or batched or growth or complex_wavenumber or factor_now" > gpurun_out/r7/t.log 2>&1; echo tests_rc=$? >> gpurun_out/r7/t.log; tail -12 gpurun_out/r7/t.log
python bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r7/b.json 2> gpurun_out/r7/b.err; echo bench_rc=$?; python - <<'PY'
import json
j=json.loads([l for l in open('gpurun_out/r7/b.json') if l.startswith('{')][-1])
print(j['value'], j['ms_per_step'], j['stage_ms_per_step'], j['roofline']['achieved'], j['roofline']['frac'])
PY
