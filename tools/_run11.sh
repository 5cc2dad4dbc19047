mkdir -p gpurun_out/r11 && export TMPDIR=/tmp
python -m pytest tests -m gpu -x -q -k "symmetric_fill or end_to_end or cfg2" > gpurun_out/r11/t.log 2>&1; echo tests_rc=$? >> gpurun_out/r11/t.log; tail -5 gpurun_out/r11/t.log
for v in "" "BIEM_ABL_FILL_NOSTORE=1"; do
env $v python bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/r11/b.json 2> gpurun_out/r11/b.err; echo "$v bench_rc=$?"; python - <<'PY'
import json
j=json.loads([l for l in open('gpurun_out/r11/b.json') if l.startswith('{')][-1])
print(j['value'], j['ms_per_step'], j['stage_ms_per_step']['fill'])
PY
done
