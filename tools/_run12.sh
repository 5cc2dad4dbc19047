mkdir -p gpurun_out/r12 && export TMPDIR=/tmp
for v in "BIEM_FILL_GY=1" "BIEM_FILL_GY=2" "BIEM_FILL_GY=4" "BIEM_FILL_GY=2 BIEM_ABL_FILL_NOSTORE=1" "BIEM_FILL_GY=1 BIEM_ABL_FILL_NOSTORE=1"; do
env $v python bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/r12/b.json 2> gpurun_out/r12/b.err; echo "$v bench_rc=$?"; python - <<'PY'
import json
j=json.loads([l for l in open('gpurun_out/r12/b.json') if l.startswith('{')][-1])
print(j['value'], j['ms_per_step'], j['stage_ms_per_step']['fill'])
PY
done
