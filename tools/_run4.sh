mkdir -p gpurun_out/r4 && export TMPDIR=/tmp
python -m pytest tests -m gpu -x -q -k "symmetric_fill or end_to_end or factor_once or ldlt or cfg2 or cfg4 or golden or batched" > gpurun_out/r4/t.log 2>&1; echo tests_rc=$? >> gpurun_out/r4/t.log; tail -5 gpurun_out/r4/t.log
python bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r4/b.json 2> gpurun_out/r4/b.err; echo bench_rc=$?; python - <<'PY'
import json
j=json.loads([l for l in open('gpurun_out/r4/b.json') if l.startswith('{')][-1])
print(j['value'], j['ms_per_step'], j['stage_ms_per_step'], j['fill']['achieved'])
PY
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r4/kt -o p -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/r4/kt.log 2>&1; head -25 gpurun_out/r4/kt/p_kernel_stats.csv
find gpurun_out/r4 -name "*.csv" -size +1M -delete; find gpurun_out/r4 -name "*.db" -delete
