mkdir -p gpurun_out/r6 && export TMPDIR=/tmp
python bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r6/b.json 2> gpurun_out/r6/b.err; echo bench_rc=$?; python - <<'PY'
import json
j=json.loads([l for l in open('gpurun_out/r6/b.json') if l.startswith('{')][-1])
print(j['value'], j['ms_per_step'], j['stage_ms_per_step'], j['roofline']['achieved'], j['roofline']['frac'])
PY
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r6/kt -o p -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/r6/kt.log 2>&1; cut -c1-60,380- gpurun_out/r6/kt/p_kernel_stats.csv | head -22
python -m pytest tests -m gpu -x -q > gpurun_out/r6/t.log 2>&1; echo tests_rc=$? >> gpurun_out/r6/t.log; tail -4 gpurun_out/r6/t.log
find gpurun_out/r6 -name "*.csv" -size +1M -delete; find gpurun_out/r6 -name "*.db" -delete
