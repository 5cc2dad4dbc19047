mkdir -p gpurun_out/r9 && export TMPDIR=/tmp
python -m pytest tests -m gpu -x -q -k "symmetric_fill or sym_factor or end_to_end or cfg2 or cfg4 or cfg5 or batched" > gpurun_out/r9/t.log 2>&1; echo tests_rc=$? >> gpurun_out/r9/t.log; tail -12 gpurun_out/r9/t.log
python bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r9/b.json 2> gpurun_out/r9/b.err; echo bench_rc=$?; python - <<'PY'
import json
j=json.loads([l for l in open('gpurun_out/r9/b.json') if l.startswith('{')][-1])
print(j['value'], j['ms_per_step'], j['stage_ms_per_step'], j['roofline']['achieved'], j['roofline']['frac'])
PY
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r9/kt -o p -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/r9/kt.log 2>&1
python - <<'PY'
import csv
for r in list(csv.DictReader(open('gpurun_out/r9/kt/p_kernel_stats.csv')))[:16]:
    n=r['Name'].split('(')[0][:40]
    print(f"{n:42s} calls {r['Calls']:>5s} total {float(r['TotalDurationNs'])/1e6:9.2f} ms avg {float(r['AverageNs'])/1e3:9.1f} us")
PY
find gpurun_out/r9 -name "*.csv" -size +1M -delete; find gpurun_out/r9 -name "*.db" -delete
