#!/bin/bash
set -o pipefail
OUT=gpurun_out/r03f; mkdir -p $OUT; export TMPDIR=/tmp
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > $OUT/t_full.log 2>&1; echo "tests rc=$?"; tail -3 $OUT/t_full.log
for c in 4 5 3; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt$c -o p -- python3 bench.py --config $c --steps 2 --warmup 1 --no-cpu-baseline --sym-vs-lu-systems 0 > $OUT/kt$c.json 2> $OUT/kt$c.err; echo "kt cfg$c rc=$?"
  python - <<PY
import csv
rows=list(csv.DictReader(open("$OUT/kt$c/p_kernel_stats.csv")))
print("cfg $c")
for r in rows[:16]: print("  %-60s calls %6s total ms %9.2f avg us %9.1f  %s%%" % (r["Name"].split("(")[0][:60], r["Calls"], float(r["TotalDurationNs"])/1e6, float(r["AverageNs"])/1e3, r["Percentage"]))
PY
done
find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*.db" -delete
