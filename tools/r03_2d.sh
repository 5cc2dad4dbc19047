#!/bin/bash
set -o pipefail
OUT=gpurun_out/r03e; mkdir -p $OUT; export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests -m gpu -x -q -k "toeplitz or symmetric_fill or order_ceilings or fill_reference or cfg4 or golden_rows or close_spheres or matrix_attribute" > $OUT/t_2d.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -4 $OUT/t_2d.log
for dd in 0 1; do
  if [ $dd = 1 ]; then export BIEM_FILL_NO_DEDUPE=1; else unset BIEM_FILL_NO_DEDUPE; fi
  python bench.py --config 4 --steps 5 --warmup 2 --no-cpu-baseline --sym-vs-lu-systems 0 > $OUT/bench_cfg4_nd$dd.json 2> $OUT/bench_cfg4_nd$dd.err; echo "bench cfg4 nodedupe=$dd rc=$?"
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r03e/bench_*.json")):
    try:
        j=json.loads([l for l in open(f) if l.startswith("{")][-1])
        print(f.split("/")[-1], round(j["value"],1), "fill ms", round(j["stage_ms_per_step"]["fill"],2), "fill GB/s", round(j["fill"]["achieved"] or 0), "single", j["single_system_ms"])
    except Exception as e: print(f, "ERR", e)
PY
