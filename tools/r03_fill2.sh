#!/bin/bash
set -o pipefail
OUT=gpurun_out/r03c; mkdir -p $OUT; export TMPDIR=/tmp
timeout -k 10 300 python -m pytest tests -m gpu -x -q -k "symmetric_fill or cfg5 or cfg2" > $OUT/t_fill.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -3 $OUT/t_fill.log
[ $rc -ne 0 ] && exit $rc
for c in 3 5 4; do
  for dd in 0 1; do
    if [ $dd = 1 ]; then export BIEM_FILL_NO_DEDUPE=1; else unset BIEM_FILL_NO_DEDUPE; fi
    python bench.py --config $c --steps 4 --warmup 2 --no-cpu-baseline --sym-vs-lu-systems 0 > $OUT/bench_cfg${c}_nd$dd.json 2> $OUT/bench_cfg${c}_nd$dd.err; echo "bench cfg$c nodedupe=$dd rc=$?"
  done
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r03c/bench_*.json")):
    try:
        j=json.loads([l for l in open(f) if l.startswith("{")][-1])
        print(f.split("/")[-1], round(j["value"],1), "fill ms", round(j["stage_ms_per_step"]["fill"],2), "fill GB/s", round(j["fill"]["achieved"] or 0))
    except Exception as e: print(f, "ERR", e)
PY
export BIEM_HIPCC_FLAGS=-DBIEM_FILL_TRACE BIEM_SKIP_ISA_CHECK=1; python -m biem_helmholtz_sphere_amd._build > /dev/null 2>&1; BIEM_FILL_NO_DEDUPE=1 python tools/fill_trace.py 3; BIEM_FILL_NO_DEDUPE=1 python tools/fill_trace.py 5
