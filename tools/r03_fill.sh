#!/bin/bash
# the symmetric fill: parity of the forms, then bench lines of cfg 3 / 4 / 5 with and without pair classes
set -o pipefail
OUT=gpurun_out/r03b; mkdir -p $OUT; export TMPDIR=/tmp
timeout -k 10 600 python -m pytest tests -m gpu -x -q -k "symmetric_fill or end_to_end or golden_rows or cfg5 or cfg2 or cfg4 or complex_wavenumber or pair_classes or dedupe" > $OUT/t_fill.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -3 $OUT/t_fill.log
[ $rc -ne 0 ] && exit $rc
for c in 3 5 4; do
  for dd in 0 1; do
    if [ $dd = 1 ]; then export BIEM_FILL_NO_DEDUPE=1; else unset BIEM_FILL_NO_DEDUPE; fi
    python bench.py --config $c --steps 4 --warmup 2 --no-cpu-baseline --sym-vs-lu-systems 0 > $OUT/bench_cfg${c}_nd$dd.json 2> $OUT/bench_cfg${c}_nd$dd.err; echo "bench cfg$c nodedupe=$dd rc=$?"
    BIEM_FILL_FORM=entry python bench.py --config $c --steps 4 --warmup 2 --no-cpu-baseline --sym-vs-lu-systems 0 > $OUT/bench_cfg${c}_nd${dd}_old.json 2> $OUT/bench_cfg${c}_nd${dd}_old.err; echo "bench cfg$c old form nodedupe=$dd rc=$?"
  done
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r03b/bench_*.json")):
    try:
        j=json.loads([l for l in open(f) if l.startswith("{")][-1])
        print(f.split("/")[-1], round(j["value"],1), "fill ms", round(j["stage_ms_per_step"]["fill"],2), "fill GB/s", round(j["fill"]["achieved"] or 0), "relerr", j["max_rel_err_uscat"])
    except Exception as e: print(f, "ERR", e)
PY
