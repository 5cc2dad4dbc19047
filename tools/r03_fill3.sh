#!/bin/bash
set -o pipefail
OUT=gpurun_out/r03j; mkdir -p $OUT; export TMPDIR=/tmp
for w in 16 8; do for nc in 2 1; do
  export BIEM_FILL_RED_WAVES=$w BIEM_FILL_NC=$nc
  for c in 3 5; do
    for dd in 0 1; do
      if [ $dd = 1 ]; then export BIEM_FILL_NO_DEDUPE=1; else unset BIEM_FILL_NO_DEDUPE; fi
      python bench.py --config $c --steps 3 --warmup 1 --no-cpu-baseline --sym-vs-lu-systems 0 > $OUT/b_w${w}_nc${nc}_cfg${c}_nd$dd.json 2> $OUT/b.err; echo "w=$w nc=$nc cfg$c nd=$dd rc=$?"
    done
  done
done; done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r03j/b_*.json")):
    try:
        j=json.loads([l for l in open(f) if l.startswith("{")][-1])
        print(f.split("/")[-1], round(j["value"],1), "fill ms", round(j["stage_ms_per_step"]["fill"],2))
    except Exception as e: print(f, "ERR", e)
PY
