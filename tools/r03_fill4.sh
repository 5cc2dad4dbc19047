#!/bin/bash
OUT=gpurun_out/r03k; mkdir -p $OUT
timeout -k 10 300 python -m pytest tests -m gpu -x -q -k "symmetric_fill or cfg5 or cfg2 or pair_classes" > $OUT/t.log 2>&1; echo "tests rc=$?"; tail -2 $OUT/t.log
for c in 3 5 2; do for dd in 0 1; do
  if [ $dd = 1 ]; then export BIEM_FILL_NO_DEDUPE=1; else unset BIEM_FILL_NO_DEDUPE; fi
  python bench.py --config $c --steps 4 --warmup 2 --no-cpu-baseline --sym-vs-lu-systems 0 > $OUT/b_cfg${c}_nd$dd.json 2>/dev/null
done; done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r03k/b_*.json")):
    j=json.loads([l for l in open(f) if l.startswith("{")][-1]); print(f.split("/")[-1], round(j["value"],1), "fill ms", round(j["stage_ms_per_step"]["fill"],2), round(j["fill"]["frac"],3))
PY
