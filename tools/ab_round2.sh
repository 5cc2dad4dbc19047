#!/bin/bash
# A/B against the round-2 code on ONE box: checks out 2b3cde2 into a scratch directory, builds it, runs the same bench lines
# (first, here: git archive 2b3cde2 -o tools/_r2src.tar - the archive is git-ignored and travels with the gpurun snapshot)
set -o pipefail
OUT=gpurun_out/ab; mkdir -p $OUT
rm -rf /tmp/r2 && mkdir -p /tmp/r2 && tar -C /tmp/r2 -xf tools/_r2src.tar
(cd /tmp/r2 && python -m biem_helmholtz_sphere_amd._build > /dev/null 2>&1)
for c in 1 2; do
  python bench.py --config $c --steps 10 --warmup 3 --no-cpu-baseline --sym-vs-lu-systems 0 > $OUT/new_cfg$c.json 2>/dev/null
  (cd /tmp/r2 && python bench.py --config $c --steps 10 --warmup 3 --no-cpu-baseline > $OLDPWD/$OUT/old_cfg$c.json 2>/dev/null)
  python bench.py --config $c --steps 10 --warmup 3 --no-cpu-baseline --sym-vs-lu-systems 0 > $OUT/new2_cfg$c.json 2>/dev/null
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/ab/*.json")):
    try:
        j=json.loads([l for l in open(f) if l.startswith("{")][-1])
        print(f.split("/")[-1], round(j["value"],1), "ms/step", round(j["ms_per_step"],3), "single", j["single_system_ms"], {k:round(v,2) for k,v in j["stage_ms_per_step"].items() if v>0.01})
    except Exception as e: print(f,"ERR",e)
PY
