#!/bin/bash
set -o pipefail
OUT=gpurun_out/r03g; mkdir -p $OUT; export TMPDIR=/tmp
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > $OUT/t_full.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -3 $OUT/t_full.log
for c in 3 5 4 2; do
  for f in fused split; do
    if [ $f = split ]; then export BIEM_STRIP_FORM=split; else unset BIEM_STRIP_FORM; fi
    python bench.py --config $c --steps 4 --warmup 2 --no-cpu-baseline --sym-vs-lu-systems 0 > $OUT/bench_cfg${c}_$f.json 2> $OUT/bench_cfg${c}_$f.err; echo "bench cfg$c $f rc=$?"
  done
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r03g/bench_*.json")):
    try:
        j=json.loads([l for l in open(f) if l.startswith("{")][-1])
        st=j["stage_ms_per_step"]
        print(f.split("/")[-1], round(j["value"],1), "ms/step", round(j["ms_per_step"],1), "fill", round(st["fill"],2), "panel", round(st["panel"],1), "gemm_small", round(st["gemm_small"],1), "gemm", round(st["gemm"],1), "back", round(st["back"],1), "single", j["single_system_ms"])
    except Exception as e: print(f, "ERR", e)
PY
