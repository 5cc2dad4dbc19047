#!/bin/bash
# full GPU suite + bench lines of all configurations without the CPU baseline (a quick check of a build on one box)
O=gpurun_out/r03_eval; mkdir -p $O
timeout -k 10 1000 python -m pytest tests -m gpu -q > $O/t_full.log 2>&1; echo "tests exit $?"; tail -3 $O/t_full.log
for cfg in 1 2 4 5; do
  timeout -k 10 300 python bench.py --config $cfg --no-cpu-baseline > $O/bench_cfg$cfg.json 2> $O/bench_cfg$cfg.err || { echo "bench $cfg failed"; tail -3 $O/bench_cfg$cfg.err; exit 1; }
  python - <<PY
import json
d=json.loads(open("$O/bench_cfg$cfg.json").read().strip().splitlines()[-1])
print("cfg $cfg:", d["value"], d["unit"], "ms/step", d["ms_per_step"], "single_system_ms", d.get("single_system_ms"))
PY
done
timeout -k 10 600 python bench.py --no-cpu-baseline > $O/bench_cfg3.json 2> $O/bench_cfg3.err || { echo "bench 3 failed"; exit 1; }
python - <<PY
import json
d=json.loads(open("$O/bench_cfg3.json").read().strip().splitlines()[-1])
print("cfg 3:", d["value"], d["unit"], "ms/step", d["ms_per_step"], "roofline", d["roofline"]["frac"], "stages", d.get("stage_ms_per_step"))
PY
