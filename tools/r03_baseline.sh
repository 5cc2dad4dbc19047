#!/bin/bash
# round-3 baseline on one box: GPU tests, bench lines (with and without pair classes), PMC traffic at the timed launch shape
set -o pipefail
OUT=gpurun_out/r03a; mkdir -p $OUT; export TMPDIR=/tmp
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > $OUT/t_full.log 2>&1; echo "tests rc=$?"; tail -3 $OUT/t_full.log
python bench.py --steps 5 --warmup 2 --no-cpu-baseline > $OUT/bench_cfg3.json 2> $OUT/bench_cfg3.err; echo "bench cfg3 rc=$?"
BIEM_FILL_NO_DEDUPE=1 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/bench_cfg3_nodedupe.json 2> $OUT/bench_cfg3_nodedupe.err; echo "bench cfg3 nodedupe rc=$?"
for c in 4 5; do
  python bench.py --config $c --steps 5 --warmup 2 --no-cpu-baseline > $OUT/bench_cfg$c.json 2> $OUT/bench_cfg$c.err; echo "bench cfg$c rc=$?"
  BIEM_FILL_NO_DEDUPE=1 python bench.py --config $c --steps 5 --warmup 2 --no-cpu-baseline > $OUT/bench_cfg${c}_nodedupe.json 2> $OUT/bench_cfg${c}_nodedupe.err; echo "bench cfg$c nodedupe rc=$?"
done
for n in WRITE_SIZE FETCH_SIZE; do
  rocprofv3 --pmc $n --output-format csv -d $OUT/pmc256_$n -o p -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline > $OUT/pmc256_$n.log 2>&1 && echo "pmc256 $n ok"
  python tools/pmc_summary.py $OUT/pmc256_$n > $OUT/pmc256_$n.txt 2>&1
done
find $OUT -name "*.db" -delete; find $OUT -name "*counter_collection.csv" -size +3M -delete; find $OUT -name "*kernel_trace.csv" -delete
ls $OUT
