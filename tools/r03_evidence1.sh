#!/bin/bash
# full GPU suite, then PMC passes: LDS counters of the fill (8 systems, no pair classes) and HBM-side traffic at the timed launch shape (256 systems)
set -o pipefail
OUT=gpurun_out/r03d; mkdir -p $OUT; export TMPDIR=/tmp
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > $OUT/t_full.log 2>&1; echo "tests rc=$?"; tail -3 $OUT/t_full.log
for c in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAVES"; do
  n=$(echo $c | cut -d' ' -f1)
  BIEM_FILL_NO_DEDUPE=1 rocprofv3 --pmc $c --output-format csv -d $OUT/pmc8_$n -o p -- python3 bench.py --systems-per-gpu 8 --steps 1 --warmup 0 --no-cpu-baseline --sym-vs-lu-systems 0 > $OUT/pmc8_$n.log 2>&1 && echo "pmc8 $n ok"
  python tools/pmc_summary.py $OUT/pmc8_$n > $OUT/pmc8_$n.txt 2>&1
done
for n in WRITE_SIZE FETCH_SIZE; do
  rocprofv3 --pmc $n --output-format csv -d $OUT/pmc256_$n -o p -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --sym-vs-lu-systems 0 > $OUT/pmc256_$n.log 2>&1 && echo "pmc256 $n ok"
  python tools/pmc_summary.py $OUT/pmc256_$n > $OUT/pmc256_$n.txt 2>&1
done
find $OUT -name "*.db" -delete; find $OUT -name "*counter_collection.csv" -size +3M -delete; find $OUT -name "*kernel_trace.csv" -delete
grep -A8 "k_fill_red" $OUT/pmc8_SQ_WAVE_CYCLES.txt | head -12
