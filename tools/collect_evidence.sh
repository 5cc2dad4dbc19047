#!/bin/bash
# Evidence run on the GPU box (gpurun): bench lines, rocprofv3 kernel stats and PMC passes; everything lands under gpurun_out/ev/.
# usage: bash tools/collect_evidence.sh [main|pmc|configs|extra]   (one gpurun call each: a call is limited to 20 minutes)
set -o pipefail
OUT=gpurun_out/ev; mkdir -p $OUT; export TMPDIR=/tmp
part=${1:-main}
if [ "$part" = main ]; then
  python bench.py --steps 10 --warmup 3 --single-system > $OUT/bench_cfg3.json 2> $OUT/bench_cfg3.err; echo "bench cfg3 rc=$?"
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -o p -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --sym-vs-lu-systems 0 --no-single-system > $OUT/bench_under_rocprof.json 2> $OUT/kt.err; echo "kernel trace rc=$?"
  BIEM_FILL_NO_DEDUPE=1 python bench.py --steps 5 --warmup 2 --no-cpu-baseline > $OUT/bench_cfg3_no_pair_classes.json 2> $OUT/bench_cfg3_nd.err; echo "bench cfg3 no pair classes rc=$?"
fi
if [ "$part" = pmc ]; then
  for c in "WRITE_SIZE" "FETCH_SIZE" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAVES" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
    n=$(echo $c | cut -d' ' -f1)
    rocprofv3 --pmc $c --output-format csv -d $OUT/pmc_$n -o p -- python3 bench.py --systems-per-gpu 8 --steps 1 --warmup 0 --no-cpu-baseline --sym-vs-lu-systems 0 --no-single-system > $OUT/pmc_$n.log 2>&1 && echo "pmc $n ok"
  done
  for n in WRITE_SIZE FETCH_SIZE SQ_WAVE_CYCLES SQ_INSTS_VALU; do echo "== pass $n (8 systems per launch, 1 step, pair classes on from 8 systems)"; python tools/pmc_summary.py $OUT/pmc_$n; done > $OUT/pmc_summary.txt 2>&1
  # the fill without pair classes (every ball pair contracted on its own): LDS counters
  BIEM_FILL_NO_DEDUPE=1 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS --output-format csv -d $OUT/pmc_nd -o p -- python3 bench.py --systems-per-gpu 8 --steps 1 --warmup 0 --no-cpu-baseline --sym-vs-lu-systems 0 --no-single-system > $OUT/pmc_nd.log 2>&1 && echo "pmc nodedupe ok"
  (echo "== LDS counters without pair classes (BIEM_FILL_NO_DEDUPE=1), 8 systems"; python tools/pmc_summary.py $OUT/pmc_nd k_fill) >> $OUT/pmc_summary.txt 2>&1
  for cfg in 5; do
    BIEM_FILL_NO_DEDUPE=1 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS --output-format csv -d $OUT/pmc_nd5 -o p -- python3 bench.py --config 5 --systems-per-gpu 16 --steps 1 --warmup 0 --no-cpu-baseline --sym-vs-lu-systems 0 --no-single-system > $OUT/pmc_nd5.log 2>&1 && echo "pmc cfg5 ok"
    (echo "== cfg 5 (4-D), 16 systems, no pair classes"; python tools/pmc_summary.py $OUT/pmc_nd5 k_fill) >> $OUT/pmc_summary.txt 2>&1
  done
fi
if [ "$part" = configs ]; then
  for c in 1 2 4 5; do timeout -k 10 420 python bench.py --config $c --steps 5 --warmup 2 --cpu-baseline-reps 1 > $OUT/bench_cfg$c.json 2> $OUT/bench_cfg$c.err; echo "bench cfg$c rc=$?"; done
  for c in 4 5; do BIEM_FILL_NO_DEDUPE=1 python bench.py --config $c --steps 5 --warmup 2 --no-cpu-baseline > $OUT/bench_cfg${c}_no_pair_classes.json 2> $OUT/bench_cfg${c}_nd.err; echo "bench cfg$c no pair classes rc=$?"; done
fi
if [ "$part" = extra ]; then
  BIEM_SOLVER=lu python bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/bench_cfg3_lu.json 2> $OUT/bench_cfg3_lu.err; echo "lu rc=$?"
  BIEM_BENCH_SHARE_GPU=1 python bench.py --gpus 2 --steps 1 --warmup 1 --systems-per-gpu 32 --no-cpu-baseline > $OUT/bench_2ranks_weak_shared_gpu.json 2> $OUT/b2w.err; echo "2-rank weak rc=$?"
  BIEM_BENCH_SHARE_GPU=1 python bench.py --gpus 2 --scaling strong --steps 1 --warmup 1 --systems-per-gpu 64 --no-cpu-baseline > $OUT/bench_2ranks_strong_shared_gpu.json 2> $OUT/b2s.err; echo "2-rank strong rc=$?"
  BIEM_BENCH_FORCE_DIST=1 python bench.py --gpus 1 --steps 2 --warmup 1 --no-cpu-baseline > $OUT/bench_rccl_1rank.json 2> $OUT/b1r.err; echo "rccl 1 rank rc=$?"
  BIEM_BENCH_FORCE_DIST=1 python bench.py --gpus 1 --config 5 --steps 2 --warmup 1 --no-cpu-baseline > $OUT/bench_rccl_1rank_cfg5.json 2> $OUT/b1r5.err; echo "rccl 1 rank cfg5 rc=$?"
  for s in 32 64 128; do python bench.py --systems-per-gpu $s --steps 4 --warmup 2 --no-cpu-baseline --sym-vs-lu-systems 0 --no-single-system > $OUT/bench_cfg3_${s}sys.json 2> $OUT/b$s.err; echo "cfg3 $s systems rc=$?"; done
fi
find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*.db" -delete; find $OUT -name "*counter_collection.csv" -size +3M -delete
ls $OUT
