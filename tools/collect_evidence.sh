#!/bin/bash
# Evidence run on the GPU box (gpurun): bench lines, rocprofv3 kernel stats and PMC passes; everything lands under gpurun_out/ev/.
# usage: bash tools/collect_evidence.sh [full]   ("full" adds the other configurations and the 5-repetition CPU baseline)
set -o pipefail
OUT=gpurun_out/ev; mkdir -p $OUT; export TMPDIR=/tmp
python bench.py --steps 10 --warmup 3 > $OUT/bench_cfg3.json 2> $OUT/bench_cfg3.err; echo "bench cfg3 rc=$?"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -o p -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > $OUT/bench_under_rocprof.json 2> $OUT/kt.err; echo "kernel trace rc=$?"
for c in "WRITE_SIZE" "FETCH_SIZE" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAVES" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
  n=$(echo $c | cut -d' ' -f1)
  rocprofv3 --pmc $c --output-format csv -d $OUT/pmc_$n -o p -- python3 bench.py --systems-per-gpu 8 --steps 1 --warmup 0 --no-cpu-baseline > $OUT/pmc_$n.log 2>&1 && echo "pmc $n ok"
done
for n in WRITE_SIZE FETCH_SIZE SQ_WAVE_CYCLES SQ_INSTS_VALU; do echo "== pass $n (8 systems per launch, 1 step)"; python tools/pmc_summary.py $OUT/pmc_$n; done > $OUT/pmc_summary.txt 2>&1
if [ "$1" = "full" ]; then
  for c in 1 2 4 5; do timeout -k 10 420 python bench.py --config $c --steps 5 --warmup 2 > $OUT/bench_cfg$c.json 2> $OUT/bench_cfg$c.err; echo "bench cfg$c rc=$?"; done
  timeout -k 10 600 python bench.py --steps 3 --warmup 1 --cpu-baseline-reps 5 > $OUT/bench_cfg3_cpu5.json 2> $OUT/bench_cfg3_cpu5.err; echo "cpu baseline protocol rc=$?"
  BIEM_SOLVER=lu python bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/bench_cfg3_lu.json 2> $OUT/bench_cfg3_lu.err; echo "lu rc=$?"
  BIEM_BENCH_SHARE_GPU=1 python bench.py --gpus 2 --steps 1 --warmup 1 --systems-per-gpu 32 --no-cpu-baseline > $OUT/bench_2ranks_weak_shared_gpu.json 2> $OUT/b2w.err; echo "2-rank weak rc=$?"
  BIEM_BENCH_SHARE_GPU=1 python bench.py --gpus 2 --scaling strong --steps 1 --warmup 1 --systems-per-gpu 64 --no-cpu-baseline > $OUT/bench_2ranks_strong_shared_gpu.json 2> $OUT/b2s.err; echo "2-rank strong rc=$?"
  BIEM_BENCH_FORCE_DIST=1 python bench.py --gpus 1 --steps 2 --warmup 1 --no-cpu-baseline > $OUT/bench_rccl_1rank.json 2> $OUT/b1r.err; echo "rccl 1 rank rc=$?"
fi
find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*.db" -delete; find $OUT -name "*counter_collection.csv" -size +3M -delete
ls $OUT
