mkdir -p gpurun_out/r3 && export TMPDIR=/tmp
python -m pytest tests -m gpu -x -q --durations=8 > gpurun_out/r3/t.log 2>&1; echo tests_rc=$? >> gpurun_out/r3/t.log; tail -25 gpurun_out/r3/t.log
python bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r3/b.json 2> gpurun_out/r3/b.err; echo bench_rc=$?; cat gpurun_out/r3/b.json
for c in "WRITE_SIZE" "FETCH_SIZE" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAVES"; do
  n=$(echo $c | cut -d' ' -f1)
  rocprofv3 --pmc $c --output-format csv -d gpurun_out/r3/pmc_$n -o p -- python3 bench.py --systems-per-gpu 8 --steps 1 --warmup 0 --no-cpu-baseline > gpurun_out/r3/pmc_$n.log 2>&1 && echo pmc $n ok
done
for n in WRITE_SIZE FETCH_SIZE SQ_WAVE_CYCLES; do for k in k_fill k_pair_tables k_sym_rhs k_gemm3m_pipe; do python tools/pmc_summary.py gpurun_out/r3/pmc_$n $k; done; done > gpurun_out/r3/pmc_summary.txt 2>&1
cat gpurun_out/r3/pmc_summary.txt
find gpurun_out/r3 -name "*.csv" -size +1M -delete; find gpurun_out/r3 -name "*.db" -delete
