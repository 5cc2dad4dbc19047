mkdir -p gpurun_out/r2 && export TMPDIR=/tmp
python -m pytest tests -m gpu -x -q -k "growth or factor_now or kind_inner or accuracy_sweep or sharded_solve or natural_fallback or near_a_resonance or ldlt_and_lu" > gpurun_out/r2/t2.log 2>&1; echo tests_rc=$? >> gpurun_out/r2/t2.log; tail -15 gpurun_out/r2/t2.log
BIEM_BENCH_SHARE_GPU=1 python bench.py --gpus 2 --steps 1 --warmup 1 --systems-per-gpu 32 --no-cpu-baseline > gpurun_out/r2/b_share2_weak.json 2> gpurun_out/r2/b_share2_weak.err; echo rc=$?; cat gpurun_out/r2/b_share2_weak.json | cut -c1-600
BIEM_BENCH_SHARE_GPU=1 python bench.py --gpus 2 --scaling strong --steps 1 --warmup 1 --systems-per-gpu 64 --no-cpu-baseline > gpurun_out/r2/b_share2_strong.json 2> gpurun_out/r2/b_share2_strong.err; echo rc=$?; cat gpurun_out/r2/b_share2_strong.json | cut -c1-600
BIEM_BENCH_FORCE_DIST=1 python bench.py --gpus 1 --steps 1 --warmup 1 --systems-per-gpu 64 --no-cpu-baseline > gpurun_out/r2/b_rccl1.json 2> gpurun_out/r2/b_rccl1.err; echo rc=$?; python - <<'PY'
import json
j=json.load(open('gpurun_out/r2/b_rccl1.json')); print({k:j[k] for k in ('value','n_gpus','scaling','marshalling_ms','rccl_ranks')})
PY
for c in "WRITE_SIZE" "FETCH_SIZE" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAVES"; do
  n=$(echo $c | cut -d' ' -f1)
  rocprofv3 --pmc $c -d gpurun_out/r2/pmc_fill_$n -o p -- python3 bench.py --systems-per-gpu 8 --steps 1 --warmup 0 --no-cpu-baseline > gpurun_out/r2/pmc_$n.log 2>&1 && echo pmc $n ok
done
for n in WRITE_SIZE FETCH_SIZE SQ_WAVE_CYCLES; do python tools/pmc_summary.py gpurun_out/r2/pmc_fill_$n k_fill; python tools/pmc_summary.py gpurun_out/r2/pmc_fill_$n k_symmetrize; python tools/pmc_summary.py gpurun_out/r2/pmc_fill_$n k_pair_tables; done > gpurun_out/r2/fill_pmc_before.txt 2>&1
cat gpurun_out/r2/fill_pmc_before.txt
find gpurun_out/r2 -name "*.csv" -size +2M -delete; find gpurun_out/r2 -name "*.db" -delete
