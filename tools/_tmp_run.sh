mkdir -p gpurun_out/pp && export TMPDIR=/tmp
timeout -k 10 300 python -m pytest tests -m gpu -x -q -k "sym_factor or lu_factor_solve or ldlt_factor or end_to_end or cfg2 or factor_now" > gpurun_out/pp/t.log 2>&1; echo tests_rc=$? >> gpurun_out/pp/t.log; tail -6 gpurun_out/pp/t.log
for v in 1 0 1 0; do
BIEM_GEMM_PP=$v timeout -k 10 200 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/pp/b$v.json 2> gpurun_out/pp/b$v.err; echo "PP=$v rc=$?"; python - <<PY
import json
j=json.loads([l for l in open('gpurun_out/pp/b$v.json') if l.startswith('{')][-1])
print(j['value'], j['ms_per_step'], j['stage_ms_per_step']['gemm'], j['stage_ms_per_step']['gemm_small'], j['stage_ms_per_step']['panel'], j['roofline']['achieved'], j['max_rel_err_uscat'])
PY
done
