mkdir -p gpurun_out/r5 && export TMPDIR=/tmp
python -m pytest tests -m gpu -x -q -k "sym_factor or symmetric_fill or end_to_end or factor_once or ldlt or cfg2 or cfg4 or golden or batched or growth or complex_wavenumber" > gpurun_out/r5/t.log 2>&1; echo tests_rc=$? >> gpurun_out/r5/t.log; tail -30 gpurun_out/r5/t.log
