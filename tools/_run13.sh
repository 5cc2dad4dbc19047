mkdir -p gpurun_out/r13 && export TMPDIR=/tmp
python -m pytest tests -m gpu -x -q -k "order_ceilings or symmetric_fill" > gpurun_out/r13/t.log 2>&1; echo tests_rc=$? >> gpurun_out/r13/t.log; tail -25 gpurun_out/r13/t.log
