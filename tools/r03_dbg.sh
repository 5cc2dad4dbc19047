#!/bin/bash
O=gpurun_out/r03_dbg; mkdir -p $O
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -q -x -k "test_sym_factor_solve_vs_numpy" > $O/t_blk.log 2>&1; echo "blk (lds) exit $?"; tail -2 $O/t_blk.log
BIEM_DBG_SHFL=1 timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -q -x -k "test_sym_factor_solve_vs_numpy" > $O/t_shfl.log 2>&1; echo "blk (shfl) exit $?"; tail -2 $O/t_shfl.log
