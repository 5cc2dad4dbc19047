#!/bin/bash
O=gpurun_out/r03_dbg; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -q -k "test_sym_factor or rejections or close_spheres or ldlt" > $O/t.log 2>&1; echo "exit $?"; tail -3 $O/t.log
CFGS="3 4" LINES_SHOWN=6 bash tools/r03_single.sh
