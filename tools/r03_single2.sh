#!/bin/bash
O=gpurun_out/r03_single; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py -q -x > $O/t.log 2>&1; echo "tests exit $?"; tail -4 $O/t.log
CFGS="${CFGS:-2 4}" LINES_SHOWN=11 bash tools/r03_single.sh
