#!/bin/bash
# uscat per-lane kernel for kind inner and tree caa: parity tests, then timings
O=gpurun_out/r03_uscat; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -q -k "uscat or inner or caa" > $O/t.log 2>&1; echo "tests exit $?"; tail -15 $O/t.log
timeout -k 10 300 python tools/time_uscat.py 64 all > $O/time.log 2>&1; cat $O/time.log
