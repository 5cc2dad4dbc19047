#!/bin/bash
bash tools/collect_evidence.sh configs
CFGS="1 2 3 4" LINES_SHOWN=14 bash tools/r03_single.sh > gpurun_out/r03_single/all.txt 2>&1; tail -3 gpurun_out/r03_single/all.txt
