#!/bin/bash
CFGS="${CFGS:-2}" LINES_SHOWN=4 bash tools/r03_single.sh
