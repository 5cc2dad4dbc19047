#!/bin/bash
# kernel timelines of one system per call (tools/single_timeline.py) on the GPU box: CFGS="1 2 3 4" LINES_SHOWN=14 bash tools/single_timeline.sh
# (also runs tools/gemm_small when it has been built: the small update launches alone)
O=gpurun_out/r03_single; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
[ -x $R/tools/gemm_small ] && timeout -k 10 120 $R/tools/gemm_small 4096
for cfg in ${CFGS:-4 2 3}; do
  rm -rf /tmp/st$cfg
  timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/st$cfg -- python3 $R/tools/single_timeline.py run $cfg > $R/$O/run$cfg.log 2>&1 || { echo "cfg $cfg failed"; tail -5 $R/$O/run$cfg.log; exit 1; }
  grep "^call" $R/$O/run$cfg.log | tail -2
  python3 $R/tools/single_timeline.py report /tmp/st$cfg > $R/$O/timeline$cfg.txt 2>&1; head -${LINES_SHOWN:-12} $R/$O/timeline$cfg.txt
done
