#!/bin/bash
# text Gantt of one steady step of bench.py (default three-stream step):  bash tools/gantt.sh <tag> [bench args...] -> gpurun_out/gantt_<tag>.txt
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
rm -rf /tmp/gt_$tag
cd $R
rocprofv3 --kernel-trace --output-format csv -d /tmp/gt_$tag -- python3 bench.py --steps 12 --warmup 4 --no-cpu-baseline --no-other-modes "$@" > /dev/null 2>&1
f=$(find /tmp/gt_$tag -name "*kernel_trace.csv" | head -1)
python tools/timeline.py $f 10 --gantt > gpurun_out/gantt_$tag.txt 2>&1
