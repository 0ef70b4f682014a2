#!/bin/bash
# usage: tools/launches.sh <tag> [bench args...] -> gpurun_out/launches_<tag>.txt
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
rm -rf /tmp/lc_$tag
cd $R
rocprofv3 --kernel-trace --output-format csv -d /tmp/lc_$tag -- python3 bench.py --steps 16 --warmup 4 --no-cpu-baseline --no-other-modes "$@" > /dev/null 2>&1
f=$(find /tmp/lc_$tag -name "*kernel_trace.csv" | head -1)
python tools/launch_count.py $f > gpurun_out/launches_$tag.txt 2>&1
python tools/timeline.py $f > gpurun_out/tl_$tag.txt 2>&1
