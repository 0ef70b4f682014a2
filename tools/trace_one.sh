#!/bin/bash
# usage: tools/trace_one.sh <tag> [bench args...] -> gpurun_out/tl_<tag>.txt, gpurun_out/gaps_<tag>.txt
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
rm -rf /tmp/tl_$tag
cd $R
rocprofv3 --kernel-trace --output-format csv -d /tmp/tl_$tag -- python3 bench.py --steps 16 --warmup 4 --no-cpu-baseline --no-other-modes "$@" > /dev/null 2>&1
f=$(find /tmp/tl_$tag -name "*kernel_trace.csv" | head -1)
python tools/timeline.py $f > gpurun_out/tl_$tag.txt 2>&1
python tools/gaps.py $f 12 ${HEAD_US:-0} > gpurun_out/gaps_$tag.txt 2>&1
