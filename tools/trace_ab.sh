#!/bin/bash
# kernel-trace timelines of the fused core and of the node-per-module graph on one box
# usage: tools/trace_ab.sh [bench args...]   -> gpurun_out/tl_fused.txt, gpurun_out/tl_nodes.txt
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
rm -rf /tmp/tla /tmp/tlb
cd $R
rocprofv3 --kernel-trace --output-format csv -d /tmp/tla -- python3 bench.py --steps 16 --warmup 4 --no-cpu-baseline --no-other-modes "$@" > /dev/null 2>&1
python tools/timeline.py $(find /tmp/tla -name "*kernel_trace.csv" | head -1) > gpurun_out/tl_fused.txt 2>&1
export SMIN_NODE_GRAPH=1
rocprofv3 --kernel-trace --output-format csv -d /tmp/tlb -- python3 bench.py --steps 16 --warmup 4 --no-cpu-baseline --no-other-modes "$@" > /dev/null 2>&1
python tools/timeline.py $(find /tmp/tlb -name "*kernel_trace.csv" | head -1) > gpurun_out/tl_nodes.txt 2>&1
