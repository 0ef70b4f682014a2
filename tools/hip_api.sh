#!/bin/bash
# host time inside HIP runtime calls per train step: rocprofv3 --hip-runtime-trace --stats of a short bench run
# usage: tools/hip_api.sh <tag> [bench args...] -> gpurun_out/hipapi_<tag>.txt
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
rm -rf /tmp/ha_$tag
cd $R
rocprofv3 --hip-runtime-trace --stats --output-format csv -d /tmp/ha_$tag -- python3 bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-other-modes "$@" > /tmp/ha_$tag.log 2>&1
f=$(find /tmp/ha_$tag -name "*hip_api_stats.csv" | head -1)
echo "file $f" > gpurun_out/hipapi_$tag.txt
head -30 $f >> gpurun_out/hipapi_$tag.txt
tail -3 /tmp/ha_$tag.log | cut -c1-300 >> gpurun_out/hipapi_$tag.txt
