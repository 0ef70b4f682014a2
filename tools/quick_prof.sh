#!/bin/bash
# Kernel stats (weight stream folded into main) + one FETCH_SIZE pass of the headline workload:  bash tools/quick_prof.sh <tag>
TAG=${1:-q}; ROOT=$(pwd); OUT=$ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/qp_ks /tmp/qp_fetch
SMIN_SYNC_WEIGHTS=1 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/qp_ks -- python3 $ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-other-modes > $OUT/qp_${TAG}_ks.log 2>&1
python3 $ROOT/profiles/summarize.py $(find /tmp/qp_ks -name "*kernel_stats.csv" | head -1) 13 60 > $OUT/qp_${TAG}_kernel_stats.txt
SMIN_SYNC_WEIGHTS=1 timeout -k 10 280 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/qp_fetch -- python3 $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-other-modes > $OUT/qp_${TAG}_fetch.log 2>&1
python3 - <<PY > $OUT/qp_${TAG}_fetch.txt
import csv,glob,collections
agg=collections.defaultdict(list)
for f in glob.glob('/tmp/qp_fetch/*/*counter_collection.csv')+glob.glob('/tmp/qp_fetch/*/*/*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        if r['Counter_Name']=='FETCH_SIZE': agg[r['Kernel_Name']].append(float(r['Counter_Value']))
for k,v in sorted(agg.items(), key=lambda kv:-sum(kv[1]))[:40]:
    print(f"{k[:110]:110s} launches {len(v):4d}  fetch MB/launch (x2 corrected) {sum(v)/len(v)*1024*2/1e6:9.1f}")
PY
head -45 $OUT/qp_${TAG}_kernel_stats.txt | cut -c1-150
