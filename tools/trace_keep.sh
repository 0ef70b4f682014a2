#!/bin/bash
# usage: tools/trace_keep.sh <tag> [bench args...] -> gpurun_out/trace_<tag>.csv.gz (kernel trace of a short bench run) + underfill summary
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
rm -rf /tmp/tl_$tag
cd $R
rocprofv3 --kernel-trace --output-format csv -d /tmp/tl_$tag -- python3 bench.py --steps 8 --warmup 4 --no-cpu-baseline --no-other-modes "$@" > /dev/null 2>&1
f=$(find /tmp/tl_$tag -name "*kernel_trace.csv" | head -1)
python tools/underfill.py $f > gpurun_out/underfill_$tag.txt 2>&1
python tools/timeline.py $f > gpurun_out/tl_$tag.txt 2>&1
python - "$f" gpurun_out/trace_$tag.csv.gz <<'PY'
import csv, gzip, sys
rows = list(csv.DictReader(open(sys.argv[1])))
keep = [k for k in rows[0] if k in ("Kernel_Name", "Start_Timestamp", "End_Timestamp", "Queue_Id", "Grid_Size_X", "Grid_Size_Y", "Grid_Size_Z", "Workgroup_Size_X", "Workgroup_Size_Y", "Workgroup_Size_Z", "Grid_Size", "Workgroup_Size")]
with gzip.open(sys.argv[2], "wt") as f:
    w = csv.DictWriter(f, keep); w.writeheader()
    for r in rows:
        r = {k: r[k] for k in keep}; r["Kernel_Name"] = r["Kernel_Name"][:120]; w.writerow(r)
PY
