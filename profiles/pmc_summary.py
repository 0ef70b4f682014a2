#!/usr/bin/env python3
"""Per-kernel averages of the rocprofv3 --pmc passes written by profiles/run_pmc.sh.
FETCH_SIZE is doubled (gfx950 counts 128-B requests as 64 B: MI355X_MICROARCH.md, HBM) and both sizes are
reported in bytes per launch (the counters are in KB)."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

root = sys.argv[1]
agg = defaultdict(lambda: defaultdict(list))
for f in glob.glob(os.path.join(root, "*", "*", "*counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        agg[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
durs = defaultdict(list)
for f in glob.glob(os.path.join(root, "sq1", "*", "*kernel_trace.csv")):
    for r in csv.DictReader(open(f)):
        durs[r["Kernel_Name"]].append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
out = {}
for k, cs in agg.items():
    if "smin::" not in k:
        continue
    d = {c: sum(v) / len(v) for c, v in cs.items()}
    d["launches"] = max(len(v) for v in cs.values())
    if k in durs:
        d["avg_ns_under_pmc"] = sum(durs[k]) / len(durs[k])
    if "FETCH_SIZE" in d:
        d["hbm_read_bytes"] = d["FETCH_SIZE"] * 1024 * 2
    if "WRITE_SIZE" in d:
        d["hbm_write_bytes"] = d["WRITE_SIZE"] * 1024
    if "SQ_VALU_MFMA_BUSY_CYCLES" in d and d.get("SQ_BUSY_CYCLES"):
        pass
    out[k[:140]] = d
json.dump(out, open(os.path.join(root, "pmc_summary.json"), "w"), indent=1, sort_keys=True)
for k, d in sorted(out.items(), key=lambda kv: -kv[1].get("avg_ns_under_pmc", 0))[:14]:
    print(k[:110])
    print("   ", {c: (round(v, 1) if isinstance(v, float) else v) for c, v in d.items()})
