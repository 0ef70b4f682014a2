#!/usr/bin/env python3
"""Launches per steady train step of a rocprofv3 --kernel-trace CSV of bench.py, by category and by kernel.
    python tools/launch_count.py <kernel_trace.csv>"""
import csv
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows))
ends = []
for s, e, k in ev:
    if "multi_tensor_apply" in k or "fused_adam" in k.lower():
        if ends and s - ends[-1] < 2_000_000:
            ends[-1] = e
        else:
            ends.append(e)
steps = [(a, b) for a, b in zip(ends, ends[1:]) if b - a < 100_000_000]
steps = steps[len(steps) // 2:]
cat, ker = defaultdict(int), defaultdict(int)
for a, b in steps:
    for s, e, k in ev:
        if a <= s < b:
            c = "smin" if ("smin::" in k or "clip_window" in k) else "hipBLASLt/rocBLAS" if ("Cijk" in k or "rocblas" in k) else "torch/runtime"
            cat[c] += 1
            if c != "smin":
                ker[k.split("(")[0].replace("void ", "")[:110]] += 1
n = len(steps)
print(f"{n} steady steps; launches per step: total {sum(cat.values()) / n:.0f}  " + "  ".join(f"{c} {v / n:.0f}" for c, v in cat.items()))
for k, v in sorted(ker.items(), key=lambda kv: -kv[1])[:25]:
    print(f"  {v / n:6.1f}  {k}")
