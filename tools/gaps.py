#!/usr/bin/env python3
"""Idle gaps of one steady train step in a rocprofv3 --kernel-trace CSV: every gap above a threshold with the kernels around it.
    python tools/gaps.py <kernel_trace.csv> [min_gap_us=15]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
thr = float(sys.argv[2]) if len(sys.argv) > 2 else 15.0
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "")) for r in rows))
adam = [e for e in ev if ("multi_tensor_apply" in e[2] or "fused_adam" in e[2].lower() or "FusedAdam" in e[2])]
ends = []
for s, e, *_ in adam:
    if ends and s - ends[-1] < 2_000_000:
        ends[-1] = e
    else:
        ends.append(e)
steps = [(a, b) for a, b in zip(ends, ends[1:]) if b - a < 100_000_000]
a, b = steps[-2]
w = [e for e in ev if e[0] >= a and e[1] <= b]
print(f"step of {(b - a) / 1e6:.3f} ms, {len(w)} kernels")
busy_until = a
for i, (s, e, name, q) in enumerate(w):
    if s - busy_until > thr * 1000:
        prev = w[i - 1][2][:70] if i else "-"
        print(f"  +{(s - a) / 1e3:9.1f} us  gap {(s - busy_until) / 1e3:7.1f} us   after [{prev}]  before [q{q}] {name[:70]}")
    busy_until = max(busy_until, e)
if len(sys.argv) > 3:                                   # third argument: list every kernel of the first <n> microseconds of the step
    lim = float(sys.argv[3]) * 1000
    for s, e, name, q in w:
        if s - a > lim:
            break
        print(f"    +{(s - a) / 1e3:8.1f} .. {(e - a) / 1e3:8.1f} us  q{q}  {name[:90]}")
