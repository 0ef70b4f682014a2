#!/usr/bin/env python3
"""How much of one steady train step runs with the chip under-filled: time during which all kernels in flight together have
fewer workgroups than the device has CUs, attributed to the kernels in flight then.
    python tools/underfill.py <kernel_trace.csv> [cus=256] [list_from_us list_to_us]"""
import csv
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
cus = int(sys.argv[2]) if len(sys.argv) > 2 else 256


def wgs(r):
    g = int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"]) if "Grid_Size_X" in r else int(r["Grid_Size"])
    w = int(r["Workgroup_Size_X"]) * int(r["Workgroup_Size_Y"]) * int(r["Workgroup_Size_Z"]) if "Workgroup_Size_X" in r else int(r["Workgroup_Size"])
    return max(1, g // max(1, w))


ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", ""), wgs(r)) for r in rows))
adam = [e for e in ev if ("multi_tensor_apply" in e[2] or "fused_adam" in e[2].lower() or "FusedAdam" in e[2])]
ends = []
for s, e, *_ in adam:
    if ends and s - ends[-1] < 2_000_000:
        ends[-1] = e
    else:
        ends.append(e)
steps = [(a, b) for a, b in zip(ends, ends[1:]) if b - a < 400_000_000]
a, b = steps[-2]
w = [e for e in ev if e[0] >= a and e[1] <= b]
pts = sorted({a, b} | {e[0] for e in w} | {e[1] for e in w})
under = 0
idle = 0
by = defaultdict(float)
segs = []
for p, q in zip(pts, pts[1:]):
    live = [e for e in w if e[0] <= p and e[1] >= q]
    tot = sum(e[4] for e in live)
    if not live:
        idle += q - p
    elif tot < cus:
        under += q - p
        for e in live:
            by[e[2][:90]] += (q - p) / len(live)
    segs.append((p, q, tot, live))
print(f"step {(b - a) / 1e6:.3f} ms: idle {idle / 1e6:.3f} ms, under-filled (< {cus} workgroups in flight) {under / 1e6:.3f} ms")
for k, v in sorted(by.items(), key=lambda kv: -kv[1])[:25]:
    print(f"  {v / 1e3:8.1f} us  {k}")
if len(sys.argv) > 4:
    lo, hi = float(sys.argv[3]) * 1000 + a, float(sys.argv[4]) * 1000 + a
    for s, e, name, q, n in w:
        if e >= lo and s <= hi:
            print(f"    +{(s - a) / 1e3:8.1f} .. {(e - a) / 1e3:8.1f} us  q{q}  wg {n:6d}  {name[:90]}")
