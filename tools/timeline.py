#!/usr/bin/env python3
"""Timeline view of a rocprofv3 --kernel-trace CSV of bench.py: for the steady-state train steps (delimited by the Adam
launches) the wall time covered by kernels, device idle time, time with two kernels in flight (second HIP stream), and
where the idle gaps sit.
    python tools/timeline.py <kernel_trace.csv>"""
import csv
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "")) for r in rows))
adam = [e for e in ev if ("multi_tensor_apply" in e[2] or "fused_adam" in e[2].lower() or "FusedAdam" in e[2])]
ends = []
for s, e, *_ in adam:
    if ends and s - ends[-1] < 2_000_000:
        ends[-1] = e
    else:
        ends.append(e)
# steps with Adam are consecutive cluster ends less than 100 ms apart
steps = [(a, b) for a, b in zip(ends, ends[1:]) if b - a < 100_000_000]
steps = steps[len(steps) // 2:]                       # second half: steady state
print(f"{len(steps)} steady steps, mean {sum(b - a for a, b in steps) / len(steps) / 1e6:.3f} ms")
tot = defaultdict(float)
gaps = defaultdict(lambda: [0.0, 0])
perq = defaultdict(float)
kern = defaultdict(float)
for a, b in steps:
    w = [e for e in ev if e[1] > a and e[0] < b]
    pts = []
    for s, e, *_ in w:
        pts.append((max(s, a), 1)); pts.append((min(e, b), -1))
    pts.sort()
    depth, last = 0, a
    for t, d in pts:
        if depth >= 1: tot["busy"] += t - last
        if depth >= 2: tot["overlap"] += t - last
        if depth == 0: tot["idle"] += t - last
        depth += d; last = t
    tot["idle"] += b - last
    tot["span"] += b - a
    cur = a
    for s, e, k, q in w:
        if s > cur:
            g = gaps[k[:80]]; g[0] += s - cur; g[1] += 1
        cur = max(cur, e)
        perq[q] += min(e, b) - max(s, a)
        kern[k[:100]] += min(e, b) - max(s, a)
n = len(steps)
print(f"per step: span {tot['span']/n/1e6:.3f} ms, device busy {tot['busy']/n/1e6:.3f} ms ({tot['busy']/tot['span']*100:.1f}%), "
      f"idle {tot['idle']/n/1e6:.3f} ms, two kernels in flight {tot['overlap']/n/1e6:.3f} ms, kernel time summed {sum(kern.values())/n/1e6:.3f} ms")
for q, v in sorted(perq.items(), key=lambda kv: -kv[1]):
    print(f"  queue {q}: busy {v/n/1e6:.3f} ms per step")
print("idle gaps by the kernel that follows (us per step, count per step):")
for k, (t, c) in sorted(gaps.items(), key=lambda kv: -kv[1][0])[:12]:
    print(f"  {t/n/1e3:8.1f} us {c/n:6.1f}  {k}")
print("kernel time per step (us):")
for k, v in sorted(kern.items(), key=lambda kv: -kv[1])[:int(sys.argv[2]) if len(sys.argv) > 2 else 45]:
    print(f"  {v/n/1e3:8.1f}  {k}")
# optional: text Gantt of the last steady step -- one line per kernel: start offset (us), duration (us), queue, kernels in flight
# at its start, name.      python tools/timeline.py <csv> 45 --gantt
if "--gantt" in sys.argv:
    a, b = steps[-1]
    w = [e for e in ev if e[1] > a and e[0] < b]
    print(f"gantt of the last steady step ({(b - a) / 1e6:.3f} ms):")
    for s, e, k, q in w:
        inflight = sum(1 for s2, e2, *_ in w if s2 <= s < e2)
        short = k.replace("smin::", "").replace("void ", "")
        print(f"  {(s - a) / 1e3:9.1f} {(e - s) / 1e3:8.1f}  q{q} x{inflight}  {short[:110]}")
