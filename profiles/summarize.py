#!/usr/bin/env python3
"""Summarise a rocprofv3 --kernel-trace --stats kernel_stats.csv: per-kernel calls, total/avg time, share."""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
steps = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
once = [r for r in rows if "loss_fwd_kernel" in r["Name"]]          # launched once per step: the step count of the whole run
if once:
    steps = float(once[0]["Calls"])
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"# total kernel time {tot/1e6:.3f} ms over {steps:g} steps = {tot/1e6/steps:.3f} ms/step")
print(f"{'kernel':100s} {'calls':>6s} {'total_ms':>10s} {'avg_us':>10s} {'pct':>6s}")
for r in rows[: int(sys.argv[3]) if len(sys.argv) > 3 else 30]:
    print(f"{r['Name'][:100]:100s} {r['Calls']:>6s} {float(r['TotalDurationNs'])/1e6:10.3f} {float(r['AverageNs'])/1e3:10.1f} {float(r['Percentage']):6.2f}")
