#!/usr/bin/env python3
"""profiles/pmc_moment.json (what bench.py reads for roofline.traffic) from a pmc_summary.json of profiles/pmc_summary.py:
per launch of the three moment-unit contractions, HBM bytes (FETCH_SIZE x 2 on gfx950, + WRITE_SIZE), the matrix-pipe busy
fraction SQ_VALU_MFMA_BUSY_CYCLES / GRBM_GUI_ACTIVE / 128 and the clock GRBM_GUI_ACTIVE / 8 XCDs / duration.
    python profiles/make_pmc_moment.py <pmc_summary.json> [N_valid_cells=100759] [D=512]"""
import json
import sys

src = json.load(open(sys.argv[1]))
N = int(sys.argv[2]) if len(sys.argv) > 2 else 100759
D = int(sys.argv[3]) if len(sys.argv) > 3 else 512
pick = {"moment_fwd": ("gemm_nt_kernel", "EpMomentOut"), "moment_dx": ("gemm_nt_kernel", "EpSplitStore"), "moment_dw": ("gemm_tn_kernel", "CatMat")}
alg = {"moment_fwd": 4 * N * 4 * D, "moment_dx": 4 * N * 4 * D, "moment_dw": 4 * N * 3 * D}      # operands + result, fp32
out = {}
for tag, (a, b) in pick.items():
    cands = [(k, v) for k, v in src.items() if a in k and b in k and "hbm_read_bytes" in v]
    if not cands:
        continue
    k, v = max(cands, key=lambda kv: kv[1].get("avg_ns_under_pmc", 0))
    d = {"kernel": k[:120], "hbm_read_bytes_per_launch": v["hbm_read_bytes"], "hbm_write_bytes_per_launch": v.get("hbm_write_bytes"),
         "hbm_bytes_per_launch": v["hbm_read_bytes"] + v.get("hbm_write_bytes", 0.0), "algorithmic_bytes_per_launch": alg[tag],
         "avg_us_under_pmc": v.get("avg_ns_under_pmc", 0) / 1e3, "launches_averaged": v.get("launches")}
    if "SQ_VALU_MFMA_BUSY_CYCLES" in v and v.get("GRBM_GUI_ACTIVE"):
        d["mfma_busy_fraction"] = v["SQ_VALU_MFMA_BUSY_CYCLES"] / v["GRBM_GUI_ACTIVE"] / 128
        if v.get("avg_ns_under_pmc"):
            d["clock_ghz"] = v["GRBM_GUI_ACTIVE"] / 8 / v["avg_ns_under_pmc"]
    out[tag] = d
# the counters describe the GEMM engine as built from these sources: bench.py drops `traffic` when they have changed since
import hashlib
import os
csrc = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "video-moment-localization_amd", "csrc")
out["sources_sha256"] = {f: hashlib.sha256(open(os.path.join(csrc, f), "rb").read()).hexdigest() for f in ("gemm.h", "moment_unit.hip")}
json.dump(out, sys.stdout, indent=1)
print()
