#!/usr/bin/env python3
"""Micro-benchmark of the content attention core (smin_content_attn_fwd / _bwd) at the bench workload's geometry:
the packed cell list of the activitynet_t256 batch (64 samples, 100 759 valid cells), random operands.
    python tools/attn_bench.py [--iters 20] [--layers-like both|rows|mean]
Prints per-launch milliseconds measured by the library's own HIP events (smin_prof_*)."""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--L", type=int, default=64)
    ap.add_argument("--T", type=int, default=256)
    ap.add_argument("--nq", type=int, default=20)
    ap.add_argument("--dl", type=int, default=128)
    args = ap.parse_args()
    import bench
    import models
    from vml_amd.cells import CellLayout
    from vml_amd.functional import ContentAttnFn
    lib = models.vml_amd._lib
    dev = torch.device("cuda:0")
    b = bench.make_batch(args.batch, args.T, args.L, args.nq, 500, seed=1000, device=dev)
    lay = CellLayout.from_mask(b["moment_mask"])
    N, C, dl, Nq, B = lay.N, 4, args.dl, args.nq, args.batch
    g = torch.Generator(device=dev).manual_seed(1)
    r = lambda *s: torch.randn(*s, generator=g, device=dev)
    chat = (r(N * C, dl) * 0.5).requires_grad_(True)
    Mq, uq, what, shat = (r(B, Nq, dl) * 0.2).requires_grad_(True), (r(B, Nq) * 0.2).requires_grad_(True), r(B, Nq, dl).requires_grad_(True), r(B, dl).requires_grad_(True)
    qmask = b["query_mask"].reshape(B, -1).float()
    for want_rows in (True, False):
        for it in range(3):
            cc, ccm = ContentAttnFn.apply(chat, Mq, uq, what, shat, qmask, lay, C, want_rows)
            ((cc.sum() if want_rows else 0) + ccm.sum()).backward()
        lib.prof_enable(True)
        for it in range(args.iters):
            cc, ccm = ContentAttnFn.apply(chat, Mq, uq, what, shat, qmask, lay, C, want_rows)
            ((cc.sum() if want_rows else 0) + ccm.sum()).backward()
        torch.cuda.synchronize()
        lib.prof_enable(False)
        p = lib.prof_read()
        f, bw = p["attn_fwd"], p["attn_bwd"]
        rows = N * C
        print(f"N={N} rows={rows} want_rows={want_rows}: fwd {sum(f)/len(f)*1e3:.1f} us (min {min(f)*1e3:.1f})  bwd {sum(bw)/len(bw)*1e3:.1f} us (min {min(bw)*1e3:.1f})")


if __name__ == "__main__":
    main()
