#!/usr/bin/env python3
"""The three moment-unit contractions (forward, input gradients, weight gradients) at the bench workload's size in every
contraction mode, alone on the device.  python tools/mu_bench.py [--modes f32,f32e,bf16x3,bf16] [--iters 20]"""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def timed(fn, iters):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--modes", default="f32,f32e,bf16x3,bf16")
    args = ap.parse_args()
    import models
    from vml_amd._lib import call, ptr, stream
    import vml_amd.cells as cells_mod
    V = models.vml_amd
    lib = V._lib.load()
    dev = torch.device("cuda:0")
    B, L, D = 64, 64, 512
    g = torch.Generator().manual_seed(0)
    lens = torch.where(torch.arange(B) % 2 == 0, torch.full((B,), L), torch.randint(L // 4, L + 1, (B,), generator=g))
    lm = torch.arange(L).unsqueeze(0) < lens.unsqueeze(1)
    mm = torch.triu(lm.unsqueeze(2) & lm.unsqueeze(1)).to(dev)
    lay = V.CellLayout.begin(mm).finish()
    N = lay.N
    gd = torch.Generator(device=dev).manual_seed(0)
    r = lambda *s: torch.randn(*s, generator=gd, device=dev)
    fcm, fm, fb, x1, dmu = r(N, D), r(N, D), r(B, L, D), r(N, D), r(N, D)
    W, b = r(D, 2 * D) * 0.03, r(D)
    WT = W.t().contiguous()
    mu, dfc, dfb, dW, db = torch.empty(N, D, device=dev), torch.empty(N, D, device=dev), torch.empty(B, L, D, device=dev), torch.empty_like(W), torch.empty(D, device=dev)
    nb = lib.smin_workspace_bytes(N, B, 4, D, 4, 1)
    ws = torch.empty(nb, dtype=torch.uint8, device=dev)
    flops = 2.0 * N * D * 2 * D
    print(f"N = {N} cells; {flops / 1e9:.1f} GFLOP per contraction")
    for mode in ["f32"] + args.modes.split(","):                    # (the first pass warms the device up and is not printed)
        first = mode == "f32" and not hasattr(main, "_warm")
        main._warm = True
        V.set_gemm_mode(mode)
        try:
            tf = timed(lambda: call("smin_moment_unit_fwd", stream(), ptr(fcm), ptr(fm), ptr(fb), ptr(lay.cells), N, B, L, D, ptr(W), ptr(b), ptr(mu), ptr(x1)), args.iters)
            ti = timed(lambda: call("smin_moment_unit_bwd", stream(), ptr(dmu), ptr(fcm), ptr(fb), ptr(lay.cells), ptr(lay.row_ptr), ptr(lay.cellmap), N, B, L, D, ptr(WT),
                                    ptr(dfc), ptr(dfb), None, None, ptr(ws), ws.numel(), 1, None, ptr(x1), None), args.iters)
            tw = timed(lambda: call("smin_moment_unit_bwd", stream(), ptr(dmu), ptr(fcm), ptr(fb), ptr(lay.cells), ptr(lay.row_ptr), ptr(lay.cellmap), N, B, L, D, ptr(WT),
                                    None, None, ptr(dW), ptr(db), ptr(ws), ws.numel(), 1, None, ptr(x1), None), args.iters)
        finally:
            V.set_gemm_mode("f32")
        if first:
            continue
        print(f"[{mode:7s}] fwd {tf:7.1f} us ({flops / tf / 1e6:6.1f} TF)   dX (+dfb) {ti:7.1f} us ({flops / ti / 1e6:6.1f} TF)   dW (+reduce) {tw:7.1f} us ({flops / tw / 1e6:6.1f} TF)")


if __name__ == "__main__":
    main()
