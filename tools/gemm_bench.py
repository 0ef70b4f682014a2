#!/usr/bin/env python3
"""Micro-benchmark of the content stream's row-wise contractions and the moment unit at the bench workload's sizes
(N = 100 759 cells, C = 4, D = 512, dl = 128): forward and backward of every LinearRows shape of a step, timed with
torch events over many launches.  python tools/gemm_bench.py [--iters 20]"""
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
    ap.add_argument("--cells", type=int, default=100759)
    args = ap.parse_args()
    import models
    from vml_amd._lib import call, ptr, stream
    lib = models.vml_amd._lib.load()
    dev = torch.device("cuda:0")
    N, C, D, dl = args.cells, 4, 512, 128
    R = N * C
    g = torch.Generator(device=dev).manual_seed(0)
    r = lambda *s: torch.randn(*s, generator=g, device=dev)
    import ctypes

    def parr(ts):
        return (ctypes.c_void_p * len(ts))(*[t.data_ptr() for t in ts])

    def linear_fwd(name, rows, K, O, nseg, add_rows, add_cells, Cc, bias=True):
        xs = [r(rows, K) for _ in range(nseg)]
        W = r(O, nseg * K)
        b = r(O) if bias else None
        ar = r(rows, O) if add_rows else None
        ac = r(rows // Cc, O) if add_cells else None
        y = torch.empty(rows, O, device=dev)
        t = timed(lambda: call("smin_linear_rows_fwd", stream(), parr(xs), nseg, ptr(W), ptr(b), ptr(ar), ptr(ac), Cc, rows, O, K, ptr(y)), args.iters)
        flops = 2.0 * rows * O * nseg * K
        byts = 4.0 * (rows * K * nseg + rows * O * (1 + bool(add_rows)) + (rows // Cc) * O * bool(add_cells))
        print(f"{name:46s} {t:8.1f} us  {flops / t / 1e6:6.1f} TF  {byts / t / 1e3:7.1f} GB/s")
        # backward: dX + dW
        dy = r(rows, O)
        WT = W.t().contiguous()
        dxs = [torch.empty_like(x) for x in xs]
        dW, db = torch.empty_like(W), torch.empty(O, device=dev)
        nb = lib.smin_linear_rows_bwd_workspace_bytes(rows, O, nseg * K)
        ws = torch.empty(nb + 4096, dtype=torch.uint8, device=dev)
        t = timed(lambda: call("smin_linear_rows_bwd", stream(), ptr(dy), parr(xs), nseg, ptr(WT), rows, O, K, parr(dxs), ptr(dW), ptr(db), ptr(ws), ws.numel()),
                  args.iters)
        print(f"{'   backward (dX + dW + reduce)':46s} {t:8.1f} us  {2 * flops / t / 1e6:6.1f} TF")

    linear_fwd("chat_1: rows x 128 x 128 + rows + cells", R, dl, dl, 1, True, True, C)
    linear_fwd("chat_2: rows x 128 x 256 + rows + cells", R, dl, dl, 2, True, True, C)
    linear_fwd("gate term: N x 128 x 512", N, D, dl, 1, False, False, 1, bias=False)
    linear_fwd("clip-mean update: N x 512 x 128 + rows + cells", N, dl, D, 1, True, True, 1)
    linear_fwd("g = f Wch^T: (B T) x 384 x 512", 64 * 256, D, 3 * dl, 1, False, False, 1, bias=False)


if __name__ == "__main__":
    main()
