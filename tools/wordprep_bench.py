#!/usr/bin/env python3
"""Word-side operand kernels (csrc/word_prep.hip) alone: forward and backward (+reduce) per launch at the bench shape."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import models
from vml_amd.functional import WordPrepFn
dev = torch.device("cuda:0")
for (B, Nq, D, dl, nl) in [(64, 20, 512, 128, 3), (16, 20, 512, 128, 3), (32, 13, 512, 128, 3), (2, 14, 512, 128, 3)]:
    g = torch.Generator(device=dev).manual_seed(0)
    r = lambda *s: torch.randn(*s, generator=g, device=dev)
    fw, fs, qm = r(B, Nq, D).requires_grad_(True), r(B, D).requires_grad_(True), torch.ones(B, Nq, device=dev)
    params = []
    for k in range(nl):
        params += [r(dl, D), r(dl), r(dl, D), r(dl), r(dl, dl), r(dl), r(dl, dl), r(dl)]
    params = [p.requires_grad_(True) for p in params]
    def fwd(): return WordPrepFn.apply(fw, fs, qm, *params)
    outs = fwd()
    gr = [torch.randn_like(o) for o in outs]
    def both():
        o = fwd(); torch.autograd.backward(o, gr)
    for _ in range(3): both()
    torch.cuda.synchronize()
    def timed(fn, n=20):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(n): fn()
        e.record(); torch.cuda.synchronize()
        return s.elapsed_time(e) / n * 1e3
    tf = timed(fwd); tb = timed(both)
    print(f"B={B} Nq={Nq} layers={nl}: forward {tf:.1f} us, forward+backward {tb:.1f} us (backward ~{tb - tf:.1f} us)")
