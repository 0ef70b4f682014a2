#!/usr/bin/env python3
"""Repeat the fused step of the small test shapes with the weight stream on / off and report which gradients ever differ from
the first run (a rare difference = a cross-stream race).  python tools/flaky_probe.py [reps]"""
import os, sys
import torch
ROOT = os.environ.get("SMIN_ROOT") or os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import models  # noqa: F401 (loads the package as vml_amd)
from oracle import smin_oracle as O          # seeded inputs / weights only (a diagnostic tool, not the product)
from tests import helpers as H
from tests.test_hip_parity import build_model
from vml_amd import loss_fn
dev = torch.device("cuda:0")
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 60
for (T, L, C, D, dl, layers, Din, Nq, Hh, B) in [(64, 16, 4, 128, 32, 3, 40, 9, 64, 5), (128, 64, 4, 512, 128, 3, 500, 20, 256, 3), (64, 16, 2, 128, 32, 5, 40, 9, 64, 3)]:
    sd = O.formula_state_dict(H.smin_shapes(T, L, C, D, dl, layers, Din, Nq, Hh), gain=1.2)
    batch = O.synthetic_batch(B, T, L, Nq, Din, seed=21)
    b = {k: v.to(dev) for k, v in batch.items()}
    ref, bad = None, {}
    for it in range(reps):
        m = build_model(dict(T=T, L=L, C=C, D=D, dl=dl, layers=layers, Din=Din, Nq=Nq, H=Hh), sd, dev)
        m.async_weights = bool(it % 2 == 0)
        out = m(*H.model_inputs(b))
        loss_fn(out[0], b["ym"], b["sm"], b["moment_mask"], out[1], b["ys"], b["ss"], out[2], b["ye"], b["se"], out[3], b["ya"], b["length_mask"]).backward()
        torch.cuda.synchronize()
        cur = {k: p.grad.clone() for k, p in m.named_parameters()}
        if ref is None:
            ref = cur
            continue
        for k in ref:
            if not torch.equal(ref[k], cur[k]):
                d = (ref[k] - cur[k]).abs()
                e = bad.setdefault(k, [0, 0.0, set()])
                e[0] += 1; e[1] = max(e[1], float(d.max() / (ref[k].abs().max() + 1e-30))); e[2].add("async" if it % 2 == 0 else "sync")
    print(f"shape {(T, L, C, D, dl, layers, Din, Nq, Hh, B)}: {len(bad)} tensors differed in {reps - 1} runs")
    for k, (n, rel, modes) in list(bad.items())[:12]:
        print(f"    {k}: {n} runs, max rel {rel:.2e}, in {sorted(modes)}")
