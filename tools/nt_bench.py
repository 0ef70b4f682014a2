#!/usr/bin/env python3
"""Plain NT engine at the moment unit's shapes (forward N x 512 x 1024, input gradient N x 1024 x 512): TF per launch."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import models
from vml_amd.functional import gemm_nt
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)
for (M, N, K) in [(100759, 512, 1024), (100759, 1024, 512), (403036, 128, 128)]:
    a, b = torch.randn(M, K, generator=g, device=dev), torch.randn(N, K, generator=g, device=dev)
    for _ in range(3): gemm_nt(a, b)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(20): gemm_nt(a, b)
    e.record(); torch.cuda.synchronize()
    t = s.elapsed_time(e) / 20
    print(f"{M} x {N} x {K}: {t*1e3:.1f} us  {2.0*M*N*K/t/1e9:.1f} TF")
