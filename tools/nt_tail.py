"""NT engine: cost of the rows beyond a whole number of rounds (98 304 = 4 x 768 tiles of 128 x 128 at N = 512)."""
import os, sys, torch
sys.path.insert(0, os.environ.get("SMIN_ROOT") or os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import models
from vml_amd.functional import gemm_nt
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)
for K in (1024, 2048):
    for extra in (0, 32, 128, 512, 1024, 2455, 4096, 8192, 24576):
        M, N = 98304 + extra, 512
        a, b = torch.randn(M, K, generator=g, device=dev), torch.randn(N, K, generator=g, device=dev)
        for _ in range(3): gemm_nt(a, b)
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(20): gemm_nt(a, b)
        e.record(); torch.cuda.synchronize()
        t = s.elapsed_time(e) / 20
        print(f"K={K} rows 98304+{extra}: {t*1e3:.1f} us  {2.0*M*N*K/t/1e9:.1f} TF", flush=True)
