import os, sys, torch
sys.path.insert(0, os.getcwd())
import models
from vml_amd.functional import gemm_nt
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)
for (M, N, K) in [(100759, 512, 1024), (100759, 512, 512), (100759, 1024, 1024), (100759, 1024, 512), (100759, 256, 1024), (100759, 512, 2048), (98304, 512, 1024), (98304, 1024, 512), (196608, 512, 1024), (49152, 512, 1024)]:
    a, b = torch.randn(M, K, generator=g, device=dev), torch.randn(N, K, generator=g, device=dev)
    for _ in range(3): gemm_nt(a, b)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(20): gemm_nt(a, b)
    e.record(); torch.cuda.synchronize()
    t = s.elapsed_time(e) / 20
    print(f"{M} x {N} x {K}: {t*1e3:.1f} us  {2.0*M*N*K/t/1e9:.1f} TF", flush=True)
