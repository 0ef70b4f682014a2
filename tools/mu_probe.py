"""Upper bound for a moment-unit forward with a materialised left operand: the plain NT engine at N x 512 x 1024."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import models
from vml_amd.functional import gemm_nt
dev = torch.device("cuda:0")
N = 100759
a = torch.randn(N, 1024, device=dev); b = torch.randn(512, 1024, device=dev)
for _ in range(3): gemm_nt(a, b)
torch.cuda.synchronize()
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(20): gemm_nt(a, b)
e.record(); torch.cuda.synchronize()
us = s.elapsed_time(e) / 20 * 1e3
print(f"plain NT {N}x512x1024: {us:.1f} us  {2*N*512*1024/us/1e6:.1f} TF")
