"""Micro-benchmark of the content-stream contractions at the default workload's shapes (run on the GPU box)."""
import importlib.util, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import models  # noqa: F401  (loads the package as vml_amd)
from vml_amd import _lib
from vml_amd._lib import call, ptr, stream

dev = torch.device("cuda:0")
N, C = 100759, 4
M = N * C


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3


def fwd(R, O, K, add, cells=False):
    x = torch.randn(R, K, device=dev)
    W = torch.randn(O, K, device=dev)
    a = torch.randn(R, O, device=dev) if add else None
    ac = torch.randn(R // C, O, device=dev) if cells else None
    y = torch.empty(R, O, device=dev)
    us = timeit(lambda: call("smin_linear_rows_fwd", stream(), (__import__("ctypes").c_void_p * 1)(x.data_ptr()), 1, ptr(W), None, ptr(a), ptr(ac), C, R, O, K, ptr(y)))
    ref = timeit(lambda: torch.matmul(x, W.t()))
    print(f"fwd R={R} O={O} K={K} add={add}: {us:7.1f} us  {2*R*O*K/us/1e6:6.1f} TF   (torch.matmul {ref:7.1f} us)")


def bwd(R, O, K):
    import ctypes
    x = torch.randn(R, K, device=dev)
    WT = torch.randn(K, O, device=dev)
    dy = torch.randn(R, O, device=dev)
    dx = torch.empty(R, K, device=dev)
    dW = torch.empty(O, K, device=dev)
    db = torch.empty(O, device=dev)
    nb = _lib.load().smin_linear_rows_bwd_workspace_bytes(R, O, K)
    ws = torch.empty(nb, dtype=torch.uint8, device=dev)
    xs, dxs = (ctypes.c_void_p * 1)(x.data_ptr()), (ctypes.c_void_p * 1)(dx.data_ptr())
    us = timeit(lambda: call("smin_linear_rows_bwd", stream(), ptr(dy), xs, 1, ptr(WT), R, O, K, dxs, ptr(dW), ptr(db), ptr(ws), nb))
    print(f"bwd R={R} O={O} K={K}: {us:7.1f} us (dx + dW)  {4*R*O*K/us/1e6:6.1f} TF")


fwd(M, 128, 128, False)
fwd(M, 128, 128, True)
fwd(M, 128, 128, True, True)
fwd(M, 128, 256, True)
fwd(N, 512, 128, True)
fwd(N, 128, 512, False)
bwd(M, 128, 128)
bwd(N, 512, 128)
bwd(N, 128, 512)
