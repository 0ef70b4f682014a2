"""C += A B^T through the accumulate epilogue with A = 0: C must come back unchanged -- alone and beside another stream.  MODE=f32e"""
import os, sys, ctypes, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import models
from vml_amd._lib import call, ptr
V = models.vml_amd
lib = V._lib.load()
dev = torch.device("cuda:0")
V.set_gemm_mode(os.environ.get("MODE", "f32e"))
g = torch.Generator().manual_seed(0)
arr = lambda ts: (ctypes.c_void_p * len(ts))(*[t.data_ptr() for t in ts])
side, main = torch.cuda.Stream(), torch.cuda.current_stream()
sp = lambda s: ctypes.c_void_p(s.cuda_stream)
r = lambda *s: torch.randn(*s, generator=g).to(dev)
Rb, Ob, Kb_ = 100000, 512, 1024
xb, Wb, yb = r(Rb, Kb_), r(Ob, Kb_), torch.empty(Rb, Ob, device=dev)
for (M, N, K) in [(128, 512, 512), (40, 512, 512), (1024, 512, 512), (100000, 512, 128)]:
    A0, A1, Bm = torch.zeros(M, K, device=dev), r(M, K), r(N, K)
    C0 = r(M, N)
    ref = (C0.double() + A1.double() @ Bm.double().t())
    for rep in range(8):
        if rep % 2 == 1:
            for _ in range(2):
                call("smin_linear_rows_fwd", sp(side), arr([xb]), 1, ptr(Wb), None, None, None, 1, Rb, Ob, Kb_, ptr(yb))
        Ca, Cb = C0.clone(), C0.clone()
        torch.cuda.synchronize() if rep % 2 == 0 else None
        call("smin_gemm_nt_acc", sp(main), ptr(A0), ptr(Bm), ptr(Ca), M, N, K)
        call("smin_gemm_nt_acc", sp(main), ptr(A1), ptr(Bm), ptr(Cb), M, N, K)
        torch.cuda.synchronize()
        bad0 = int((Ca != C0).sum())
        err = float((Cb.double() - ref).abs().max() / ref.abs().max())
        print((M, N, K), "rep", rep, "concurrent" if rep % 2 else "alone", ": C += 0 changed", bad0, "entries;  C += A B^T rel err", f"{err:.2e}")
