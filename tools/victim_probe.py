"""Plain torch kernels (rocBLAS mm, element-wise, reduction) repeated on one stream beside this library's contraction kernels on
another: do THEIR results change?  MODE=f32e|f32  (a torch victim going wrong would put the cause outside this library's victims)"""
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
a, b, c = r(128, 512), r(512, 512), r(128, 512)
big = r(4096, 512)
idx = torch.randint(0, 4096, (8192,), generator=g).to(dev)
victims = {"rocBLAS mm 128x512x512": lambda: torch.mm(a, b),
           "elementwise a*c+a": lambda: a * c + a,
           "row gather + weighted sum (loop of 64 fmas)": lambda: sum(big[idx[k * 128:(k + 1) * 128]] * float(k + 1) for k in range(64)),
           "sum over rows": lambda: big.sum(0)}
for name, fn in victims.items():
    ref = fn(); torch.cuda.synchronize()
    bad = 0
    for rep in range(int(os.environ.get("REPS", 40))):
        for _ in range(2):
            call("smin_linear_rows_fwd", sp(side), arr([xb]), 1, ptr(Wb), None, None, None, 1, Rb, Ob, Kb_, ptr(yb))
        out = fn()
        torch.cuda.synchronize()
        if not torch.equal(out, ref):
            bad += 1
    print(f"{name:50s}: {bad} of 40 concurrent repetitions differ from the quiet result")
