"""Run each GEMM engine entry twice on the same inputs in a contraction mode and compare bit for bit.  MODE=f32e"""
import os, sys, ctypes, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import models
from vml_amd._lib import call, ptr, stream
V = models.vml_amd
lib = V._lib.load()
dev = torch.device("cuda:0")
V.set_gemm_mode(os.environ.get("MODE", "f32e"))
g = torch.Generator().manual_seed(0)
arr = lambda ts: (ctypes.c_void_p * len(ts))(*[t.data_ptr() for t in ts])
bad = 0
for (R, O, K) in [(2048, 512, 500), (100759, 512, 128), (4096, 128, 512), (1280, 2048, 512), (403036, 128, 128), (37, 20, 16), (2048, 384, 512)]:
    dy = torch.randn(R, O, generator=g).to(dev)
    x = torch.randn(R, K, generator=g).to(dev)
    W = torch.randn(O, K, generator=g).to(dev)
    WT = W.t().contiguous()
    nb = lib.smin_linear_rows_bwd_workspace_bytes(R, O, K)
    ws = torch.empty(nb + 64, dtype=torch.uint8, device=dev)
    outs = []
    for rep in range(3):
        y = torch.empty(R, O, device=dev)
        if K % 4 == 0 and O % 4 == 0:
            call("smin_linear_rows_fwd", stream(), arr([x]), 1, ptr(W), None, None, None, 1, R, O, K, ptr(y))
            dW, db, dx = torch.empty(O, K, device=dev), torch.empty(O, device=dev), torch.empty(R, K, device=dev)
            call("smin_linear_rows_bwd", stream(), ptr(dy), arr([x]), 1, ptr(WT), R, O, K, arr([dx]), ptr(dW), ptr(db), ptr(ws), ws.numel())
        else:
            call("smin_gemm_nt", stream(), ptr(x), ptr(W), ptr(y), R, O, K)
            dW = db = dx = y
        torch.cuda.synchronize()
        outs.append((y.clone(), dW.clone(), db.clone(), dx.clone()))
    for name, i in (("fwd", 0), ("dW", 1), ("dbias", 2), ("dX", 3)):
        same = all(torch.equal(outs[0][i], o[i]) for o in outs[1:])
        if not same:
            bad += 1
            d = max(float((outs[0][i] - o[i]).abs().max()) for o in outs[1:])
            print((R, O, K), name, "NOT reproducible, max diff", d)
print("done, non-reproducible outputs:", bad)
