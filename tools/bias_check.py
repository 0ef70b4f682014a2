import sys, ctypes, torch
sys.path.insert(0, "/root/repo")
import models
from vml_amd._lib import call, ptr, stream
V = models.vml_amd
lib = V._lib.load()
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
for (R, O, K) in [(100759, 512, 128), (1280, 2048, 512), (403036, 128, 128)]:
    dy = torch.randn(R, O, generator=g).to(dev)
    x = torch.randn(R, K, generator=g).to(dev)
    ref = dy.double().sum(0).cpu()
    arr = lambda ts: (ctypes.c_void_p * len(ts))(*[t.data_ptr() for t in ts])
    nb = lib.smin_linear_rows_bwd_workspace_bytes(R, O, K)
    ws = torch.empty(nb + 64, dtype=torch.uint8, device=dev)
    for mode in ("f32", "f32e", "bf16x3", "bf16"):
        V.set_gemm_mode(mode)
        dW, db = torch.empty(O, K, device=dev), torch.empty(O, device=dev)
        call("smin_linear_rows_bwd", stream(), ptr(dy), arr([x]), 1, None, R, O, K, None, ptr(dW), ptr(db), ptr(ws), ws.numel())
        V.set_gemm_mode("f32")
        err = (db.cpu().double() - ref).abs().max().item()
        print((R, O, K), mode, "dbias max err", err, "rel to max |colsum|", err / ref.abs().max().item(), "sum|dy| scale", float(dy.abs().sum(0).max()))
