"""A contraction repeated on one stream while another stream keeps the device busy with other contractions: are the repeated
results bit-identical?  MODE=f32e"""
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
side = torch.cuda.Stream()
main = torch.cuda.current_stream()
sp = lambda s: ctypes.c_void_p(s.cuda_stream)
# background work
Rb, Ob, Kb = 100000, 512, 1024
xb, Wb, yb = torch.randn(Rb, Kb, generator=g).to(dev), torch.randn(Ob, Kb, generator=g).to(dev), torch.empty(Rb, Ob, device=dev)
for (R, O, K, what) in [(2048, 384, 512, "dx"), (2048, 384, 512, "fwd"), (2048, 384, 512, "dw"), (40, 512, 512, "fwd"), (40, 512, 512, "dw"), (1024, 512, 512, "dx"), (100759, 128, 512, "dx"), (100759, 128, 512, "dw")]:
    dy = torch.randn(R, O, generator=g).to(dev)
    x = torch.randn(R, K, generator=g).to(dev)
    W = torch.randn(O, K, generator=g).to(dev)
    WT = W.t().contiguous()
    nb = lib.smin_linear_rows_bwd_workspace_bytes(R, O, K)
    ws = torch.empty(nb + 64, dtype=torch.uint8, device=dev)
    outs = []
    torch.cuda.synchronize()
    for rep in range(12):
        if rep % 2 == 1:                                     # every other repetition runs beside the background stream
            for _ in range(3):
                call("smin_linear_rows_fwd", sp(side), arr([xb]), 1, ptr(Wb), None, None, None, 1, Rb, Ob, Kb, ptr(yb))
        if what == "fwd":
            o = torch.empty(R, O, device=dev)
            call("smin_linear_rows_fwd", sp(main), arr([x]), 1, ptr(W), None, None, None, 1, R, O, K, ptr(o))
        elif what == "dx":
            o = torch.empty(R, K, device=dev)
            call("smin_linear_rows_bwd", sp(main), ptr(dy), arr([x]), 1, ptr(WT), R, O, K, arr([o]), None, None, ptr(ws), ws.numel())
        else:
            o = torch.empty(O, K, device=dev)
            db = torch.empty(O, device=dev)
            call("smin_linear_rows_bwd", sp(main), ptr(dy), arr([x]), 1, None, R, O, K, None, ptr(o), ptr(db), ptr(ws), ws.numel())
        torch.cuda.synchronize()
        outs.append(o.clone())
    diffs = [float((outs[0] - o).abs().max()) for o in outs[1:]]
    print((R, O, K), what, "max diff over repetitions:", max(diffs), "first differing rep:", next((i + 1 for i, d in enumerate(diffs) if d > 0), None))
