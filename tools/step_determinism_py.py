"""As step_determinism.py on the Python host, with hooks on the backbone outputs (f, fs, fw): which of their gradients is not
reproducible from run to run?  MODE=f32e"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench, models
from vml_amd import loss_fn
dev = torch.device("cuda:0")
T, L, C, D, dl, layers, Din, Nq, Hh, B = bench.WORKLOADS[os.environ.get("WL", "longvideo")]
B = int(os.environ.get("B", 2))
torch.manual_seed(43)
model = models.SMIN(T, L, C, D, dl, layers, Din, Nq, Hh, dev).to(dev)
model.native_host = False
model.overlap_prep = False
batch = bench.make_batch(B, T, L, Nq, Din, seed=1000, device=dev)
models.vml_amd.set_gemm_mode(os.environ.get("MODE", "f32e"))
cap = {}
orig = model.backbone.forward


def wrapped(*a, **k):
    f, fs, fw = orig(*a, **k)
    for name, t in (("f", f), ("fs", fs), ("fw", fw)):
        t.register_hook(lambda g, name=name: cap.__setitem__(name, g.detach().clone()))
    return f, fs, fw


model.backbone.forward = wrapped


def grads():
    model.zero_grad(set_to_none=True)
    cap.clear()
    pm, ps, pe, pa = model(batch["video_features"], batch["video_mask"], batch["query_features"], batch["query_mask"], batch["length_mask"], batch["moment_mask"])
    loss = loss_fn(pm, batch["ym"], batch["sm"], batch["moment_mask"], ps, batch["ys"], batch["ss"], pe, batch["ye"], batch["se"], pa, batch["ya"], batch["length_mask"])
    loss.backward()
    torch.cuda.synchronize()
    out = dict(cap)
    for k, p in model.named_parameters():
        out[k] = p.grad.clone()
    return out


for ob in (1, 0):
    model.overlap_boundary = bool(ob)
    ref = grads()
    bad = {}
    for rep in range(6):
        cur = grads()
        for n in ref:
            if not torch.equal(ref[n], cur[n]):
                bad[n] = max(bad.get(n, 0.0), float((ref[n] - cur[n]).abs().max()))
    print(f"python host, overlap_boundary={ob}: differ:", {k: v for k, v in bad.items() if k in ("f", "fs", "fw")}, "params differing:", len([k for k in bad if k not in ("f", "fs", "fw")]))
    if os.environ.get("VERBOSE"):
        print("   same:", [k for k in ref if k not in bad])
