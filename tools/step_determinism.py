"""Run the train step's forward + backward several times on the same weights and batch and compare every gradient bit for bit,
for each combination of stream switches.  MODE=f32e WL=...  (a timing-dependent difference = a cross-stream race)"""
import os, sys, itertools
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
batch = bench.make_batch(B, T, L, Nq, Din, seed=1000, device=dev)
models.vml_amd.set_gemm_mode(os.environ.get("MODE", "f32e"))


def grads():
    model.zero_grad(set_to_none=True)
    pm, ps, pe, pa = model(batch["video_features"], batch["video_mask"], batch["query_features"], batch["query_mask"], batch["length_mask"], batch["moment_mask"])
    loss = loss_fn(pm, batch["ym"], batch["sm"], batch["moment_mask"], ps, batch["ys"], batch["ss"], pe, batch["ye"], batch["se"], pa, batch["ya"], batch["length_mask"])
    loss.backward()
    torch.cuda.synchronize()
    return [pm.detach().clone()] + [p.grad.clone() for p in model.parameters()]


names = ["pm"] + [k for k, _ in model.named_parameters()]
for fused, aw, ob, op in ([(1, 0, 0, 1), (1, 0, 1, 0)] if os.environ.get("VERBOSE") else [(1, 1, 1, 1), (1, 0, 1, 1), (1, 1, 0, 1), (1, 1, 1, 0), (1, 0, 0, 0), (0, 0, 1, 1), (0, 0, 0, 0)]):
    model.fused_core, model.async_weights, model.overlap_boundary, model.overlap_prep = bool(fused), bool(aw), bool(ob), bool(op)
    ref = grads()
    bad = {}
    for rep in range(int(os.environ.get("REPS", 6))):
        cur = grads()
        for n, a, b in zip(names, ref, cur):
            if not torch.equal(a, b):
                bad[n] = max(bad.get(n, 0.0), float((a - b).abs().max()))
    print(f"fused={fused} async_weights={aw} overlap_boundary={ob} overlap_prep={op}: {len(bad)} tensors differ between runs")
    if os.environ.get("VERBOSE"):
        print("   differ:", sorted(bad))
        print("   same:  ", [n for n in names if n not in bad])
