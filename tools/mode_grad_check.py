"""Gradients of the bench workload in a bf16-core contraction mode against the exact-fp32 run (same weights, same batch):
per-parameter max deviation relative to the largest gradient entry of that parameter and of the model.  MODE=f32e|bf16x3|bf16"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench, models
from vml_amd import loss_fn
dev = torch.device("cuda:0")
T, L, C, D, dl, layers, Din, Nq, Hh, B = bench.WORKLOADS[os.environ.get("WL", "activitynet_t256")]
B = int(os.environ.get("B", B))
torch.manual_seed(43)
model = models.SMIN(T, L, C, D, dl, layers, Din, Nq, Hh, dev).to(dev)
batch = bench.make_batch(B, T, L, Nq, Din, seed=1000, device=dev)


def grads(mode):
    models.vml_amd.set_gemm_mode(mode)
    try:
        model.zero_grad(set_to_none=True)
        pm, ps, pe, pa = model(batch["video_features"], batch["video_mask"], batch["query_features"], batch["query_mask"], batch["length_mask"], batch["moment_mask"])
        loss = loss_fn(pm, batch["ym"], batch["sm"], batch["moment_mask"], ps, batch["ys"], batch["ss"], pe, batch["ye"], batch["se"], pa, batch["ya"], batch["length_mask"])
        loss.backward()
        return float(loss), {k: p.grad.double().clone() for k, p in model.named_parameters()}
    finally:
        models.vml_amd.set_gemm_mode("f32")


l0, g0 = grads("f32")
l0b, g0b = grads("f32")
mode = os.environ.get("MODE", "f32e")
l1, g1 = grads(mode)
gmax = max(float(v.abs().max()) for v in g0.values())
print(f"loss f32 {l0:.9f} (repeat {l0b:.9f}) {mode} {l1:.9f}; largest gradient entry {gmax:.3e}")
rows = sorted(((float((g1[k] - g0[k]).abs().max()), k) for k in g0), reverse=True)
for err, k in rows[:8]:
    print(f"  {k:50s} max |dg| {err:.3e}  = {err / (float(g0[k].abs().max()) + 1e-30):.2e} of its own max, {err / gmax:.2e} of the model's")
assert all(torch.equal(g0[k], g0b[k]) for k in g0), "exact mode is not reproducible"
