"""Host time of each phase of the train step (no device synchronisation inside the step): how long the Python thread needs to
queue forward / loss / backward / optimizer, against the step's wall time.  WL=<workload> FUSED=0|1"""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench, models
from vml_amd import loss_fn
dev = torch.device("cuda:0")
T, L, C, D, dl, layers, Din, Nq, Hh, B = bench.WORKLOADS[os.environ.get("WL", "activitynet_t256")]
torch.manual_seed(43)
model = models.SMIN(T, L, C, D, dl, layers, Din, Nq, Hh, dev).to(dev)
model.fused_core = os.environ.get("FUSED", "1") != "0"
opt = torch.optim.Adam(model.parameters(), lr=5e-4, fused=True)
batch = bench.make_batch(B, T, L, Nq, Din, seed=1000, device=dev)
acc = [0.0] * 5


def step(rec):
    t = [time.perf_counter()]
    opt.zero_grad(set_to_none=True); t.append(time.perf_counter())
    pm, ps, pe, pa = model(batch["video_features"], batch["video_mask"], batch["query_features"], batch["query_mask"], batch["length_mask"], batch["moment_mask"])
    t.append(time.perf_counter())
    loss = loss_fn(pm, batch["ym"], batch["sm"], batch["moment_mask"], ps, batch["ys"], batch["ss"], pe, batch["ye"], batch["se"], pa, batch["ya"], batch["length_mask"])
    t.append(time.perf_counter())
    loss.backward(); t.append(time.perf_counter())
    opt.step(); t.append(time.perf_counter())
    if rec:
        for i in range(5):
            acc[i] += t[i + 1] - t[i]


for _ in range(8):
    step(False)
torch.cuda.synchronize()
n = 30
t0 = time.perf_counter()
for _ in range(n):
    step(True)
torch.cuda.synchronize()
wall = (time.perf_counter() - t0) / n * 1e3
names = ["zero_grad", "forward", "loss", "backward", "opt.step"]
print(f"{os.environ.get('WL', 'activitynet_t256')} fused={model.fused_core}: wall {wall:.3f} ms/step; host " +
      ", ".join(f"{k} {v / n * 1e3:.3f}" for k, v in zip(names, acc)) + f"  (sum {sum(acc) / n * 1e3:.3f} ms; forward includes the wait for the cell count)")
