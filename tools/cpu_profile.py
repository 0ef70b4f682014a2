"""Host-side profile of the train step (where does the Python thread spend its time while the GPU runs?)."""
import cProfile, os, pstats, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench, models
from vml_amd import loss_fn
dev = torch.device("cuda:0")
T, L, C, D, dl, layers, Din, Nq, Hh, B = bench.WORKLOADS[os.environ.get("WL", "activitynet_t256")]
torch.manual_seed(43)
model = models.SMIN(T, L, C, D, dl, layers, Din, Nq, Hh, dev).to(dev)
opt = torch.optim.Adam(model.parameters(), lr=5e-4, fused=True)
batch = bench.make_batch(B, T, L, Nq, Din, seed=1000, device=dev)


def step():
    opt.zero_grad(set_to_none=True)
    pm, ps, pe, pa = model(batch["video_features"], batch["video_mask"], batch["query_features"], batch["query_mask"], batch["length_mask"], batch["moment_mask"])
    loss = loss_fn(pm, batch["ym"], batch["sm"], batch["moment_mask"], ps, batch["ys"], batch["ss"], pe, batch["ye"], batch["se"], pa, batch["ya"], batch["length_mask"])
    loss.backward()
    opt.step()


for _ in range(5):
    step()
torch.cuda.synchronize()
# host time to enqueue one step's forward when the GPU is kept out of the way
t0 = time.perf_counter()
for _ in range(10):
    step()
t_enq = time.perf_counter() - t0
torch.cuda.synchronize()
t_all = time.perf_counter() - t0
print(f"10 steps: host returned after {t_enq*100:.2f} ms/step, GPU done after {t_all*100:.2f} ms/step")
pr = cProfile.Profile()
pr.enable()
for _ in range(10):
    step()
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(22)
