"""How much would pipelining two half batches against each other buy?  The train step's forward + backward on the full batch
versus the same on two halves of the batch issued on two streams (one model, gradients accumulate), with no code changes:
the hardware overlaps the halves as it likes.  The halves share the library's side streams, so the overlap is partial; a
result well below the full-batch time would justify staggering halves inside the node.
    python tools/halves_probe.py"""
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
full = bench.make_batch(B, T, L, Nq, Din, seed=1000, device=dev)
IN = ["video_features", "video_mask", "query_features", "query_mask", "length_mask", "moment_mask"]


def part(lo, hi):
    return {k: v[lo:hi].contiguous() for k, v in full.items()}


def fb(b, n=None):
    model.known_cell_count = n
    pm, ps, pe, pa = model(*[b[k] for k in IN])
    loss = loss_fn(pm, b["ym"], b["sm"], b["moment_mask"], ps, b["ys"], b["ss"], pe, b["ye"], b["se"], pa, b["ya"], b["length_mask"])
    loss.backward()
    model.known_cell_count = None


def timeit(f, reps=12, warm=4):
    for _ in range(warm):
        model.zero_grad(set_to_none=True); f()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        model.zero_grad(set_to_none=True); f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


nfull = int(full["moment_mask"].sum())
print("full batch, one stream: %.3f ms" % timeit(lambda: fb(full, nfull)))
for parts in (2, 4):
    hs = [part(i * B // parts, (i + 1) * B // parts) for i in range(parts)]
    ns = [int(h["moment_mask"].sum()) for h in hs]
    streams = [torch.cuda.Stream() for _ in range(parts)]

    def seq():
        for h, n in zip(hs, ns):
            fb(h, n)

    def par():
        cur = torch.cuda.current_stream()
        for s in streams:
            s.wait_stream(cur)
        for h, n, s in zip(hs, ns, streams):
            with torch.cuda.stream(s):
                fb(h, n)
        for s in streams:
            cur.wait_stream(s)

    print("%d parts in sequence on one stream: %.3f ms" % (parts, timeit(seq)))
    print("%d parts on %d streams:             %.3f ms" % (parts, parts, timeit(par)))
