"""Probe: does capturing the query encoder (padded BiLSTM, two streams) in a HIP graph work, and what does it save?"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import models

dev = torch.device("cuda:0")
torch.manual_seed(0)
qe = models.QueryEncoder(20, 256).to(dev)
B, Nq = 64, 20
x = torch.randn(B, Nq, 300, device=dev)
mask = torch.ones(B, Nq, 1, dtype=torch.uint8, device=dev)
mask[1::2, 13:] = 0


def run(mod, n=30):
    for _ in range(5):
        fs, fw = mod(x, mask)
        (fs.sum() + fw.sum()).backward()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fs, fw = mod(x, mask)
        (fs.sum() + fw.sum()).backward()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


print("eager  fwd+bwd ms:", run(qe))
torch.autograd.graph.set_warn_on_accumulate_grad_stream_mismatch(False)
with torch.backends.cudnn.flags(enabled=False):
    print("native (no MIOpen) fwd+bwd ms:", run(qe))
    fs0, fw0 = qe(x, mask)
    g0 = torch.autograd.grad(fs0.sum() + (fw0 * fw0).sum(), list(qe.parameters()))
    sys.stdout.flush()
    gq = torch.cuda.make_graphed_callables(qe, (x, mask), allow_unused_input=True)
    print("graphed native fwd+bwd ms:", run(gq))
    fs1, fw1 = gq(x, mask)
    g1 = torch.autograd.grad(fs1.sum() + (fw1 * fw1).sum(), list(qe.parameters()))
    print("max diff out", (fs0 - fs1).abs().max().item(), (fw0 - fw1).abs().max().item(),
          "grads", max((a - b).abs().max().item() for a, b in zip(g0, g1)))
