"""CPU, world_size 2, gloo: the N > 1 plumbing of bench.py / training -- batch sharding, the DDP wrapper, MAX / SUM
reductions over ranks (vml_amd/distributed.py).  The product has no CPU path (every module forward raises on CPU tensors),
so the network under DDP here is a stand-in torch module defined in this file, with a batch-mean-of-per-sample-means loss
of the same form as the restated loss: averaged shard gradients must equal full-batch gradients.  The
SMIN HIP path under DDP is tests/test_ddp_hip.py (-m gpu)."""
import os
import socket

import pytest
import torch
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


class _StandIn(torch.nn.Module):
    """Batch-independent stand-in with the hot path's input signature (video rows, mask, words, mask)."""

    def __init__(self, din, d):
        super().__init__()
        self.v, self.q = torch.nn.Linear(din, d), torch.nn.Linear(300, d)
        self.overlap_boundary, self.overlap_prep = True, True      # the flags distributed.wrap manages on SMIN

    def forward(self, vf, vm, qf, qm):
        fw = torch.tanh(self.q(qf)) * qm.float()
        fs = fw.sum(1) / qm.float().sum(1).clamp(min=1)
        return torch.tanh(self.v(vf)) * vm.float() * fs.unsqueeze(1), fs, fw


def _loss(model, batch):
    f, fs, fw = model(batch["video_features"], batch["video_mask"], batch["query_features"], batch["query_mask"])
    per_sample = f.square().mean(dim=(1, 2)) + fs.abs().mean(dim=1) + fw.mean(dim=(1, 2))
    return per_sample.mean()                     # batch mean of per-sample means, like main.py:100-106


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), WORLD_SIZE=str(world), RANK=str(rank), LOCAL_RANK=str(rank))
    torch.set_num_threads(2)
    import models
    from oracle import smin_oracle as O
    D = models.vml_amd.distributed
    D.init(backend="gloo")
    torch.manual_seed(7)
    model = _StandIn(24, 32)
    full = O.synthetic_batch(4, 16, 8, 6, 24, seed=3, with_labels=False)
    ref = _StandIn(24, 32)
    ref.load_state_dict(model.state_dict())
    _loss(ref, full).backward()
    net = D.wrap(model)
    assert net is not model and model.ddp_overrides == {"overlap_prep": False} and model.overlap_boundary   # CPU gloo keeps the rest
    assert D.describe() == ("gloo", 2)
    shard = D.shard_batch(full, rank, world)
    assert shard["video_features"].shape[0] == 2
    _loss(net, shard).backward()
    worst = 0.0
    for (k, p), (_, r) in zip(model.named_parameters(), ref.named_parameters()):
        worst = max(worst, (p.grad - r.grad).abs().max().item() / max(r.grad.abs().max().item(), 1e-9))
    tmax = D.max_over_ranks(1.0 + rank, torch.device("cpu"))
    tsum = D.sum_over_ranks(10 + rank, torch.device("cpu"))
    D.barrier()
    q.put((rank, worst, tmax, tsum))
    torch.distributed.destroy_process_group()


@pytest.mark.timeout(300)
def test_ddp_gradients_match_full_batch():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, worst, tmax, tsum in res:
        assert worst < 1e-5, (rank, worst)
        assert tmax == 2.0 and tsum == 21.0


def test_shard_batch_rejects_uneven_split():
    import models
    with pytest.raises(ValueError):
        models.vml_amd.distributed.shard_batch({"x": torch.zeros(5, 3)}, 0, 2)
