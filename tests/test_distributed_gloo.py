"""CPU, world_size 2, gloo: the N > 1 path of bench.py / training -- batch sharding + DDP gradient all-reduce.
The HIP kernels cannot run here, so the model under DDP is the torch-only part of the drop-in (Backbone) with a
per-sample-mean loss of the same form as the restated loss: averaged shard gradients must equal
full-batch gradients, timings reduce with MAX, counts with SUM."""
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
    model = models.Backbone(16, 32, 24, 6, 16)
    full = O.synthetic_batch(4, 16, 8, 6, 24, seed=3, with_labels=False)
    ref = models.Backbone(16, 32, 24, 6, 16)
    ref.load_state_dict(model.state_dict())
    _loss(ref, full).backward()
    net = D.wrap(model)
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
