"""One rank of the data-parallel HIP-path check (launched by tests/conftest.py before the test process touches the GPU,
collected by tests/test_ddp_hip.py).  SURVEY 8e: averaged shard gradients under DistributedDataParallel must equal the
single-process full-batch gradients, and every rank must hold bitwise-equal gradients after the all-reduce.

    python tests/ddp_hip_worker.py <mode> <rank> <world> <port> <outdir>

mode "gloo2": 2 ranks sharing cuda:0, gloo backend (a one-GPU box cannot host two RCCL ranks on one device);
mode "rccl1": DDP over RCCL at world size 1 -- the two-stream step (boundary unit on the side stream) under DDP's hooks."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    mode, rank, world, port, outdir = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), sys.argv[4], sys.argv[5]
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=port, WORLD_SIZE=str(world), RANK=str(rank), LOCAL_RANK="0",
                      HSA_ENABLE_IPC_MODE_LEGACY="0")
    if mode == "rccl1":
        os.environ["SMIN_FORCE_DDP"] = "1"
    import torch
    import torch.distributed as dist
    import models
    from oracle import smin_oracle as O                      # test infrastructure: seeded inputs and weights only
    from tests import helpers as H
    from vml_amd import distributed as D, loss_fn

    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    models.vml_amd._lib.load()
    D.init(backend="gloo" if mode == "gloo2" else "nccl", device=dev)
    T, L, C, Dm, dl, layers, Din, Nq, Hh, B = 64, 16, 4, 128, 32, 3, 40, 9, 64, 4 * world
    sd = O.formula_state_dict(H.smin_shapes(T, L, C, Dm, dl, layers, Din, Nq, Hh), gain=1.2)
    full = {k: v.to(dev) for k, v in O.synthetic_batch(B, T, L, Nq, Din, seed=77).items()}
    # equal shards need equal per-sample weights in the loss: the restated loss is a batch mean of per-sample means

    def run(net, batch):
        pm, ps, pe, pa = net(*H.model_inputs(batch))
        loss = loss_fn(pm, batch["ym"], batch["sm"], batch["moment_mask"], ps, batch["ys"], batch["ss"], pe, batch["ye"], batch["se"],
                       pa, batch["ya"], batch["length_mask"])
        loss.backward()
        return float(loss)

    def fresh():
        m = models.SMIN(T, L, C, Dm, dl, layers, Din, Nq, Hh, dev)
        m.load_state_dict(sd)
        return m.to(dev)

    ref = fresh()
    loss_full = run(ref, full)
    model = fresh()
    flags_before = (model.overlap_boundary, model.overlap_prep)
    net = D.wrap(model, dev)
    assert net is not model, "wrap() returned the bare model: DDP was not engaged"
    shard = D.shard_batch(full, rank, world)
    loss_shard = run(net, shard)
    torch.cuda.synchronize()
    # relative to the parameter's own gradient scale; a parameter whose true gradient is zero (the key bias of a word
    # attention shifts every score of a row equally: softmax-invariant, its "gradient" is rounding noise) is held to the
    # scale of the whole gradient instead
    gmax = max(r.grad.abs().max().item() for r in ref.parameters())
    worst, worst_name = 0.0, ""
    for (k, p), (_, r) in zip(model.named_parameters(), ref.named_parameters()):
        e = (p.grad - r.grad).abs().max().item() / (r.grad.abs().max().item() + 1e-4 * gmax)
        if e > worst:
            worst, worst_name = e, k
    flat = torch.cat([p.grad.reshape(-1) for p in model.parameters()]).cpu()
    gathered = [torch.empty_like(flat) for _ in range(world)]
    if world > 1:
        dist.all_gather(gathered, flat)
    else:
        gathered = [flat]
    bitwise = all(torch.equal(g, gathered[0]) for g in gathered)
    mean_loss = D.sum_over_ranks(loss_shard, dev) / world
    loaded = sorted({ln.split()[-1] for ln in open("/proc/self/maps") if "libsmin_hip" in ln})
    res = dict(mode=mode, rank=rank, world=world, backend=dist.get_backend(), worst_rel=worst, worst_name=worst_name,
               bitwise_equal_across_ranks=bool(bitwise), loss_full=loss_full, loss_shard_mean=mean_loss,
               flags_before=flags_before, flags_after=(model.overlap_boundary, model.overlap_prep),
               overrides=model.ddp_overrides, native_so=loaded, n_params=int(flat.numel()))
    D.barrier()
    with open(os.path.join(outdir, f"{mode}_rank{rank}.json"), "w") as f:
        json.dump(res, f)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
