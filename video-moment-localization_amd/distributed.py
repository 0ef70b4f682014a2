"""Data-parallel plumbing for the train step: one process per GPU, batch sharded across ranks,
gradients all-reduced by torch DistributedDataParallel over RCCL (backend "nccl" on ROCm; "gloo" in CPU tests).
The hot path itself has no exchange step -- samples are independent -- so this is the only collective."""
import logging
import os

import torch
import torch.distributed as dist

log = logging.getLogger("vml_amd.distributed")


def env_world():
    return int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0"))


def init(backend=None, device=None):
    """Initialise the default process group from the torchrun environment (MASTER_ADDR should be 127.0.0.1)."""
    world, rank, _ = env_world()
    if (world == 1 and not os.environ.get("SMIN_FORCE_DDP")) or dist.is_initialized():
        return
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29541")
    os.environ.setdefault("RANK", "0")
    os.environ.setdefault("WORLD_SIZE", "1")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if backend is None:
        backend = "nccl" if torch.cuda.is_available() else "gloo"
    kwargs = {"device_id": device} if (backend == "nccl" and device is not None) else {}
    dist.init_process_group(backend, **kwargs)


def shard_batch(batch, rank, world):
    """Contiguous equal split of every tensor's leading (sample) dimension; B must divide by world."""
    out = {}
    for k, v in batch.items():
        B = v.shape[0]
        if B % world:
            raise ValueError(f"batch of {B} does not split over {world} ranks")
        per = B // world
        out[k] = v[rank * per:(rank + 1) * per]
    return out


def wrap(model, device=None, bucket_cap_mb=8):
    """DistributedDataParallel with small buckets so the all-reduce overlaps the backward kernels
    (28-36 MB of fp32 gradients per step: SURVEY 5)."""
    if not dist.is_initialized() or (dist.get_world_size() == 1 and not os.environ.get("SMIN_FORCE_DDP")):
        return model
    changed = {}
    gloo_on_gpu = dist.get_backend() == "gloo" and device is not None and device.type == "cuda"
    if gloo_on_gpu and getattr(model, "overlap_boundary", False):
        # gloo stages every bucket through the host and synchronises the streams a gradient touched: with the two-stream
        # step that serialises the whole backward pass (8x slower, measured); RCCL is unaffected
        model.overlap_boundary = changed["overlap_boundary"] = False
    one_node = getattr(model, "fused_core", False) and getattr(model, "native_host", False) and not os.environ.get("SMIN_DDP_NO_PREP_OVERLAP")
    if getattr(model, "overlap_prep", False) and (gloo_on_gpu or not one_node):
        # node-per-module graph: DDP creates every parameter's gradient accumulator on the stream it is constructed on; parameters
        # that the step touches only on the second stream then make the main stream wait at each accumulation (measured +0.7 ms
        # over keeping that work on the main stream; the boundary unit's overlap still pays).  The one-node step hands every
        # gradient over on the main stream and keeps its second-stream tail under RCCL.
        model.overlap_prep = changed["overlap_prep"] = False
    model.ddp_overrides = changed                      # what wrap() switched off, for the caller to report
    if changed:
        log.warning("distributed.wrap (%s, world %d): set %s on the model", dist.get_backend(), dist.get_world_size(),
                    ", ".join(f"{k}={v}" for k, v in changed.items()))
    from torch.nn.parallel import DistributedDataParallel as DDP
    ids = [device.index] if (device is not None and device.type == "cuda") else None
    return DDP(model, device_ids=ids, bucket_cap_mb=bucket_cap_mb, gradient_as_bucket_view=True)


def describe():
    """(backend, world size) of the default process group as the collective library reports them -- what a benchmark
    line may claim about its gradient exchange."""
    if not dist.is_initialized():
        return None, 1
    return dist.get_backend(), dist.get_world_size()


def max_over_ranks(value, device):
    if not dist.is_initialized():
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def sum_over_ranks(value, device):
    if not dist.is_initialized():
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t)
    return float(t.item())


def barrier():
    if dist.is_initialized():
        dist.barrier()
