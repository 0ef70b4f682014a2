#!/usr/bin/env python3
"""Headline benchmark: proposals/sec of one SMIN train step (zero_grad -> forward -> restated loss ->
backward [-> RCCL gradient all-reduce over xGMI when N > 1] -> Adam step) on synthetic
ActivityNet-Captions-shaped input (BASELINE.json configs[2]: activitynet.yml with T=256, batch 64 per GPU).

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Rank 0 prints ONE JSON line (contract in the task statement).  `value` = dense proposals (B_total * L * L)
per second with all inputs resident in HBM; `roofline` is measured live with HIP events around the dominant
kernel (the fp32-MFMA moment-unit GEMM) inside the timed steps; `cpu_baseline` times the CPU oracle
(a port of the reference's dense algorithm) on a bounded micro-batch on the host cores (rank 0, N=1 only).
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

WORKLOADS = {
    # name: (T, L, C, D, dl, layers, Din, Nq, H, per-GPU batch)      BASELINE.json configs / config/*.yml keys
    "activitynet_t256": (256, 64, 4, 512, 128, 3, 500, 20, 256, 64),
    "activitynet_yml": (128, 64, 4, 512, 128, 3, 500, 20, 256, 64),
    "charadessta": (64, 16, 4, 512, 128, 3, 1024, 13, 256, 32),
    "tacos": (128, 32, 4, 512, 128, 3, 4096, 14, 256, 2),
    "longvideo": (1024, 512, 4, 512, 128, 3, 500, 20, 256, 16),
}
PEAK_F32_MFMA_TFLOPS = 157.3      # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense
PEAK_HBM_GBS = 8000.0


def make_batch(B, T, L, Nq, Din, seed, device):
    """SURVEY 8d synthetic inputs: even samples full length, odd samples ragged; dataset-style masks."""
    import math
    g = torch.Generator().manual_seed(seed)
    vf = torch.randn(B, T, Din, generator=g)
    qf = torch.randn(B, Nq, 300, generator=g)
    vmask = torch.zeros(B, T, 1, dtype=torch.uint8)
    qmask = torch.zeros(B, Nq, 1, dtype=torch.uint8)
    lmask = torch.zeros(B, L, dtype=torch.bool)
    for b in range(B):
        if b % 2 == 0:
            nf, nq = T, Nq
        else:
            nf = int(torch.randint(max(1, T // 2), T, (1,), generator=g))
            nq = int(torch.randint(min(3, Nq - 1), Nq, (1,), generator=g))
        vf[b, nf:] = 0
        qf[b, nq:] = 0
        vmask[b, :nf] = 1
        qmask[b, :nq] = 1
        lmask[b, : math.ceil(nf / (T / L))] = True
    mmask = torch.triu(lmask.unsqueeze(2) & lmask.unsqueeze(1))
    sm = torch.rand(B, L, L, generator=g) * mmask
    ss, se = torch.rand(B, L, generator=g), torch.rand(B, L, generator=g)
    batch = dict(video_features=vf, video_mask=vmask, query_features=qf, query_mask=qmask, length_mask=lmask, moment_mask=mmask,
                 sm=sm, ym=sm > 0.5, ss=ss, ys=ss > 0.5, se=se, ye=se > 0.5, ya=torch.rand(B, L, generator=g) > 0.5)
    return {k: v.to(device) for k, v in batch.items()}


def cpu_baseline(cfg, seconds_budget=20.0):
    """Time the CPU oracle (oracle/smin_oracle.py, a port of the reference's dense algorithm, pinned to the
    reference by tests/golden) on a micro-batch of the same workload.  Checker code, used here only as the
    reported baseline."""
    from oracle import smin_oracle as O
    from tests import helpers as H
    T, L, C, D, dl, layers, Din, Nq, Hh, _ = cfg
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    threads = max(1, min(16, avail))                    # a 1-GPU box owns a 16-core share of the host
    torch.set_num_threads(threads)
    Bc = 2 if L >= 64 else 8
    sd = O.formula_state_dict(H.smin_shapes(T, L, C, D, dl, layers, Din, Nq, Hh), gain=1.0)
    sd = {k: v.requires_grad_(True) for k, v in sd.items()}
    batch = O.synthetic_batch(Bc, T, L, Nq, Din, seed=0)

    def step():
        for v in sd.values():
            v.grad = None
        pm, ps, pe, pa = O.smin_forward(sd, dict(T=T, L=L, C=C), *H.model_inputs(batch))
        O.loss_fn(pm, batch["ym"], batch["sm"], batch["moment_mask"], ps, batch["ys"], batch["ss"], pe, batch["ye"], batch["se"],
                  pa, batch["ya"], batch["length_mask"]).backward()

    t0 = time.perf_counter()
    step()                                              # warm-up (page-in, thread pool)
    first = time.perf_counter() - t0
    times = []
    while sum(times) + first < seconds_budget and len(times) < 5:
        t0 = time.perf_counter()
        step()
        times.append(time.perf_counter() - t0)
    t = min(times) if times else first
    return {"value": Bc * L * L / t, "unit": "proposals/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"{len(times) or 1} fwd+bwd steps of a B={Bc} micro-batch of the same workload, best step {t:.3f} s"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="activitynet_t256", choices=sorted(WORKLOADS))
    ap.add_argument("--batch", type=int, default=0, help="per-GPU batch (default: the workload's)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-optimizer", action="store_true", help="time fwd+bwd only")
    ap.add_argument("--gemm", default="f32", choices=["f32", "bf16x3"], help="arithmetic of the dense NT contractions")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world == 1 and args.gpus > 1:
        print("bench.py: --gpus N > 1 must be launched through torch.distributed.run", file=sys.stderr)
        sys.exit(2)
    assert torch.cuda.is_available(), "bench.py needs a HIP device (no CPU fallback)"
    local_rank = local_rank % max(torch.cuda.device_count(), 1)  # rehearsal: several ranks may share the one GPU of a test box
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    import models
    from vml_amd import distributed as dp, functional as Fn, loss_fn
    models.vml_amd._lib.load()
    models.vml_amd.set_gemm_mode(args.gemm)
    dp.init(backend=os.environ.get("SMIN_DIST_BACKEND", "nccl"), device=dev)   # nccl == RCCL on ROCm (xGMI inside the node)

    cfg = WORKLOADS[args.workload]
    T, L, C, D, dl, layers, Din, Nq, Hh, B = cfg
    if args.batch:
        B = args.batch
    torch.manual_seed(43)                                    # config/*.yml: seed 43, default initialisation
    model = models.SMIN(T, L, C, D, dl, layers, Din, Nq, Hh, dev).to(dev)
    if os.environ.get("SMIN_NO_OVERLAP"):
        model.overlap_boundary = False
    if os.environ.get("SMIN_NO_PREP_OVERLAP"):
        model.overlap_prep = False
    opt = torch.optim.Adam(model.parameters(), lr=5e-4)      # main.py:78-83, activitynet.yml lr
    net = dp.wrap(model, dev)                                 # DDP: bucketed gradient all-reduce overlapped with backward
    batch = make_batch(B, T, L, Nq, Din, seed=1000 + rank, device=dev)
    n_valid = int(batch["moment_mask"].sum().item())

    def step():
        opt.zero_grad(set_to_none=True)
        pm, ps, pe, pa = net(batch["video_features"], batch["video_mask"], batch["query_features"], batch["query_mask"],
                             batch["length_mask"], batch["moment_mask"])
        loss = loss_fn(pm, batch["ym"], batch["sm"], batch["moment_mask"], ps, batch["ys"], batch["ss"], pe, batch["ye"], batch["se"],
                       pa, batch["ya"], batch["length_mask"])
        loss.backward()
        if not args.no_optimizer:
            opt.step()
        return loss

    def fence():
        dp.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    Fn.KERNEL_EVENTS.clear()
    Fn.RECORD_EVENTS = True                                  # HIP events around the dominant kernel, on its stream
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    fence()
    elapsed = time.perf_counter() - t0
    Fn.RECORD_EVENTS = False
    elapsed = dp.max_over_ranks(elapsed, dev)                  # slowest rank defines the step
    n_valid_total = int(dp.sum_over_ranks(n_valid, dev))

    # dominant kernel: moment-unit forward GEMM  mu = X[N, 2D] @ Wcat[D, 2D]^T  (one launch per layer per step)
    durs = [s.elapsed_time(e) for (name, s, e) in Fn.KERNEL_EVENTS if name == "moment_unit_fwd"]
    roofline = None
    if durs:
        avg_ms = sum(durs) / len(durs)
        flops = 2.0 * n_valid * D * (2 * D)                  # algorithmic: 4*D^2 per valid cell
        achieved = flops / (avg_ms * 1e-3) / 1e12
        traffic = None
        tr_path = os.path.join(ROOT, "profiles", "pmc_moment_fwd.json")
        if os.path.exists(tr_path) and args.workload == "activitynet_t256" and B == cfg[-1] and args.gemm == "f32":
            try:                                            # PMC traffic was collected on exactly this workload
                traffic = json.load(open(tr_path)).get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        roofline = {"bound": "mfma", "kernel": "gemm_nt_kernel<CatMat,PlainMat,EpMomentOut> (moment-unit forward)",
                    "achieved": achieved, "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s", "frac": achieved / PEAK_F32_MFMA_TFLOPS,
                    "traffic": traffic, "avg_launch_ms": avg_ms, "launches_timed": len(durs),
                    "flops_per_launch": flops, "valid_cells_per_launch": n_valid}

    if rank == 0:
        total_B = B * world
        ms = elapsed / args.steps * 1e3
        out = {
            "metric": "proposals/sec (fwd+bwd)", "value": total_B * L * L / (elapsed / args.steps), "unit": "proposals/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{args.workload}: SMIN T={T} L={L} C={C} d={D} dl={dl} Nq={Nq} Din={Din} layers={layers}, "
                                   f"batch {B}/GPU, default init seed 43; step = zero_grad+fwd+restated loss+bwd"
                                   + ("" if args.no_optimizer else "+Adam") + ("+RCCL grad all-reduce (DDP)" if world > 1 else ""),
                       "global_batch": total_B, "valid_cells_per_step": n_valid_total,
                       "valid_cells_per_s": n_valid_total / (elapsed / args.steps), "parallelism": f"dp{world}",
                       "final_loss": float(loss.item())},
            "roofline": roofline,
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(cfg)
        print(json.dumps(out), flush=True)
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
