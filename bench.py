#!/usr/bin/env python3
"""Headline benchmark: proposals/sec of one SMIN train step (zero_grad -> forward -> restated loss ->
backward [-> RCCL gradient all-reduce over xGMI when N > 1] -> Adam step) on synthetic
ActivityNet-Captions-shaped input (BASELINE.json configs[2]: activitynet.yml with T=256, batch 64 per GPU).

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Rank 0 prints ONE JSON line (contract in the task statement).  `value` = dense proposals (B_total * L * L)
per second of the full train step (Adam included) with all inputs resident in HBM; `ms_fwd_bwd` is the same step
without the optimizer (SURVEY 8d's metric definition), timed right after.  `roofline` is measured live with HIP
events recorded by the library on the launch stream (smin_prof_*) around the three moment-unit contractions
(forward / input gradient / weight gradient) and the attention core inside the timed steps; `roofline.step` prices
the whole step against both roofs with SURVEY 8d's per-cell figures.  `cpu_baseline` times the CPU oracle (a port of
the reference's dense algorithm -- the reference itself cannot travel to the GPU box) on micro-batches of 8 on the
host cores (rank 0, N=1 only).
"""
import argparse
import json
import os
import sys
import time

os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")      # see video-moment-localization_amd/__init__.py (read when the HIP runtime initialises)

import torch  # noqa: E402

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


def cpu_baseline(cfg, seconds_budget=25.0):
    """Time the CPU oracle (oracle/smin_oracle.py, a port of the reference's dense algorithm, pinned to the
    reference by tests/golden) on micro-batches of the same workload, BASELINE.md section 3 style: micro-batch 8 (1 for
    the 512 x 512 map), the 16 host threads of a one-GPU box's CPU share, median step.  Checker code, used here only as the
    reported baseline; `kind` stays "port" because /root/reference does not exist on the GPU box."""
    import statistics
    from oracle import smin_oracle as O
    from tests import helpers as H
    T, L, C, D, dl, layers, Din, Nq, Hh, _ = cfg
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    # a one-GPU box owns a 16-core share of a 256-CPU host: the affinity mask still lists every CPU, and a 256-thread
    # torch pool on 16 cores ran this step 100x slower (318 s) than 16 threads do
    torch.set_num_threads(max(1, min(16, avail)))
    Bc = 8 if L <= 64 else 1
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
    while sum(times) + first < seconds_budget and len(times) < 9:
        t0 = time.perf_counter()
        step()
        times.append(time.perf_counter() - t0)
    t = statistics.median(times) if times else first
    return {"value": Bc * L * L / t, "unit": "proposals/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"{len(times) or 1} fwd+bwd steps of a B={Bc} micro-batch of the same workload after one warm-up, "
                      f"median step {t:.3f} s (min {min(times or [first]):.3f}), {torch.get_num_threads()} threads of {os.cpu_count()} host CPUs"}


def step_work(C, D, dl, Nq, layers):
    """Per-valid-cell work of one fwd+bwd step (fp32), SURVEY 8d.  Algorithmic = the reference's formulation; executed =
    what this build's content stream runs (the content unit's D x dl maps re-associated into the dl space)."""
    f_layer = 4 * C * D * dl + 2 * C * dl * dl + 4 * C * Nq * dl + 4 * C * C * dl + 4 * D * D
    flops_alg = 3 * (layers * f_layer + 2 * D)
    f_exec = 2 * D
    for k in range(layers):
        f_exec += 4 * D * D                              # moment unit: [x1 | mean_c f_c] Wcat^T
        f_exec += 2 * C * dl * dl * k                    # chat_k: history [cc_1 .. cc_k] (Wch Wc_l)^T
        f_exec += (2 * dl * D) if k > 0 else 0           # chat_k: gate term (sum hbar) Wch^T
        f_exec += 4 * C * Nq * dl + 4 * C * C * dl       # attention core with W_q / W_k folded per sample
        f_exec += 2 * dl * D                             # clip-mean update (mean_c cc) Wc^T
    U = C * D + D
    bytes_alg = 4 * ((U + layers * (2 * U + D) + D) + (layers * 3 * U + U + 2 * D))
    return flops_alg, 3 * f_exec, bytes_alg


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="activitynet_t256", choices=sorted(WORKLOADS))
    ap.add_argument("--batch", type=int, default=0, help="per-GPU batch (default: the workload's)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-optimizer", action="store_true", help="time fwd+bwd only")
    ap.add_argument("--graph", action="store_true", help="replay the step as one HIP graph (training.CapturedStep; the host-bound small workloads)")
    ap.add_argument("--no-other-modes", action="store_true", help="skip the extra timings of the bf16-core contraction modes")
    ap.add_argument("--gemm", "--dtype", dest="gemm", default=os.environ.get("SMIN_GEMM_MODE", "f32"), choices=["f32", "f32e", "bf16x3", "bf16"],
                    help="arithmetic of the dense contractions (f32 = the reference's, the headline; bf16 = BASELINE.json configs[1])")
    ap.add_argument("--feed", default="resident", choices=["resident", "host"],
                    help="resident (default, the headline): inputs live in HBM; host: every step's batch comes through BatchFeeder "
                         "(pinned host buffers -> async H2D on a copy stream -> masks/targets built on the device), the PCIe-inclusive rate")
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
    backend, world_seen = dp.describe()                      # what the collective library itself reports

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
    if os.environ.get("SMIN_SYNC_WEIGHTS"):                  # A/B switch: weight gradients on the main stream
        model.async_weights = False
    if os.environ.get("SMIN_NODE_GRAPH"):                    # A/B switch: one autograd node per module instead of the fused core
        model.fused_core = False
    # main.py:78-83, activitynet.yml lr.  fused=True: torch's single-launch implementation of the same update (the default on a GPU
    # is the multi-tensor "foreach" form: ~10 launches and ~0.7 ms of host time per step); SMIN_FOREACH_ADAM=1 restores it
    opt = torch.optim.Adam(model.parameters(), lr=5e-4, fused=not os.environ.get("SMIN_FOREACH_ADAM"), capturable=bool(args.graph))
    net = dp.wrap(model, dev)                                 # DDP: bucketed gradient all-reduce overlapped with backward
    batch = make_batch(B, T, L, Nq, Din, seed=1000 + rank, device=dev)
    n_valid = int(batch["moment_mask"].sum().item())
    feed_iter = None
    if args.feed == "host":
        import itertools
        gh = torch.Generator().manual_seed(7 + rank)
        hosts = []
        for s_ in range(2):                                   # two distinct host batches, cycled
            hb = make_batch(B, T, L, Nq, Din, seed=1000 + rank + 17 * s_, device="cpu")
            dur = torch.rand(B, generator=gh) * 100 + 20
            ts = torch.rand(B, generator=gh) * dur * 0.5
            # (pinned, as a DataLoader with pin_memory=True hands them over: the feeder then stages nothing on the host)
            hosts.append(dict(video_features=hb["video_features"].pin_memory(), query_features=hb["query_features"].pin_memory(), nfeats=hb["video_mask"].sum((1, 2)),
                              qlen=hb["query_mask"].sum((1, 2)), times=torch.stack([ts, ts + 1.0 + torch.rand(B, generator=gh) * (dur - ts - 1.0)], 1), duration=dur))
        feeder = models.vml_amd.BatchFeeder(T, L, Nq, dev)
        feed_iter = feeder.feed(itertools.cycle(hosts))

    captured = {}
    if args.graph:
        assert world == 1 and feed_iter is None, "--graph: one GPU, resident inputs"
        captured = {True: models.vml_amd.CapturedStep(model, None if args.no_optimizer else opt), False: models.vml_amd.CapturedStep(model, None)}

    def step(with_opt=True, eager=False):
        nonlocal batch
        if captured and not eager:
            return captured[with_opt and not args.no_optimizer](batch)[0]
        if feed_iter is not None:
            batch = next(feed_iter)
            batch["sm"] = torch.nan_to_num(batch["sm"])       # 0/0 IoU of degenerate windows (the reference has the same NaNs)
        opt.zero_grad(set_to_none=True)
        pm, ps, pe, pa = net(batch["video_features"], batch["video_mask"], batch["query_features"], batch["query_mask"],
                             batch["length_mask"], batch["moment_mask"])
        loss = loss_fn(pm, batch["ym"], batch["sm"], batch["moment_mask"], ps, batch["ys"], batch["ss"], pe, batch["ye"], batch["se"],
                       pa, batch["ya"], batch["length_mask"])
        loss.backward()
        if with_opt and not args.no_optimizer:
            opt.step()
        return loss

    def fence():
        dp.barrier()
        torch.cuda.synchronize()

    lib = models.vml_amd._lib
    for _ in range(args.warmup):
        step()
    lib.prof_enable(True)                                    # HIP events around the tagged launches, on their launch stream
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    fence()
    elapsed = time.perf_counter() - t0
    lib.prof_enable(False)
    prof = lib.prof_read()
    if captured:                                              # a replay runs no host code, so no launch is bracketed: kernel timings from eager steps
        lib.prof_enable(True)
        for _ in range(5):
            step(with_opt=False, eager=True)
        fence()
        lib.prof_enable(False)
        prof = lib.prof_read()
    elapsed = dp.max_over_ranks(elapsed, dev)                  # slowest rank defines the step
    # the same K steps without the optimizer: SURVEY 8d defines the metric on zero_grad + fwd + loss + bwd (+ all-reduce)
    if captured:
        step(with_opt=False)                                  # (captures that variant's graph outside the timed region)
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step(with_opt=False)
    fence()
    elapsed_fb = dp.max_over_ranks(time.perf_counter() - t0, dev)
    # kernel-quality timings of the backward kernels: in the timed region they share the chip with the weight-gradient stream, so
    # their event durations there measure the overlap, not the kernel; a short pass with that stream folded into the main one
    prof_serial = {}
    if getattr(model, "async_weights", False) and getattr(model, "fused_core", False):
        model.async_weights = False
        step(with_opt=False, eager=True)
        lib.prof_enable(True)
        for _ in range(max(2, args.steps // 4)):
            step(with_opt=False, eager=True)
        fence()
        lib.prof_enable(False)
        prof_serial = lib.prof_read()
        model.async_weights = True
    n_valid_total = int(dp.sum_over_ranks(n_valid, dev))

    # the same step with the contractions on the bf16 matrix cores (never the headline: `value` above is the exact-fp32 run)
    other = {}
    # (not under the in-node gradient exchange: it is limited to the exact-fp32 mode, distributed.wrap would have to be redone per mode)
    if args.gemm == "f32" and not args.no_other_modes and not getattr(model, "grad_sync", False):
        def scores():
            with torch.no_grad():
                return [o.float().clone() for o in net(batch["video_features"], batch["video_mask"], batch["query_features"], batch["query_mask"],
                                                         batch["length_mask"], batch["moment_mask"])]
        notes = {"f32e": "fp32 emulated: exact 3-way bf16 split of each operand, 6 products, fp32 accumulate (error vs fp64 at the exact-fp32 engine's level)",
                 "bf16x3": "2-way bf16 split, 3 products, fp32 accumulate (~1e-5 relative on a dot product)"}
        for mode in ("f32e", "bf16x3"):
            ref_scores = scores()                              # exact fp32, current weights
            models.vml_amd.set_gemm_mode(mode)
            try:
                dev_max = max(float((a - b).abs().max()) for a, b in zip(scores(), ref_scores))
                for _ in range(3):
                    step()
                fence()
                t0 = time.perf_counter()
                for _ in range(args.steps):
                    step()
                fence()
                el = dp.max_over_ranks(time.perf_counter() - t0, dev)
            finally:
                models.vml_amd.set_gemm_mode("f32")
            other[mode] = {"ms_per_step": el / args.steps * 1e3, "value": B * world * L * L / (el / args.steps), "unit": "proposals/s",
                           "max_score_deviation_from_f32": dev_max, "arithmetic": notes[mode]}

    def moment_kernels_alone():
        """The three moment-unit contractions at this step's size, each alone on an otherwise idle device (after a warm-up pass):
        what the kernels reach without the rest of the step around them (cache state, clocks under sustained mixed load)."""
        from vml_amd._lib import call, ptr, stream
        libc = lib.load()
        lay = models.vml_amd.CellLayout.begin(batch["moment_mask"]).finish()
        N = lay.N
        gd = torch.Generator(device=dev).manual_seed(0)
        r = lambda *sh: torch.randn(*sh, generator=gd, device=dev)
        fcm, fm_, fb_, x1, dmu = r(N, D), r(N, D), r(B, L, D), r(N, D), r(N, D)
        W, bvec = r(D, 2 * D) * 0.03, r(D)
        WT = W.t().contiguous()
        mu, dfc, dfb = torch.empty(N, D, device=dev), torch.empty(N, D, device=dev), torch.empty(B, L, D, device=dev)
        dW, db = torch.empty_like(W), torch.empty(D, device=dev)
        ws = torch.empty(libc.smin_workspace_bytes(N, B, 4, D, 4, 1), dtype=torch.uint8, device=dev)
        fns = {"moment_fwd": lambda: call("smin_moment_unit_fwd", stream(), ptr(fcm), ptr(fm_), ptr(fb_), ptr(lay.cells), N, B, L, D, ptr(W), ptr(bvec), ptr(mu), ptr(x1)),
               "moment_dx": lambda: call("smin_moment_unit_bwd", stream(), ptr(dmu), ptr(fcm), ptr(fb_), ptr(lay.cells), ptr(lay.row_ptr), ptr(lay.cellmap), N, B, L, D,
                                         ptr(WT), ptr(dfc), ptr(dfb), None, None, ptr(ws), ws.numel(), 1, None, ptr(x1), None),
               "moment_dw": lambda: call("smin_moment_unit_bwd", stream(), ptr(dmu), ptr(fcm), ptr(fb_), ptr(lay.cells), ptr(lay.row_ptr), ptr(lay.cellmap), N, B, L, D,
                                         ptr(WT), None, None, ptr(dW), ptr(db), ptr(ws), ws.numel(), 1, None, ptr(x1), None)}
        for _ in range(12):                                     # warm-up of all three, interleaved: ~35 ms of matrix work (the first
            for f in fns.values():                              # launches after a pause run ~15 % slow, tools/nt_tail.py)
                f()
        out = {}
        lib.prof_enable(True)
        for f in fns.values():
            for _ in range(10):
                f()
        torch.cuda.synchronize()
        lib.prof_enable(False)
        rec = lib.prof_read()
        for k in fns:
            durs = rec.get(k, [])
            if durs:
                out[k] = sum(durs) / len(durs)
        return out

    alone_ms = {}
    if rank == 0 and world == 1 and args.workload != "longvideo":
        try:
            alone_ms = moment_kernels_alone()
        except Exception as e:                                   # a diagnostic, never fatal
            alone_ms = {}
            print(f"# moment_kernels_alone failed: {e}", file=sys.stderr)

    def kernel_line(tag, name, flops):
        durs = prof.get(tag, [])
        if not durs:
            return None
        in_step_ms = sum(durs) / len(durs)
        alone = prof_serial.get(tag, []) if tag != "moment_fwd" else []      # the forward contraction runs alone in the timed region
        avg_ms = sum(alone) / len(alone) if alone else in_step_ms
        ach = flops / (avg_ms * 1e-3) / 1e12
        line = {"kernel": name, "achieved": ach, "frac": ach / PEAK_F32_MFMA_TFLOPS, "avg_launch_ms": avg_ms, "launches_timed": len(alone) if alone else len(durs),
                "flops_per_launch": flops}
        if tag in alone_ms:
            line["alone_ms"] = alone_ms[tag]
            line["alone_tflops"] = flops / (alone_ms[tag] * 1e-3) / 1e12
        if alone:
            line["avg_launch_ms_in_timed_region"] = in_step_ms
            line["note"] = "avg_launch_ms: pass without the concurrent weight-gradient stream; in the timed region the two streams share the chip"
        return line

    # dominant kernels: the three moment-unit contractions (one launch each per layer per step), 4*D^2 FLOP per valid cell each
    mu_flops = 2.0 * n_valid * D * (2 * D)
    gemms = {"moment_fwd": kernel_line("moment_fwd", "gemm_nt_kernel<CatMat,PlainMat,EpMomentOut>  mu = [x1|mean_c f_c] Wcat^T", mu_flops),
             "moment_dx": kernel_line("moment_dx", "gemm_nt_kernel<PlainMat,PlainMat,EpSplitStore>  dX = dmu Wcat", mu_flops),
             "moment_dw": kernel_line("moment_dw", "gemm_tn_kernel<PlainMat,CatMat>  dWcat = dmu^T [x1|mean_c f_c]", mu_flops)}
    pmc = {}
    pmc_path = os.path.join(ROOT, "profiles", "pmc_moment.json")
    if os.path.exists(pmc_path) and args.workload == "activitynet_t256" and B == cfg[-1] and args.gemm == "f32":
        try:                                                # PMC traffic (rocprofv3 --pmc, gfx950 corrections) of exactly this workload
            pmc = json.load(open(pmc_path))
            import hashlib                                   # ... measured on the GEMM engine as built from these sources: stale -> no traffic figure
            csrc = os.path.join(ROOT, "video-moment-localization_amd", "csrc")
            stamp = pmc.get("sources_sha256") or {}
            if not stamp or any(hashlib.sha256(open(os.path.join(csrc, f), "rb").read()).hexdigest() != h for f, h in stamp.items()):
                pmc = {"stale": "profiles/pmc_moment.json was measured on other gemm.h / moment_unit.hip sources (or carries no stamp): traffic omitted; "
                                "regenerate with profiles/run_pmc.sh + profiles/make_pmc_moment.py"}
        except Exception:
            pmc = {}
    roofline = None
    if gemms["moment_fwd"]:
        g = gemms["moment_fwd"]
        roofline = {"bound": "mfma", "kernel": g["kernel"], "achieved": g["achieved"], "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s",
                    "frac": g["frac"], "traffic": pmc.get("moment_fwd", {}).get("hbm_bytes_per_launch"), "avg_launch_ms": g["avg_launch_ms"],
                    "launches_timed": g["launches_timed"], "flops_per_launch": mu_flops, "valid_cells_per_launch": n_valid,
                    "algorithmic_bytes_per_launch": 4.0 * n_valid * 4 * D}
        if "stale" in pmc:
            roofline["traffic_note"] = pmc["stale"]
        if args.gemm == "bf16":                                # bf16 products run at 16x the fp32 matrix rate: the contraction is bound by its operand traffic
            ach = roofline["algorithmic_bytes_per_launch"] / (g["avg_launch_ms"] * 1e-3) / 1e9
            roofline.update({"bound": "hbm", "achieved": ach, "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": ach / PEAK_HBM_GBS, "traffic": None,
                             "tflops": g["achieved"]})
        for k in ("moment_dx", "moment_dw"):
            if gemms[k]:
                gemms[k]["traffic"] = pmc.get(k, {}).get("hbm_bytes_per_launch")
        roofline["moment_gemms"] = {k: v for k, v in gemms.items() if v}
        attn = {}
        # attention core: HBM-side view.  fwd reads chat, writes cc (+ mean); bwd reads chat + 1-2 gradients, writes dchat
        rows = n_valid * C
        for tag, nbytes in (("attn_fwd", 4.0 * dl * (2 * rows + n_valid)), ("attn_bwd", 4.0 * dl * (3 * rows + n_valid))):
            durs = prof_serial.get(tag, []) if tag == "attn_bwd" and prof_serial.get(tag) else prof.get(tag, [])
            if durs:
                avg_ms = sum(durs) / len(durs)
                attn[tag] = {"avg_launch_ms": avg_ms, "launches_timed": len(durs), "algorithmic_bytes_per_launch": nbytes,
                             "achieved_GBs": nbytes / (avg_ms * 1e-3) / 1e9, "frac_hbm": nbytes / (avg_ms * 1e-3) / 1e9 / PEAK_HBM_GBS}
        roofline["attention_core"] = attn
        fa, fe, ba = step_work(C, D, dl, Nq, layers)
        t_fb = elapsed_fb / args.steps
        per_rank_valid = n_valid_total / world
        roofline["step"] = {"valid_cells": n_valid_total, "flops_algorithmic_per_cell": fa, "flops_executed_per_cell": fe,
                            "bytes_algorithmic_per_cell": ba, "timed_on": "ms_fwd_bwd",
                            "frac_mfma": per_rank_valid * fa / t_fb / 1e12 / PEAK_F32_MFMA_TFLOPS,
                            "frac_mfma_executed": per_rank_valid * fe / t_fb / 1e12 / PEAK_F32_MFMA_TFLOPS,
                            "frac_hbm": per_rank_valid * ba / t_fb / 1e9 / PEAK_HBM_GBS,
                            "bound": "mfma (fp32: ~110 FLOP/B is above the 20 FLOP/B ridge)" if args.gemm != "bf16"
                                     else "hbm (bf16 products: the 312 FLOP/B ridge is above the path's ~110 FLOP/B)"}

    if rank == 0:
        total_B = B * world
        ms = elapsed / args.steps * 1e3
        exchange = ""
        if world_seen > 1 or backend is not None:
            lib_name = {"nccl": "RCCL (torch backend nccl)", "gloo": "gloo (host-staged; NOT RCCL)"}.get(backend, str(backend))
            how = {"in_node": "issued by the one-node backward as gradient groups become final, overlapped with it",
                   "torch_ddp": "torch DistributedDataParallel buckets"}.get(getattr(model, "grad_exchange", None), "none")
            exchange = f"+gradient all-reduce ({how}) over {lib_name}, process group of {world_seen}"
        out = {
            "metric": "proposals/sec (fwd+bwd)", "value": total_B * L * L / (elapsed / args.steps), "unit": "proposals/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms, "ms_with_adam": None if args.no_optimizer else ms,
            "ms_fwd_bwd": elapsed_fb / args.steps * 1e3, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None,
            "dtype": {"f32": "f32", "f32e": "f32 (emulated on the bf16 matrix cores: exact 3-way split, 6 products, fp32 accumulate)", "bf16x3": "f32 (split-bf16 products, fp32 accumulate)", "bf16": "bf16 (products, and storage of the contraction-only tensors: pair product, attention outputs; fp32 accumulate, fp32 storage of everything an element-wise kernel reads)"}[args.gemm],
            "data": "synthetic" if args.feed == "resident" else "synthetic, fed from pinned host memory every step (BatchFeeder: async H2D + device-side targets)",
            "config": {"workload": f"{args.workload}: SMIN T={T} L={L} C={C} d={D} dl={dl} Nq={Nq} Din={Din} layers={layers}, "
                                   f"batch {B}/GPU, default init seed 43; step = zero_grad+fwd+restated loss+bwd"
                                   + ("" if args.no_optimizer else "+Adam (torch fused)" if not os.environ.get("SMIN_FOREACH_ADAM") else "+Adam (torch foreach)") + exchange
                                   + (", replayed as one HIP graph per (shapes, valid-cell count) with one scalar device read ahead of each step (training.CapturedStep)" if args.graph else ""),
                       "global_batch": total_B, "valid_cells_per_step": n_valid_total,
                       "valid_cells_per_s": n_valid_total / (elapsed / args.steps), "parallelism": f"dp{world}", "dist_backend": backend, "dist_world_size": world_seen,
                       "ddp_overrides": getattr(model, "ddp_overrides", None), "grad_exchange": getattr(model, "grad_exchange", None),
                       "final_loss": float(loss.item())},
            "roofline": roofline,
        }
        if other:
            out["other_arithmetic"] = other
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(cfg)
        print(json.dumps(out), flush=True)
    if torch.distributed.is_initialized():
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
