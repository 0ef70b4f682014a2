"""Caller-side pieces of the reference that the train step needs: the loss and the
R@n/IoU metric, restated so that the unmodified training loop becomes runnable.  ``loss_fn`` / ``compute_ious`` are the
product entry points (HIP kernels, HIP tensors only); ``*_torch`` are the same formulas as plain torch ops, kept under their
own names as a second restatement the tests compare against -- nothing routes to them silently."""
import torch
import torch.nn.functional as F


def bce_loss(p, y, s, mask):
    """reference main.py:89-108.  The reference builds BCELoss(reduction=None), which raises on every torch
    release; the evident intent -- element-wise loss, masked, per-sample mean, batch mean -- is what runs here."""
    yf = y.to(p.dtype)
    mk = mask.to(p.dtype)
    if s is not None:
        loss = (F.binary_cross_entropy(p, yf, weight=s * yf, reduction="none")
                + F.binary_cross_entropy(1 - p, 1 - yf, weight=(1 - s) * (1 - yf), reduction="none"))
    else:
        loss = F.binary_cross_entropy(p, yf, reduction="none")
    loss = loss * mk
    dims = (1, 2) if mask.dim() == 3 else (1,)
    return (loss.sum(dim=dims) / mk.sum(dim=dims)).mean()


def loss_fn_torch(pm, ym, sm, moment_mask, ps, ys, ss, pe, ye, se, pa, ya, length_mask):
    """reference main.py:110-116: L_m + L_s + L_e + 0.5 L_a, as plain torch ops (any device)."""
    return (bce_loss(pm, ym, sm, moment_mask) + bce_loss(ps, ys, ss, length_mask)
            + bce_loss(pe, ye, se, length_mask) + 0.5 * bce_loss(pa, ya, None, length_mask))


NATIVE_LOSS = True          # loss through the torch-extension binding (False: the ctypes / Python autograd host, same kernels)


def _require_hip(t, what):
    if not t.is_cuda:
        from ._lib import SminHipError
        raise SminHipError(f"{what} runs on a HIP device only (got a CPU tensor); there is no CPU fallback -- "
                           f"the plain-torch restatement is available under the explicit name {what}_torch")


def loss_fn(pm, ym, sm, moment_mask, ps, ys, ss, pe, ye, se, pa, ya, length_mask):
    """reference main.py:110-116 (same argument order): one fused forward and one fused backward kernel
    (functional.LossFn, csrc/loss.hip) instead of ~40 element-wise launches.  HIP tensors only."""
    _require_hip(pm, "loss_fn")
    from . import _lib
    if NATIVE_LOSS:
        return _lib.load_torch().smin_loss(pm, ym, sm, moment_mask, ps, ys, ss, pe, ye, se, pa, ya, length_mask)
    from .functional import LossFn
    with torch.cuda.device(pm.device):
        return LossFn.apply(pm, ps, pe, pa, ym, sm, moment_mask, ys, ss, ye, se, ya, length_mask)


def compute_ious_torch(pm, ps, pe, moment_mask, sm, n=(1, 5), m=(0.1, 0.3, 0.5, 0.7)):
    """reference utils.py:10-31 as plain torch ops with a single host sync (the reference syncs once per (n, m) pair)."""
    score = pm * torch.sqrt(ps.unsqueeze(2)) * torch.sqrt(pe.unsqueeze(1)) * moment_mask
    B = score.shape[0]
    _, top = score.reshape(B, -1).topk(k=max(n), dim=1)
    ious = torch.gather(sm.reshape(B, -1), 1, top)
    counts = torch.stack([((ious[:, :n_] > m_).sum(dim=1) > 0).sum() for n_ in n for m_ in m]).tolist()
    keys = [f"R@{n_}, IoU={m_}" for n_ in n for m_ in m]
    return {k: float(v) for k, v in zip(keys, counts)}


def compute_ious(pm, ps, pe, moment_mask, sm, n=(1, 5), m=(0.1, 0.3, 0.5, 0.7)):
    """reference utils.py:10-31 on the device.  With the reference's default n / m the whole metric is one kernel
    (csrc/metrics.hip) and one host read, for any L -- the 512 x 512 long-video map included.  Other (n, m), or a map
    with fewer than five proposals (where the reference's topk raises), take the torch form on the device."""
    _require_hip(pm, "compute_ious")
    if tuple(n) == (1, 5) and tuple(m) == (0.1, 0.3, 0.5, 0.7) and pm.shape[1] * pm.shape[2] >= 5:
        from ._lib import call, ptr, stream
        B, L = ps.shape
        pm_, ps_, pe_, sm_ = (x.detach().float().contiguous() for x in (pm, ps, pe, sm))
        mm_ = (moment_mask if moment_mask.dtype in (torch.bool, torch.uint8) else moment_mask != 0).contiguous()
        counts, ws = pm_.new_empty(8), pm_.new_empty((B, 8))
        with torch.cuda.device(pm.device):
            call("smin_compute_ious", stream(), ptr(pm_), ptr(ps_), ptr(pe_), ptr(mm_), ptr(sm_), B, L, ptr(counts), ptr(ws))
        vals = counts.tolist()
        return {f"R@{n_}, IoU={m_}": vals[a * 4 + c] for a, n_ in enumerate(n) for c, m_ in enumerate(m)}
    return compute_ious_torch(pm, ps, pe, moment_mask, sm, n, m)


MODEL_INPUTS = ("video_features", "video_mask", "query_features", "query_mask", "length_mask", "moment_mask")
LOSS_TARGETS = ("ym", "sm", "ys", "ss", "ye", "se", "ya")


class CapturedStep:
    """The train step of the reference's loop (main.py:141-160: zero_grad -> forward -> loss -> backward -> optimizer.step) replayed
    as ONE HIP graph.  The small configurations are bound by the host, not the device: ~210 launches of ~15 us of work each cost
    2.8 ms (tacos.yml) / 3.4 ms (charadessta.yml) per step when every launch is issued by the interpreter; a graph replay issues
    them from the driver.

    What makes the step capturable: nothing inside it may wait for the device, and the one value the forward normally asks the
    device for is the number of valid cells of ``moment_mask`` (it sizes every per-cell tensor).  Here it is read BEFORE the step
    (one scalar read per call) and handed to the model (``SMIN.known_cell_count``); a graph is captured per (tensor shapes, cell
    count) and kept (``max_graphs``, least recently used evicted).  Batches of full-length videos -- the common case: the
    reference's dataset resamples every video longer than T to exactly T frames (dataset.py:40-74) -- share one count; a batch
    with a count not seen before is captured on first sight (three eager steps + the capture).  Limits: the optimizer must be
    capturable (``torch.optim.Adam(..., capturable=True)``); not combined with data parallel (collectives inside a captured
    step are not handled here).  Opt-in: results are bit-identical to the eager step (tests/test_hip_parity.py).

        step = CapturedStep(model, optimizer)
        loss, (pm, ps, pe, pa) = step(batch)        # batch: dict with MODEL_INPUTS + LOSS_TARGETS; outputs are the graph's
                                                    # static tensors, overwritten by the next call
    """

    def __init__(self, model, optimizer=None, max_graphs=4, warmup=3):
        if getattr(model, "grad_sync", False):
            raise ValueError("CapturedStep: not combined with the in-node gradient exchange")
        self.model, self.optimizer, self.max_graphs, self.warmup = model, optimizer, max_graphs, warmup
        # fewer stream branches in the captured step -- and in this process's eager steps: measured on tacos.yml, a graph replays in
        # 2.5 ms/step when the backward's extra tail streams were never used by the process, in 3.5 ms once they were
        model.tail_split = False
        self.entries = {}                                          # key -> dict(graph, static, loss, outputs, used)
        self.clock = 0

    @staticmethod
    def _key(batch, n):
        return (n,) + tuple((k, tuple(batch[k].shape), str(batch[k].dtype)) for k in MODEL_INPUTS + LOSS_TARGETS)

    def _run(self, b):
        from . import _lib  # noqa: F401
        if self.optimizer is not None:
            self.optimizer.zero_grad(set_to_none=True)
        else:
            for p in self.model.parameters():
                p.grad = None
        out = self.model(*[b[k] for k in MODEL_INPUTS])
        pm, ps, pe, pa = out
        loss = loss_fn(pm, b["ym"], b["sm"], b["moment_mask"], ps, b["ys"], b["ss"], pe, b["ye"], b["se"], pa, b["ya"], b["length_mask"])
        loss.backward()
        if self.optimizer is not None:
            self.optimizer.step()
        return loss, out

    def __call__(self, batch):
        dev = batch["moment_mask"].device
        _require_hip(batch["moment_mask"], "CapturedStep")
        from . import _lib
        status = _lib.load_torch().layout_status(dev)
        # the one device -> host read of the step, ahead of it (it also orders this call behind the previous replay)
        n, bad = (int(v) for v in torch.stack([(batch["moment_mask"] != 0).sum().to(torch.int64), status[0].to(torch.int64)]).tolist())
        if bad:
            status.zero_()
            raise RuntimeError("CapturedStep: the previous step ran with a cell count that did not match its moment_mask")
        key = self._key(batch, n)
        self.clock += 1
        e = self.entries.get(key)
        if e is None:
            e = self._capture(batch, n, key)
        else:
            for k in MODEL_INPUTS + LOSS_TARGETS:
                e["static"][k].copy_(batch[k], non_blocking=True)
        e["used"] = self.clock
        self.model.known_cell_count = n                            # (for an eager call right after; the graph has it baked in)
        e["graph"].replay()
        self.model.known_cell_count = None
        return e["loss"], e["outputs"]

    def _capture(self, batch, n, key):
        if len(self.entries) >= self.max_graphs:
            del self.entries[min(self.entries, key=lambda k: self.entries[k]["used"])]
        static = {k: batch[k].clone() for k in MODEL_INPUTS + LOSS_TARGETS}
        self.model.known_cell_count = n
        try:
            # torch's capture protocol: a few eager steps on a side stream first (allocator pools, lazily built tables and kernel
            # attributes, the optimizer's state), then the capture.  NOTE: these steps update the parameters like any other step.
            s = torch.cuda.Stream(device=static["moment_mask"].device)
            s.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(s):
                for _ in range(self.warmup):
                    self._run(static)
            torch.cuda.current_stream().wait_stream(s)
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                loss, out = self._run(static)
        finally:
            self.model.known_cell_count = None
        e = dict(graph=g, static=static, loss=loss, outputs=out, used=self.clock)
        self.entries[key] = e
        return e
