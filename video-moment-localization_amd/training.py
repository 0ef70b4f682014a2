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
