"""Device-side construction of the masks and training targets of one batch.

The reference builds them per sample on the host in ``AbstractDataset.__getitem__`` (dataset.py:129-187) from the
annotation ``(start, end, duration)`` and the number of sampled frames, with Python loops over torch CPU tensors, and
ships eleven tensors per sample through the DataLoader.  Here the whole batch is built by a handful of broadcast
ops on the device from three small vectors, so a multi-GPU job only has to feed features, ``nfeats`` and the
annotation times.  Formulas and their reference lines:

* ``video_mask[:nfeats] = 1``                                                     dataset.py:141-142
* ``length_mask[:ceil(nfeats / (T / L))] = 1``                                    dataset.py:144-145
* ``moment_mask = triu(outer(length_mask, length_mask))``                         dataset.py:147-148
* ``sm`` = IoU of every (start snippet i, end snippet j) window with the ground truth: ``s_i = i*dur/L``,
  ``e_j = (j+1)*dur/L``, ``inter = max(0, min(e_j, te) - max(s_i, ts))``,
  ``union = max(0, max(e_j, te) - min(s_i, ts))``, ``iou = inter / union``           dataset.py:95-110
* ``ss, se`` = unnormalised Gaussians around the true boundaries with ``sigma = (te - ts) / 5``   dataset.py:112-120
* ``ya`` = snippets fully inside the ground-truth window                           dataset.py:122-126
* binary labels ``ym, ys, ye`` = the scores thresholded at 0.5                       dataset.py:151, 154-155
"""
import math

import torch


def build_targets(times, duration, nfeats, T, L, device=None):
    """times (B, 2) = ground-truth (start, end) in seconds, duration (B,), nfeats (B,) sampled frames (<= T).

    Returns the dict of batch tensors ``main.py`` reads (``video_mask`` uint8 (B,T,1), ``length_mask`` bool (B,L),
    ``moment_mask`` bool (B,L,L), ``sm`` (B,L,L), ``ym``, ``ss``/``se`` (B,L), ``ys``/``ye``, ``ya`` bool (B,L)),
    all on ``device``.  No host loop, no sync."""
    times = torch.as_tensor(times, dtype=torch.float32, device=device)
    device = times.device
    duration = torch.as_tensor(duration, dtype=torch.float32, device=device)
    nfeats = torch.as_tensor(nfeats, device=device).to(torch.int64).clamp(max=T)
    ts, te = times[:, 0:1], times[:, 1:2]                                        # (B, 1)
    dur = duration.unsqueeze(1)
    idx = torch.arange(L, device=device, dtype=torch.float32).unsqueeze(0)        # (1, L)
    s_times = idx * dur / L                                                       # (B, L) start time of snippet i
    e_times = (idx + 1.0) * dur / L                                               # (B, L) end time of snippet j

    video_mask = (torch.arange(T, device=device).unsqueeze(0) < nfeats.unsqueeze(1)).to(torch.uint8).unsqueeze(-1)
    # ceil(nfeats / (T / L)) with the reference's float division (T / L is exact for L | T)
    n_len = torch.ceil(nfeats.to(torch.float64) / (T / L)).to(torch.int64)
    length_mask = torch.arange(L, device=device).unsqueeze(0) < n_len.unsqueeze(1)
    moment_mask = torch.triu(length_mask.unsqueeze(2) & length_mask.unsqueeze(1))

    si, ej = s_times.unsqueeze(2), e_times.unsqueeze(1)                           # (B, L, 1), (B, 1, L)
    gs, ge = ts.unsqueeze(2), te.unsqueeze(2)
    inter = (torch.minimum(ej, ge) - torch.maximum(si, gs)).clamp(min=0.0)
    union = (torch.maximum(ej, ge) - torch.minimum(si, gs)).clamp(min=0.0)
    sm = inter / union                                                            # 0/0 -> nan exactly where the reference has it

    sigma = (te - ts) / 5.0
    ss = torch.exp(-(s_times - ts) ** 2 / (2.0 * sigma ** 2))
    se = torch.exp(-(e_times - te) ** 2 / (2.0 * sigma ** 2))
    ya = (s_times >= ts) & (e_times <= te)
    return dict(video_mask=video_mask, length_mask=length_mask, moment_mask=moment_mask, sm=sm, ym=sm > 0.5,
                ss=ss, ys=ss > 0.5, se=se, ye=se > 0.5, ya=ya)


def expected_length(nfeats, T, L):
    """ceil(nfeats / (T / L)) -- the number of valid snippets of a video with nfeats sampled frames (dataset.py:145)."""
    return math.ceil(min(nfeats, T) / (T / L))
