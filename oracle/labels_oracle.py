"""TEST INFRASTRUCTURE ONLY -- per-sample restatement of the reference's target construction
(`/root/reference/dataset.py:95-126, 141-155`), one function per reference method, same torch CPU ops in the same
order.  PARITY UNPINNED: `dataset.py` cannot be imported in the build container (it needs torchtext and h5py, which
are not installed, and downloads GloVe at class-definition time, dataset.py:19), and the reference holds no fixtures
for these functions; the restatement below follows the source text line by line instead."""
import math

import numpy as np
import torch


def get_iou(gt_spos, gt_epos, duration, L):                                   # dataset.py:95-110
    s_times = torch.arange(0, L).float() * duration / L
    e_times = torch.arange(1, L + 1).float() * duration / L
    preds = torch.stack([s_times.repeat_interleave(L), e_times.repeat(L)], dim=1)
    gts = torch.tensor([gt_spos, gt_epos]).unsqueeze(0)
    inter = torch.max(torch.tensor(0.0), torch.min(preds[:, 1], gts[:, 1]) - torch.max(preds[:, 0], gts[:, 0]))
    union = torch.max(torch.tensor(0.0), torch.max(preds[:, 1], gts[:, 1]) - torch.min(preds[:, 0], gts[:, 0]))
    return (inter / union).reshape(L, L)


def get_boundary_penalties(tau_s, tau_e, duration, L):                         # dataset.py:112-120
    s_times = torch.arange(0, L).float() * duration / L
    e_times = torch.arange(1, L + 1).float() * duration / L
    sigma = (tau_e - tau_s) / 5.0
    return torch.exp(-(s_times - tau_s) ** 2 / (2.0 * sigma ** 2)), torch.exp(-(e_times - tau_e) ** 2 / (2.0 * sigma ** 2))


def get_snippet_label(tau_s, tau_e, duration, L):                              # dataset.py:122-126
    s_times = torch.arange(0, L).float() * duration / L
    e_times = torch.arange(1, L + 1).float() * duration / L
    return torch.logical_and(s_times >= tau_s, e_times <= tau_e)


def sample_targets(spos, epos, duration, nfeats, T, L):                        # dataset.py:141-155
    nfeats = min(nfeats, T)
    video_mask = np.zeros((T, 1)); video_mask[:nfeats] = 1
    length_mask = np.zeros(L); length_mask[:math.ceil(nfeats / (T / L))] = 1
    moment_mask = np.zeros((L, L))
    moment_mask[np.triu_indices(L)] = np.logical_and.outer(length_mask, length_mask)[np.triu_indices(L)]
    ious = get_iou(spos, epos, duration, L)
    s_s, s_e = get_boundary_penalties(spos, epos, duration, L)
    return dict(video_mask=torch.ByteTensor(video_mask), length_mask=torch.BoolTensor(length_mask), moment_mask=torch.BoolTensor(moment_mask),
                sm=ious, ym=ious > 0.5, ss=s_s, ys=s_s > 0.5, se=s_e, ye=s_e > 0.5, ya=get_snippet_label(spos, epos, duration, L))
