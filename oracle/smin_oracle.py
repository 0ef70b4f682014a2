"""CPU oracle for the SMIN hot path -- TEST INFRASTRUCTURE ONLY.

This file is a from-scratch, functional (no nn.Module) restatement of the
algorithm in the reference's ``models.py`` (ChanukyaVardhan/Video-Moment-
Localization).  It is written from SURVEY.md section 8a and from reading the
reference source as text; every function cites the reference ``file:line`` it
follows.  It is the *checker* for the HIP path: only ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may
import it.  The product package (``video-moment-localization_amd``) never does.

Parity pinning: the reference ships no golden vectors or numeric tests for this
path, so the oracle is pinned by fixtures generated in the
build container by importing ``/root/reference/models.py`` on CPU
(``tests/golden/make_golden.py``; fixtures committed under ``tests/golden``).
``tests/test_oracle_golden.py`` asserts this file reproduces every fixture.

All functions are dtype-generic (fp32 like the reference, or fp64 to measure
the fp32 noise floor) and take the reference's ``state_dict`` key names.
"""
import math

import torch

# --------------------------------------------------------------------------
# 2D temporal proposal map
# --------------------------------------------------------------------------

def content_windows(T, L, C):
    """Clip windows of the content matrix, as index arithmetic.

    Restates models.py:88-98 (compute_content_matrix).  For moment (i, j),
    i <= j, the window covers ``n = (j-i+1)*(T//L)`` frames starting at
    ``i*(T//L)``; it is cut into ``min(C, n)`` clips of ``cs = max(1, n//C)``
    frames (trailing ``n - C*cs`` frames are dropped, clips ``c >= n`` are
    empty).  Returns int64 tensors ``start, size`` of shape (L, L, C) with
    ``size == 0`` for empty clips / lower-triangular cells.
    """
    r = T // L
    i = torch.arange(L).view(L, 1, 1)
    j = torch.arange(L).view(1, L, 1)
    c = torch.arange(C).view(1, 1, C)
    n = (j - i + 1) * r
    cs = torch.clamp(torch.div(n, C, rounding_mode="floor"), min=1)
    valid = (j >= i) & (c < torch.minimum(torch.full_like(n, C), n))
    start = i * r + c * cs
    size = torch.where(valid, cs.expand(L, L, C), torch.zeros(1, dtype=torch.long))
    start = torch.where(valid, start, torch.zeros(1, dtype=torch.long))
    return start.contiguous(), size.contiguous()


def content_matrix(T, L, C, dtype=torch.float32):
    """Dense (L, L, C, T) averaging matrix (models.py:88-98), vectorised."""
    start, size = content_windows(T, L, C)
    t = torch.arange(T).view(1, 1, 1, T)
    s = start.unsqueeze(-1)
    z = size.unsqueeze(-1)
    inside = (t >= s) & (t < s + z)
    w = torch.where(z > 0, 1.0 / z.clamp(min=1).to(torch.float32), torch.zeros(1))
    # the reference stores 1/clip_size as a python float rounded into a float32 tensor
    return (inside.to(torch.float32) * w).to(dtype)


def proposal_generation(f, moment_mask, T, L, C):
    """models.py:115-126: fc (B,L,L,C,D), fm (B,L,L,D), fb (B,L,D)."""
    Wc = content_matrix(T, L, C, f.dtype)
    fc = torch.einsum("lmit,btj->blmij", Wc, f) * moment_mask[:, :, :, None, None]
    fm = fc.mean(dim=3)
    r = T // L
    B, _, D = f.shape
    nl = f.shape[1] // r
    fb = f[:, : nl * r].reshape(B, nl, r, D).mean(dim=2)   # AvgPool1d(r, r) over time
    return fc, fm, fb


# --------------------------------------------------------------------------
# Backbone (feeds the path; stays plain torch in the product too)
# --------------------------------------------------------------------------

def video_encoder(sd, video_features, video_mask, prefix="backbone.videoencoder."):
    """models.py:25-36."""
    dt = video_features.dtype
    vm = video_mask.to(dt)
    x = torch.nn.functional.linear(video_features, sd[prefix + "ve.weight"].to(dt), sd[prefix + "ve.bias"].to(dt)) * vm
    Tn = video_mask.shape[1]
    p = sd[prefix + "pe.weight"].to(dt)[:Tn].unsqueeze(0) * vm
    return x + p


def _lstm_dir(x, lengths, w_ih, w_hh, b_ih, b_hh, reverse):
    """One direction of one LSTM layer over variable-length sequences.

    Equivalent to nn.LSTM on a packed sequence (models.py:52-54): positions at
    or beyond ``lengths[b]`` are never processed and their outputs are zero.
    Gate order i, f, g, o (torch convention).
    """
    B, N, _ = x.shape
    H = w_hh.shape[1]
    h = x.new_zeros(B, H)
    c = x.new_zeros(B, H)
    out = x.new_zeros(B, N, H)
    steps = range(N - 1, -1, -1) if reverse else range(N)
    for t in steps:
        live = (lengths > t).to(x.dtype).unsqueeze(1)
        g = x[:, t] @ w_ih.t() + b_ih + h @ w_hh.t() + b_hh
        gi, gf, gg, go = g.chunk(4, dim=1)
        c_new = torch.sigmoid(gf) * c + torch.sigmoid(gi) * torch.tanh(gg)
        h_new = torch.sigmoid(go) * torch.tanh(c_new)
        c = live * c_new + (1 - live) * c
        h = live * h_new + (1 - live) * h
        out[:, t] = live * h_new
    return out


def query_encoder(sd, query_features, query_mask, prefix="backbone.queryencoder.lstm."):
    """models.py:48-64: 2-layer BiLSTM; fs = [fwd h at last word, bwd h at word 0]."""
    dt = query_features.dtype
    lengths = query_mask.reshape(query_mask.shape[0], -1).sum(1).long()
    x = query_features
    for layer in range(2):
        outs = []
        for suffix, rev in (("", False), ("_reverse", True)):
            outs.append(_lstm_dir(
                x, lengths,
                sd[f"{prefix}weight_ih_l{layer}{suffix}"].to(dt), sd[f"{prefix}weight_hh_l{layer}{suffix}"].to(dt),
                sd[f"{prefix}bias_ih_l{layer}{suffix}"].to(dt), sd[f"{prefix}bias_hh_l{layer}{suffix}"].to(dt), rev))
        x = torch.cat(outs, dim=2)
    fw = x
    B, N, H2 = fw.shape
    H = H2 // 2
    last = (lengths - 1).view(B, 1, 1).expand(B, 1, H)
    fs = torch.cat([fw[:, :, :H].gather(1, last).view(B, H), fw[:, 0, H:]], dim=1)
    return fs, fw


def backbone(sd, video_features, video_mask, query_features, query_mask):
    """models.py:74-83."""
    fv = video_encoder(sd, video_features, video_mask)
    fs, fw = query_encoder(sd, query_features, query_mask)
    return fv * fs.unsqueeze(1), fs, fw


# --------------------------------------------------------------------------
# Cross-modal attention blocks
# --------------------------------------------------------------------------

def _lin(sd, name, x):
    return torch.nn.functional.linear(x, sd[name + ".weight"].to(x.dtype), sd[name + ".bias"].to(x.dtype))


def _word_attention(sd, prefix, query, key, value, qmask_row, scale_dim):
    """models.py:137-154 and 207-226: softmax(mask(Wq q . (Wk k)^T / sqrt(D))) . value.

    ``query`` is (B, R, D) with R any number of rows per sample; ``qmask_row``
    is (B, 1, Nq) float.  The value is the *un-projected* key input.
    """
    q = _lin(sd, prefix + "W_q", query)
    k = _lin(sd, prefix + "W_k", key)
    s = q @ k.transpose(1, 2) / math.sqrt(scale_dim)
    s = s * qmask_row
    s = s.masked_fill(qmask_row == 0, -1e9)
    return torch.softmax(s, dim=-1) @ value


def content_unit(sd, p, f_c, f_w, f_s, f_m, query_mask, moment_mask):
    """models.py:242-276 (ContentUnit) with models.py:207-226 (ContentAttention)."""
    dt = f_c.dtype
    B, L, _, C, D = f_c.shape
    m = moment_mask.to(dt)[:, :, :, None, None]
    qm = query_mask.to(dt)
    dl = sd[p + "linear_c_hat.weight"].shape[0]
    c_hat = _lin(sd, p + "linear_c_hat", f_c) * m
    w_hat = _lin(sd, p + "linear_w_hat", f_w) * qm
    s_hat = _lin(sd, p + "linear_s_hat", f_s)
    qrow = qm.reshape(B, 1, -1)
    a = _word_attention(sd, p + "attn_layer.", c_hat.reshape(B, L * L * C, dl), w_hat, w_hat, qrow, dl)
    a = a.reshape(B, L, L, C, dl) * m
    q = c_hat * (a + s_hat[:, None, None, None, :])
    A = torch.softmax(q @ q.transpose(3, 4) / math.sqrt(dl), dim=-1) * m
    cc = _lin(sd, p + "linear_c", A @ c_hat) * m
    gated = torch.sigmoid(f_m * f_s[:, None, None, :]) * f_m
    return cc + f_c + gated.unsqueeze(3)


def boundary_unit(sd, p, f_b, f_w, f_s, f_m, query_mask, length_mask):
    """models.py:164-196 (BoundaryUnit) with models.py:137-154 (Attention)."""
    dt = f_b.dtype
    B, L, D = f_b.shape
    lm = length_mask.to(dt)
    lcol = lm.unsqueeze(-1)
    qrow = query_mask.to(dt).reshape(B, 1, -1)
    baq = _word_attention(sd, p + "attn_layer.", f_b, f_w, f_w, qrow, D) * lcol
    bq = f_b * (baq + f_s.unsqueeze(1))
    z = bq @ bq.transpose(1, 2) / math.sqrt(D)
    lrow = lm.unsqueeze(1)
    z = (z * lrow).masked_fill(lrow == 0, -1e9)
    A = torch.softmax(z, dim=-1) * lcol
    bb = (A @ f_b) * lcol
    h = torch.sigmoid(f_m * f_s[:, None, None, :]) * f_m
    bm = (A.unsqueeze(3) * h).sum(dim=2)
    return bb + f_b + bm


def moment_unit(sd, p, f_c, f_m, f_b, moment_mask):
    """models.py:288-303: two 1x1 convs == per-cell D x D linear maps."""
    dt = f_m.dtype
    D = f_m.shape[-1]
    m = moment_mask.to(dt).unsqueeze(-1)
    wfb = sd[p + "conv_layer_fb.weight"].to(dt).reshape(D, D)
    wfc = sd[p + "conv_layer_fc.weight"].to(dt).reshape(D, D)
    pair = f_b.unsqueeze(2) * f_b.unsqueeze(1)
    y_fb = (pair @ wfb.t() + sd[p + "conv_layer_fb.bias"].to(dt)) * m
    y_fc = (f_c.mean(dim=3) @ wfc.t() + sd[p + "conv_layer_fc.bias"].to(dt)) * m
    return y_fb + y_fc + f_m


def smi_layer(sd, k, f_c, f_m, f_b, f_w, f_s, query_mask, length_mask, moment_mask):
    """models.py:316-322: content and boundary from layer inputs, moment from their outputs."""
    p = f"smis.{k}."
    cu = content_unit(sd, p + "content_unit.", f_c, f_w, f_s, f_m, query_mask, moment_mask)
    bu = boundary_unit(sd, p + "boundary_unit.", f_b, f_w, f_s, f_m, query_mask, length_mask)
    mu = moment_unit(sd, p + "moment_unit.", cu, f_m, bu, moment_mask)
    return cu, mu, bu


def localization(sd, f_m, f_b, length_mask, moment_mask, p="localization."):
    """models.py:335-344: sigmoid score heads, masked."""
    dt = f_m.dtype
    D = f_m.shape[-1]
    def head(name, x):
        return torch.sigmoid(x @ sd[p + name + ".weight"].to(dt).reshape(D) + sd[p + name + ".bias"].to(dt))
    pm = head("conv_layer_pm", f_m) * moment_mask.to(dt)
    lm = length_mask.to(dt)
    return pm, head("conv_layer_ps", f_b) * lm, head("conv_layer_pe", f_b) * lm, head("conv_layer_pa", f_b) * lm


def num_layers(sd):
    return 1 + max(int(k.split(".")[1]) for k in sd if k.startswith("smis."))


def smin_forward(sd, cfg, video_features, video_mask, query_features, query_mask, length_mask, moment_mask,
                 return_seams=False):
    """models.py:367-377.  ``cfg`` needs T, L, C.  Returns (pm, ps, pe, pa) [, seams dict]."""
    seams = {}
    f, fs, fw = backbone(sd, video_features, video_mask, query_features, query_mask)
    fc, fm, fb = proposal_generation(f, moment_mask, cfg["T"], cfg["L"], cfg["C"])
    if return_seams:
        seams.update(f=f, fs=fs, fw=fw, fc=fc, fm=fm, fb=fb)
    for k in range(num_layers(sd)):
        fc, fm, fb = smi_layer(sd, k, fc, fm, fb, fw, fs, query_mask, length_mask, moment_mask)
        if return_seams:
            seams[f"cu{k}"], seams[f"mu{k}"], seams[f"bu{k}"] = fc, fm, fb
    out = localization(sd, fm, fb, length_mask, moment_mask)
    return (out, seams) if return_seams else out


# --------------------------------------------------------------------------
# Restated training loss (caller side; main.py:89-116 with reduction='none')
# --------------------------------------------------------------------------

def scaled_bce(p, y, s, mask):
    """main.py:89-108.  The reference passes reduction=None (a ValueError on every
    torch release); this is the evident intent, reduction='none'.  BCELoss itself
    (log terms clamped at -100, finite backward at p in {0, 1}) is torch's."""
    bce = torch.nn.functional.binary_cross_entropy
    yf = y.to(p.dtype)
    mk = mask.to(p.dtype)
    if s is not None:
        loss = (bce(p, yf, weight=s * yf, reduction="none")
                + bce(1 - p, 1 - yf, weight=(1 - s) * (1 - yf), reduction="none"))
    else:
        loss = bce(p, yf, reduction="none")
    loss = loss * mk
    dims = (1, 2) if mask.dim() == 3 else (1,)
    return (loss.sum(dim=dims) / mk.sum(dim=dims)).mean()


def loss_fn(pm, ym, sm, moment_mask, ps, ys, ss, pe, ye, se, pa, ya, length_mask):
    """main.py:110-116."""
    return (scaled_bce(pm, ym, sm, moment_mask) + scaled_bce(ps, ys, ss, length_mask)
            + scaled_bce(pe, ye, se, length_mask) + 0.5 * scaled_bce(pa, ya, None, length_mask))


# --------------------------------------------------------------------------
# Metric (caller side; utils.py:10-31)
# --------------------------------------------------------------------------

def compute_ious(pm, ps, pe, moment_mask, sm, n=(1, 5), m=(0.1, 0.3, 0.5, 0.7)):
    score = pm * torch.sqrt(ps.unsqueeze(2)) * torch.sqrt(pe.unsqueeze(1)) * moment_mask
    B = score.shape[0]
    _, top = score.reshape(B, -1).topk(k=max(n), dim=1)
    ious = torch.gather(sm.reshape(B, -1), 1, top)
    out = {}
    for n_ in n:
        for m_ in m:
            out[f"R@{n_}, IoU={m_}"] = float(((ious[:, :n_] > m_).sum(dim=1) > 0).sum().item())
    return out


# --------------------------------------------------------------------------
# Synthetic inputs (mimic dataset.py:142-176 mask construction)
# --------------------------------------------------------------------------

def synthetic_batch(B, T, L, Nq, Din, seed=0, with_labels=True, full_length=False):
    g = torch.Generator().manual_seed(seed)
    vf = torch.randn(B, T, Din, generator=g)
    qf = torch.randn(B, Nq, 300, generator=g)
    vmask = torch.zeros(B, T, 1, dtype=torch.uint8)
    qmask = torch.zeros(B, Nq, 1, dtype=torch.uint8)
    lmask = torch.zeros(B, L, dtype=torch.bool)
    for b in range(B):
        if b % 2 == 0 or full_length:
            nf, nq = T, Nq
        else:
            nf = int(torch.randint(max(1, T // 2), T, (1,), generator=g))
            nq = int(torch.randint(min(3, Nq - 1), Nq, (1,), generator=g)) if Nq > 1 else 1
        vf[b, nf:] = 0
        qf[b, nq:] = 0
        vmask[b, :nf] = 1
        qmask[b, :nq] = 1
        lmask[b, : math.ceil(nf / (T / L))] = True
    mmask = torch.triu(lmask.unsqueeze(2) & lmask.unsqueeze(1))
    batch = dict(video_features=vf, video_mask=vmask, query_features=qf, query_mask=qmask,
                 length_mask=lmask, moment_mask=mmask)
    if with_labels:
        sm = torch.rand(B, L, L, generator=g) * mmask
        ss = torch.rand(B, L, generator=g)
        se = torch.rand(B, L, generator=g)
        batch.update(sm=sm, ym=sm > 0.5, ss=ss, ys=ss > 0.5, se=se, ye=se > 0.5,
                     ya=torch.rand(B, L, generator=g) > 0.5)
    return batch


def _hash_uniform(n, salt):
    """Platform-independent pseudo-random U(-1, 1): splitmix64 of (index, salt) in
    wrapping int64 arithmetic (no dependence on any library RNG stream)."""
    def lsr(x, s):
        return (x >> s) & ((1 << (64 - s)) - 1)
    def i64(v):   # python int -> wrapped int64
        v &= (1 << 64) - 1
        return v - (1 << 64) if v >= (1 << 63) else v
    x = torch.arange(n, dtype=torch.int64) + i64(salt * 0x9E3779B97F4A7C15)
    x = (x ^ lsr(x, 30)) * i64(0xBF58476D1CE4E5B9)
    x = (x ^ lsr(x, 27)) * i64(0x94D049BB133111EB)
    x = x ^ lsr(x, 31)
    return lsr(x, 11).to(torch.float64) * (2.0 / (1 << 53)) - 1.0


def formula_state_dict(shapes, gain=1.0):
    """Closed-form, RNG-free parameter fill used by the full-size golden cases.

    Every element is a splitmix64 hash of (flat index, name hash) mapped to
    U(-a, a) with ``a`` set from the fan-in, so activations are O(1) and the
    scores spread well beyond the 0.46-0.52 band that default initialisation
    gives.  Only shapes + names are needed to regenerate it."""
    sd = {}
    for name, shape in shapes.items():
        n = 1
        for s in shape:
            n *= s
        salt = sum(ord(ch) * (i + 1) for i, ch in enumerate(name)) + 7919 * len(name)
        u = _hash_uniform(n, salt)
        if "lstm" in name:
            amp = 1.0 / math.sqrt(shape[-1] if "weight_hh" in name else max(shape[0] // 4, 1))
        elif "pe.weight" in name:
            amp = 1.0
        elif len(shape) > 1:
            amp = gain * math.sqrt(3.0 / max(n // shape[0], 1))
        else:
            amp = 0.1 * gain
        sd[name] = (amp * u).to(torch.float32).reshape(shape)
    return sd
