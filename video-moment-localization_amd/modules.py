"""nn.Module surface of the reference's models.py, backed by the HIP kernels.

Class names, constructor argument order/defaults, forward signatures, sub-module attribute names and
therefore state_dict keys follow /root/reference/models.py (cited per class) so that an unmodified
main.py / simpletest.py / config/*.yml can import this module in place of the reference's.  Parameters
live in the same torch holders (Linear / Embedding / LSTM / Conv1d / Conv2d) created in the same order,
so ``torch.manual_seed(s); SMIN(...)`` reproduces the reference's default initialisation.

Inside ``SMIN.forward`` the L x L map lives in the packed valid-cell layout (cells.py); the dense
(B, L, L, ...) tensors of the reference appear only at the stand-alone sub-module seams
(``ContentUnit.forward`` etc.), which convert at the boundary and accept arbitrary dense inputs.
Per-cell work and the whole boundary unit run in HIP (functional.py); the O(B*Nq*dl) word-side projections of the
content unit and the backbone stay plain torch library calls.
"""
import functools
import math

import os

import torch
import torch.nn as nn

from ._lib import SminHipError

from .cells import CellLayout
from .functional import (VideoFuseFn, WordPrepFn, BiLstmLayerFn, BoundaryUnitFn, ClipWindowMeansFn, ContentAttnFn, ContentUnitFn, GateFn, LinearRowsFn, MomentUnitFn,
                         ProposalMapFn, ProposalMeansFn, ScoreMapFn)


def _hip_forward(fn):
    """Every forward of this module surface runs on a HIP device only (there is no CPU path: a CPU tensor raises), with the
    tensor's device made current for the call -- the C ABI launches on torch's current stream, so a model living on
    cuda:N while another device is current (the reference's ctor accepts any ``device``) must not launch on the wrong
    device's stream.  autograd restores the forward's device for the backward nodes by itself."""
    @functools.wraps(fn)
    def wrapper(self, x, *args, **kwargs):
        if not (isinstance(x, torch.Tensor) and x.is_cuda):
            raise SminHipError(f"{type(self).__name__}.forward runs on a HIP device only (got a CPU tensor); there is no CPU fallback")
        if x.device.index != torch.cuda.current_device():
            with torch.cuda.device(x.device):
                return fn(self, x, *args, **kwargs)
        return fn(self, x, *args, **kwargs)
    return wrapper


def _rows(mask):
    """(B, N, 1) / (B, N) 0-1 mask -> float (B, N); unlike the reference's .squeeze() this is B=1 safe."""
    return mask.reshape(mask.shape[0], -1).float()


class VideoEncoder(nn.Module):
    """reference models.py:7-36."""

    def __init__(self, T=64, d=512, input_video_dim=1024, device='cpu'):
        super().__init__()
        self.T, self.d, self.d0, self.device = T, d, input_video_dim, device
        self.ve = nn.Linear(self.d0, self.d)
        self.pe = nn.Embedding(self.T, self.d)

    @_hip_forward
    def forward(self, video_features, video_mask):
        vm = video_mask.float()
        pos = torch.arange(video_mask.shape[1], device=video_features.device)
        if self.d0 % 4 == 0 and self.d % 4 == 0 and video_features.dtype == torch.float32:
            B, T, _ = video_features.shape                          # the projection on the library's MFMA engine
            y = LinearRowsFn.apply(self.ve.weight, self.ve.bias, None, None, 1, video_features.reshape(B * T, self.d0)).view(B, T, self.d)
        else:
            y = self.ve(video_features)
        return y * vm + self.pe(pos).unsqueeze(0) * vm


_SIDE_STREAMS = {}
# parameters are deliberately used on two streams (see SMIN._forward_stream); autograd then synchronises the streams where a
# gradient is accumulated, which is what we want -- not worth a warning per backward pass
if hasattr(torch.autograd.graph, "set_warn_on_accumulate_grad_stream_mismatch"):
    torch.autograd.graph.set_warn_on_accumulate_grad_stream_mismatch(False)


def _side_stream(device):
    """One auxiliary HIP stream per device for work that is independent of the main stream's chain."""
    key = (device.type, device.index)
    st = _SIDE_STREAMS.get(key)
    if st is None:
        st = _SIDE_STREAMS[key] = torch.cuda.Stream(device)
    return st


class QueryEncoder(nn.Module):
    """reference models.py:38-64 (2-layer BiLSTM over packed word sequences)."""

    fused_lstm = True              # HIP BiLSTM layer kernels on the GPU (False: torch / MIOpen, same results)

    def __init__(self, max_query_length=13, lstm_hidden_size=256):
        super().__init__()
        self.max_query_length, self.lstm_hidden_size = max_query_length, lstm_hidden_size
        self.lstm = nn.LSTM(input_size=300, hidden_size=lstm_hidden_size, num_layers=2, bidirectional=True, batch_first=True)

    @_hip_forward
    def forward(self, query_features, query_mask):
        """Same result as the reference's pack_padded_sequence / pad_packed_sequence round trip, but on padded
        tensors with the lengths kept on the device: the reference copies them to the host here
        (models.py:52), a sync that stalls the whole pipeline once per step.  Forward direction: positions past
        a sample's length never influence earlier outputs and are zeroed afterwards.  Backward direction: each
        sequence is reversed in place (gather), run forwards, and reversed back."""
        B, Nq, _ = query_features.shape
        H = self.lstm_hidden_size
        length = query_mask.reshape(B, -1).sum(1).long()
        if self.fused_lstm and H <= 256 and H % 4 == 0:
            # one HIP launch per layer runs the whole recurrence, both directions, lengths honoured in-kernel
            # (bilstm.hip): the library path below is ~600 launches of a few microseconds each per train step
            x, len32 = query_features, length.to(torch.int32)
            for layer in range(2):
                w = [getattr(self.lstm, f"{n}_l{layer}{sfx}") for sfx in ("", "_reverse") for n in ("weight_ih", "weight_hh", "bias_ih", "bias_hh")]
                x = BiLstmLayerFn.apply(x, len32, *w)
            return self._heads(x, length, B, Nq, H)
        t = torch.arange(Nq, device=query_features.device).unsqueeze(0)
        valid = (t < length.unsqueeze(1)).to(query_features.dtype).unsqueeze(-1)          # (B, Nq, 1)
        rev = (length.unsqueeze(1) - 1 - t).clamp(min=0).unsqueeze(-1)                     # (B, Nq, 1)
        h0 = query_features.new_zeros(1, B, H)
        x = query_features

        def run(layer, sfx, x):
            w = [getattr(self.lstm, f"{n}_l{layer}{sfx}") for n in ("weight_ih", "weight_hh", "bias_ih", "bias_hh")]
            xin = x if not sfx else torch.gather(x, 1, rev.expand(-1, -1, x.shape[-1])) * valid
            y = torch._VF.lstm(xin, (h0, h0), w, True, 1, 0.0, self.training, False, True)[0]
            if sfx:
                y = torch.gather(y, 1, rev.expand(-1, -1, H))
            return y * valid

        # The two directions of a layer are independent chains of Nq tiny, latency-bound kernels: on the GPU they run
        # on two HIP streams side by side (autograd replays each chain's backward on the stream of its forward).
        side = _side_stream(x.device)
        for layer in range(2):
            if side is None:
                outs = [run(layer, "", x), run(layer, "_reverse", x)]
            else:
                cur = torch.cuda.current_stream(x.device)
                side.wait_stream(cur)
                with torch.cuda.stream(side):
                    y_rev = run(layer, "_reverse", x)
                y_fwd = run(layer, "", x)
                cur.wait_stream(side)
                for tns in (x, rev, valid, h0):
                    tns.record_stream(side)
                y_rev.record_stream(cur)
                outs = [y_fwd, y_rev]
            x = torch.cat(outs, dim=2)
        return self._heads(x, length, B, Nq, H)

    def _heads(self, x, length, B, Nq, H):
        """(f_s, f_w) of models.py:56-63 from the padded layer output."""
        fw = x
        if Nq < self.max_query_length:
            fw = torch.nn.functional.pad(fw, (0, 0, 0, self.max_query_length - Nq))
        fw = fw.contiguous()
        last = (length - 1).clamp(min=0).view(B, 1, 1).expand(B, 1, H)
        fs = torch.cat([fw[:, :, :H].gather(1, last).view(B, H), fw[:, 0, H:]], dim=1)
        return fs, fw


class Backbone(nn.Module):
    """reference models.py:66-83."""

    def __init__(self, T=64, d=512, input_video_dim=1024, max_query_length=13, lstm_hidden_size=256, device='cpu'):
        super().__init__()
        self.videoencoder = VideoEncoder(T, d, input_video_dim, device)
        self.queryencoder = QueryEncoder(max_query_length, lstm_hidden_size)

    @_hip_forward
    def forward(self, video_features, video_mask, query_features, query_mask):
        fs, fw = self.queryencoder(query_features, query_mask)
        ve = self.videoencoder
        if (video_features.dtype == torch.float32 and ve.d0 % 4 == 0 and ve.d % 4 == 0
                and video_features.shape[1] <= ve.pe.weight.shape[0]):
            # projection + position embedding + mask + Hadamard product with f_s in one contraction (video_encoder.hip)
            B, T, _ = video_features.shape
            f = VideoFuseFn.apply(video_features, ve.ve.weight, ve.ve.bias, ve.pe.weight, video_mask.reshape(B * T).float(), fs)
            return f, fs, fw
        fv = ve(video_features, video_mask)
        return fv * fs.unsqueeze(1), fs, fw


def compute_content_matrix(T, L, C):
    """reference models.py:88-98, by index arithmetic instead of the triple Python loop.
    Kept for API completeness; the kernels never materialise it."""
    r = T // L
    i = torch.arange(L).view(L, 1, 1, 1)
    j = torch.arange(L).view(1, L, 1, 1)
    c = torch.arange(C).view(1, 1, C, 1)
    t = torch.arange(T).view(1, 1, 1, T)
    n = (j - i + 1) * r
    cs = torch.clamp(n // C, min=1)
    start = i * r + c * cs
    inside = (j >= i) & (c < torch.clamp(n, max=C)) & (t >= start) & (t < start + cs)
    return inside.float() / cs.float()


class ProposalGeneration(nn.Module):
    """reference models.py:100-126.  ``self.Wc`` is not built (4 GiB at L=512); use compute_content_matrix."""

    def __init__(self, T=64, L=16, C=4, device='cpu'):
        super().__init__()
        if T % L != 0:
            raise ValueError("ProposalGeneration needs L | T (the reference breaks otherwise, SURVEY 8a-2)")
        self.T, self.L, self.C, self.device = T, L, C, device

    def forward_packed(self, f, layout):
        return ProposalMapFn.apply(f, layout, self.T, self.L, self.C)

    @_hip_forward
    def forward(self, f, moment_mask):
        layout = CellLayout.from_mask(moment_mask)
        fc, fm, fb = self.forward_packed(f, layout)
        return layout.unpack(fc), layout.unpack(fm), fb


def _word_attention(W_q, W_k, query, key, value, mask, scale_dim):
    q, k = W_q(query), W_k(key)
    s = torch.matmul(q, k.transpose(-1, -2)) / math.sqrt(scale_dim)
    if mask is not None:
        m = _rows(mask)
        m = m.view(m.shape[0], *([1] * (s.dim() - 2)), m.shape[1])
        s = (s * m).masked_fill(m == 0, -1e9)
    p = torch.softmax(s, dim=-1)
    return torch.matmul(p, value), p


class Attention(nn.Module):
    """reference models.py:128-154 (boundary <-> word attention; O(B L Nq D), library GEMMs)."""

    def __init__(self, D):
        super().__init__()
        self.D, self.attn_weights = D, None
        self.W_q = nn.Linear(D, D)
        self.W_k = nn.Linear(D, D)

    @_hip_forward
    def forward(self, query, key, value, mask=None):
        out, _ = _word_attention(self.W_q, self.W_k, query, key, value, mask, self.D)
        return out


class BoundaryUnit(nn.Module):
    """reference models.py:156-196."""

    def __init__(self, D):
        super().__init__()
        self.D = D
        self.attn_layer = Attention(D)

    def forward_packed(self, f_b, f_w, f_s, hbar, query_mask, length_mask, layout):
        at = self.attn_layer
        return BoundaryUnitFn.apply(f_b, f_w, f_s, hbar, at.W_q.weight, at.W_q.bias, at.W_k.weight, at.W_k.bias,
                                    _rows(query_mask), length_mask.float(), layout)

    @_hip_forward
    def forward(self, f_b, f_w, f_s, f_m, query_mask, length_mask):
        B, L = f_m.shape[:2]
        layout = CellLayout.all_cells(torch.ones(B, L, L, dtype=torch.bool, device=f_m.device))
        hbar = GateFn.apply(layout.pack(f_m), f_s, layout)[0]
        return self.forward_packed(f_b, f_w, f_s, hbar, query_mask, length_mask, layout)


class ContentAttention(nn.Module):
    """reference models.py:198-226.  Inside ContentUnit this attention is fused into the HIP content
    kernels (W_q/W_k folded per sample); this stand-alone forward serves the 5-D module seam only."""

    def __init__(self, D):
        super().__init__()
        self.D, self.attn_weights = D, None
        self.W_q = nn.Linear(D, D)
        self.W_k = nn.Linear(D, D)

    @_hip_forward
    def forward(self, query, key, value, mask=None):
        B = query.shape[0]
        q = query.reshape(B, -1, query.shape[-1])
        out, _ = _word_attention(self.W_q, self.W_k, q, key, value, mask, self.D)
        return out.reshape(query.shape[:-1] + (value.shape[-1],))


class ContentUnit(nn.Module):
    """reference models.py:228-276."""

    def __init__(self, D, dl):
        super().__init__()
        self.D, self.dl = D, dl
        self.linear_c_hat = nn.Linear(D, dl)
        self.linear_w_hat = nn.Linear(D, dl)
        self.linear_s_hat = nn.Linear(D, dl)
        self.linear_c = nn.Linear(dl, D)
        self.attn_layer = ContentAttention(dl)

    def word_params(self):
        """The eight parameters of the word side, in the order csrc/word_prep.hip reads them."""
        at = self.attn_layer
        return [self.linear_w_hat.weight, self.linear_w_hat.bias, self.linear_s_hat.weight, self.linear_s_hat.bias,
                at.W_k.weight, at.W_k.bias, at.W_q.weight, at.W_q.bias]

    def word_operands(self, f_w, f_s, query_mask):
        """Per-sample word-side operands of the attention core: (Mq, uq, what, shat, qmask rows) -- one HIP launch
        (csrc/word_prep.hip): what = linear_w_hat(f_w) * qmask, shat = linear_s_hat(f_s), kb = W_k(what), and the clip side's W_q
        folded onto the words, W_q(c_hat) . kb^T == c_hat . (kb W_q.weight)^T + kb . W_q.bias = c_hat . Mq^T + uq."""
        qm = _rows(query_mask)
        what, shat, Mq, uq = WordPrepFn.apply(f_w, f_s, qm, *self.word_params())
        return Mq, uq, what, shat, qm

    def forward_packed(self, fc, hbar, f_w, f_s, query_mask, layout, fcmean_in=None):
        Mq, uq, what, shat, qm = self.word_operands(f_w, f_s, query_mask)
        return ContentUnitFn.apply(fc, hbar, self.linear_c_hat.weight, self.linear_c_hat.bias, Mq, uq, what, shat, qm,
                                   self.linear_c.weight, self.linear_c.bias, layout, fcmean_in)

    @_hip_forward
    def forward(self, f_c, f_w, f_s, f_m, query_mask, moment_mask):
        layout = CellLayout.all_cells(moment_mask)
        hbar = GateFn.apply(layout.pack(f_m), f_s, layout)[0]
        out, _ = self.forward_packed(layout.pack(f_c), hbar, f_w, f_s, query_mask, layout)
        return layout.unpack(out)


class MomentUnit(nn.Module):
    """reference models.py:278-303."""

    def __init__(self, D):
        super().__init__()
        self.D = D
        self.conv_layer_fb = nn.Conv2d(D, D, 1)
        self.conv_layer_fc = nn.Conv2d(D, D, 1)

    def cat_weights(self):
        """(Wcat [D, 2D], bcat [D]): the two 1x1 convolutions as one contraction over [f_b[i]*f_b[j] | mean_c f_c]."""
        D = self.D
        Wcat = torch.cat([self.conv_layer_fb.weight.view(D, D), self.conv_layer_fc.weight.view(D, D)], dim=1)
        return Wcat, self.conv_layer_fb.bias + self.conv_layer_fc.bias

    def forward_packed(self, fcmean, fm, f_b, layout):
        Wcat, bcat = self.cat_weights()
        return MomentUnitFn.apply(fcmean, fm, f_b, Wcat, bcat, layout)[0]

    @_hip_forward
    def forward(self, f_c, f_m, f_b, moment_mask):
        layout = CellLayout.all_cells(moment_mask)
        mu = self.forward_packed(layout.pack(f_c).mean(dim=1), layout.pack(f_m), f_b, layout)
        return layout.unpack(mu)


class SMI(nn.Module):
    """reference models.py:305-322: content and boundary units read the layer inputs, the moment unit their outputs."""

    def __init__(self, D, dl):
        super().__init__()
        self.D, self.dl = D, dl
        self.content_unit = ContentUnit(D, dl)
        self.boundary_unit = BoundaryUnit(D)
        self.moment_unit = MomentUnit(D)

    def forward_packed(self, fc, fm, f_b, f_w, f_s, query_mask, length_mask, layout, fcmean_in=None):
        """fcmean_in = mean_c fc marks the final layer of a stack (its content output is consumed only through the
        clip mean); the returned content tensor is then empty."""
        hbar_c, hbar_b, fm_res = GateFn.apply(fm, f_s, layout)    # sigmoid(fm*fs)*fm for both units + fm for the residual
        cu, cumean = self.content_unit.forward_packed(fc, hbar_c, f_w, f_s, query_mask, layout, fcmean_in)
        bu = self.boundary_unit.forward_packed(f_b, f_w, f_s, hbar_b, query_mask, length_mask, layout)
        mu = self.moment_unit.forward_packed(cumean, fm_res, bu, layout)
        return cu, mu, bu, cumean

    @_hip_forward
    def forward(self, f_c, f_m, f_b, f_w, f_s, query_mask, length_mask, moment_mask):
        layout = CellLayout.all_cells(moment_mask)
        cu, mu, bu, _ = self.forward_packed(layout.pack(f_c), layout.pack(f_m), f_b, f_w, f_s, query_mask, length_mask, layout)
        return layout.unpack(cu), layout.unpack(mu), bu


class Localization(nn.Module):
    """reference models.py:324-344."""

    def __init__(self, D):
        super().__init__()
        self.conv_layer_pm = nn.Conv2d(D, 1, 1)
        self.conv_layer_ps = nn.Conv1d(D, 1, 1)
        self.conv_layer_pe = nn.Conv1d(D, 1, 1)
        self.conv_layer_pa = nn.Conv1d(D, 1, 1)
        self.sigmoid = nn.Sigmoid()

    def forward_packed(self, fm, f_b, length_mask, layout):
        D = f_b.shape[-1]
        heads = (self.conv_layer_ps, self.conv_layer_pe, self.conv_layer_pa)
        wb = torch.stack([h.weight.view(D) for h in heads])
        bb = torch.cat([h.bias for h in heads])
        pm, psea = ScoreMapFn.apply(fm, f_b, self.conv_layer_pm.weight.view(D), self.conv_layer_pm.bias, wb, bb,
                                    length_mask.float(), layout)
        return pm, psea[0], psea[1], psea[2]

    @_hip_forward
    def forward(self, f_m, f_b, length_mask, moment_mask):
        layout = CellLayout.all_cells(moment_mask)
        return self.forward_packed(layout.pack(f_m), f_b, length_mask, layout)


class SMIN(nn.Module):
    """reference models.py:346-377 -- the drop-in boundary (ctor called positionally from main.py:71)."""

    def __init__(self, T, L, C, D, dl, num_smi_layers, input_video_dim, max_query_length, lstm_hidden_size, device='cpu'):
        super().__init__()
        self.T, self.L, self.C, self.D, self.dl = T, L, C, D, dl
        self.num_smi_layers, self.input_video_dim = num_smi_layers, input_video_dim
        self.max_query_length, self.lstm_hidden_size, self.device = max_query_length, lstm_hidden_size, device
        if D != 2 * lstm_hidden_size:
            raise ValueError("SMIN needs D == 2 * lstm_hidden_size (reference models.py:62,81)")
        # shape limits of the HIP kernels behind this module (the reference has none): named here, not as a bare
        # "argument rejected at csrc line N" from the middle of a forward pass
        limits = [(T % L == 0, f"L | T (T={T}, L={L}): the reference's AvgPool1d gives {T // max(T // L, 1)} != L boundary rows otherwise (SURVEY 8a-2)"),
                  (2 <= C <= 4, f"2 <= C <= 4 clips per moment (C={C}): csrc/content_attn.hip keeps the clips of a cell in one lane quad"),
                  (D % 4 == 0, f"D % 4 == 0 (D={D}): 16-byte row segments everywhere"),
                  (dl % 16 == 0 and 16 <= dl <= 128, f"dl a multiple of 16 in [16, 128] (dl={dl}): csrc/content_attn.hip tiles the attention rows in 16-feature blocks, csrc/word_prep.hip holds dl <= 128"),
                  (1 <= max_query_length <= 32, f"max_query_length <= 32 (got {max_query_length}): the word-side kernels keep a query in 32 LDS slots"),
                  (num_smi_layers >= 1, f"num_smi_layers >= 1 (got {num_smi_layers})")]
        bad = [msg for ok, msg in limits if not ok]
        if bad:
            raise ValueError("SMIN: outside the limits of the HIP kernels -- needs " + "; ".join(bad))
        self.backbone = Backbone(T, D, input_video_dim, max_query_length, lstm_hidden_size, device)
        self.pgm = ProposalGeneration(T, L, C, device)
        self.smis = nn.ModuleList([SMI(D, dl) for _ in range(num_smi_layers)])
        self.localization = Localization(D)

    native_host = True             # run the in-model path as ONE torch-extension call (csrc/torch_binding.cpp); False: Python host
    async_weights = True           # ... whose weight-gradient contractions run on a low-priority stream of their own
    known_cell_count = None        # number of valid cells of the next batches' moment_mask, when the caller knows it: the forward then
                                   # asks the device nothing (training.CapturedStep); a wrong value is flagged, see csrc/layout.hip
    tail_split = True              # the backward's closing chains on streams of their own (torch_binding.cpp "the tail"); training.CapturedStep
                                   # turns it off: a process that has used the extra streams replays its graphs ~1 ms/step slower
    grad_sync = False              # data parallel: the one-node backward averages its gradients over the process group itself, group by
                                   # group as they become final (set by distributed.wrap; torch_binding.cpp GradSync)
    bf16_operand_storage = True    # under set_gemm_mode("bf16"): tensors that only feed contractions are stored as bf16 (no bit of the step changes)
    fused_core = True              # ... with proposal map + SMI layers + localization as one autograd node (False: a node per module)
    content_stream = True          # dl < D: keep the content stream in the dl-dimensional space (see _forward_stream)
    overlap_boundary = True        # boundary unit on a second HIP stream beside the content stream
    overlap_prep = True            # parameter-only work (word-side operands, weight products) on that stream as well

    def _forward_stream(self, f, fs, fw, query_mask, length_mask, layout):
        """The same network with the content unit's two linear maps re-associated (exact in real arithmetic).

        The unit output  f_c' = cc Wc^T + bc + f_c + hbar  (models.py:269-276) is consumed only by the next unit's
        linear_c_hat (models.py:247) and, through its clip mean, by the moment unit (models.py:295); f_c itself starts
        as clip means of f (models.py:117).  All of that is linear, so with g_k = f Wch_k^T
            chat_k = clip_means(g_k) + sum_{l<k} cc_l (Wch_k Wc_l)^T + (sum_{l<k} hbar_l) Wch_k^T + const_k
            mean_c f_c^k = mean_c f_c^{k-1} + (mean_c cc_k) Wc_k^T + bc_k + hbar_k
        and the (N*C) x D tensors f_c never exist: every contraction over N*C rows is dl x dl instead of D x dl.
        The parameter products (Wch_k Wc_l, Wch_k bc_l, g_k) are tiny and stay in torch, which also routes their
        gradients back to the reference's parameters."""
        T, L, C, dl = self.T, self.L, self.C, self.dl
        N = layout.N
        nl = len(self.smis)
        cus = [smi.content_unit for smi in self.smis]
        cur = torch.cuda.current_stream(f.device)
        side = _side_stream(f.device) if self.overlap_boundary else cur
        # Everything that depends only on parameters and on the query encoding -- the word-side operands of every layer,
        # the dl x dl weight products, constants and concatenations: ~100 tiny launches forward, more backward -- is
        # formed up front on the second stream.  Their backward nodes then run there too, off the main stream's chain
        # (nothing on the critical path waits for a parameter gradient).
        prep = side if (self.overlap_prep and self._streams_allowed("torch")) else cur
        prep.wait_stream(cur)
        with torch.cuda.stream(prep):
            consts, bsum = [], None
            for cu in cus:
                consts.append(cu.linear_c_hat.bias if bsum is None else cu.linear_c_hat.bias + torch.mv(cu.linear_c_hat.weight, bsum))
                bsum = cu.linear_c.bias if bsum is None else bsum + cu.linear_c.bias
            Wch_all = torch.cat([cu.linear_c_hat.weight for cu in cus])
            # layer 0's constant rides on the clip means; later layers get theirs in the contraction that forms chat_k,
            # whose weight-gradient pass yields the constant's gradient (a column sum) for free
            const_all = consts[0]
            # every layer's word-side operands in one launch (one node: its backward runs at the tail of the backward pass; a node
            # per layer measured 1.4 ms/step slower, its backward launches take CUs from the layers' matrix kernels)
            qmr = _rows(query_mask)
            wo = WordPrepFn.apply(fw, fs, qmr, *[p for cu in cus for p in cu.word_params()])
            words = [(wo[4 * k + 2], wo[4 * k + 3], wo[4 * k], wo[4 * k + 1], qmr) for k in range(nl)]
            Pcats = [[torch.cat([torch.matmul(cus[k].linear_c_hat.weight, cus[l].linear_c.weight) for l in range(lo, min(lo + 4, k))], dim=1)
                      for lo in range(0, k, 4)] for k in range(nl)]
            mu_w = [smi.moment_unit.cat_weights() for smi in self.smis]
        cur.wait_stream(prep)
        if prep is not cur:
            # allocator bookkeeping for tensors that cross streams: made on one stream, read on the other
            for t in [Wch_all] + consts + [x for w in words for x in w] + [x for ps in Pcats for x in ps] + [x for w in mu_w for x in w]:
                t.record_stream(cur)
        if side is not cur:
            for t in (fw, fs, query_mask, length_mask, layout.cells, layout.row_ptr, layout.cellmap):
                t.record_stream(side)
        fm, fb = ProposalMeansFn.apply(f, layout, T, L, C)
        # every layer's clip-mean term and constant in one pass over f
        g_all = LinearRowsFn.apply(Wch_all, None, None, None, 1, f.reshape(-1, self.D))
        pgs = ClipWindowMeansFn.apply(g_all.view(f.shape[0], T, -1), const_all, layout, T, L, C, nl)
        cumean, H, hist = None, None, []
        for k, smi in enumerate(self.smis):
            last = k == nl - 1
            cu = smi.content_unit
            # every consumer of hbar / fm gets its own view, so the gate's backward kernel sums their gradients:
            # hbar: content stream (clip-mean chain), boundary unit, [the next layer's gate term, the running sum]
            n_hbar = 2 if last else (3 if k > 0 or nl < 3 else 4)
            views = GateFn.apply(fm, fs, layout, n_hbar, 2 if k == 0 else 1)
            hbar_c, hbar_b, fm_res = views[0], views[1], views[n_hbar]
            if k == 0:
                cumean = views[n_hbar + 1]                                   # mean_c f_c of the proposal map is f_m
            # The boundary unit reads only the layer inputs: a chain of small, latency-bound launches that runs on a
            # second HIP stream beside the content stream (whose attention kernels leave most of a CU's registers and
            # LDS free) and joins before the moment unit; autograd replays its backward on the same stream.
            side.wait_stream(cur)
            with torch.cuda.stream(side):
                bu = smi.boundary_unit.forward_packed(fb, fw, fs, hbar_b, query_mask, length_mask, layout)
            if side is not cur:
                fb.record_stream(side)
                hbar_b.record_stream(side)
            Wch = cu.linear_c_hat.weight
            chat = pgs[k]
            for n_part, lo in enumerate(range(0, len(hist), 4)):           # [cc_1 | cc_2 | ..] [Wch Wc_1 | Wch Wc_2 | ..]^T
                part = hist[lo:lo + 4]
                hp = LinearRowsFn.apply(Wch, None, None, None, 1, H) if lo == 0 else None   # (sum_l hbar_l) Wch^T, per cell
                chat = LinearRowsFn.apply(Pcats[k][n_part], consts[k] if lo == 0 else None, chat, hp, C, *[cc_l for cc_l, _, _ in part])
            Mq, uq, what, shat, qm = words[k]
            cc, ccmean = ContentAttnFn.apply(chat, Mq, uq, what, shat, qm, layout, C, not last)
            cumean = LinearRowsFn.apply(cu.linear_c.weight, cu.linear_c.bias, cumean, hbar_c, 1, ccmean)
            if not last:
                if H is None:
                    H = views[2]                                             # read by the next layer's gate term
                    Hsum = views[3] if n_hbar > 3 else views[2]              # and by the running sum after that
                else:
                    H = Hsum + views[2]
                    Hsum = H
                hist.append((cc, cu.linear_c.weight, cu.linear_c.bias))
            cur.wait_stream(side)
            if side is not cur:
                bu.record_stream(cur)
            fm, cumean = MomentUnitFn.apply(cumean, fm_res, bu, mu_w[k][0], mu_w[k][1], layout)
            fb = bu
        return self.localization.forward_packed(fm, fb, length_mask, layout)

    def _native_params(self):
        """Parameters in the order csrc/torch_binding.cpp reads them: video encoder 3, LSTM 16, 20 per SMI layer, localization 8."""
        ve, lstm = self.backbone.videoencoder, self.backbone.queryencoder.lstm
        ps = [ve.ve.weight, ve.ve.bias, ve.pe.weight]
        for layer in range(2):
            for sfx in ("", "_reverse"):
                ps += [getattr(lstm, f"{n}_l{layer}{sfx}") for n in ("weight_ih", "weight_hh", "bias_ih", "bias_hh")]
        for smi in self.smis:
            cu, bu, mu = smi.content_unit, smi.boundary_unit.attn_layer, smi.moment_unit
            for m in (cu.linear_c_hat, cu.linear_w_hat, cu.linear_s_hat, cu.linear_c, cu.attn_layer.W_q, cu.attn_layer.W_k, bu.W_q, bu.W_k,
                      mu.conv_layer_fb, mu.conv_layer_fc):
                ps += [m.weight, m.bias]
        lo = self.localization
        for m in (lo.conv_layer_pm, lo.conv_layer_ps, lo.conv_layer_pe, lo.conv_layer_pa):
            ps += [m.weight, m.bias]
        return ps

    @staticmethod
    def _streams_allowed(kind="library"):
        """Which work may run on the second / third HIP stream beside the main one.

        Measured on gfx950 (tools/bu_concurrent_probe.py): a wave executing PACKED fp32 arithmetic (v_pk_fma_f32 ...) on
        a SIMD where a wave of another kernel runs v_mfma_f32_32x32x16_bf16 occasionally gets a wrong lane result.  This library is
        built without packed fp32 arithmetic (csrc/Makefile), so its own kernels may overlap in every contraction mode.  torch's
        kernels (element-wise ops, hipBLASLt, the optimizer) are outside that guarantee: with the bf16-core contraction modes
        (f32e / bf16x3 / bf16) the parameter-only torch work (kind="torch") stays on the main stream, where nothing runs beside it."""
        from . import _lib
        return kind == "library" or _lib.get_gemm_mode() == "f32" or bool(os.environ.get("SMIN_STREAMS_IN_ALL_MODES"))

    def _prep_is_library_code(self):
        """The one-node step forms the parameter products with csrc/param_prep.hip when its shape limits hold (torch_binding.cpp):
        then nothing of torch runs on the second stream and the overlap is safe in every contraction mode."""
        return self.fused_core and self.D % 32 == 0 and self.D <= 1056 and self.dl % 32 == 0 and len(self.smis) <= 8

    def _native_ok(self, video_features, query_features):
        """The torch-extension path covers the production configuration: content stream on a mask-driven cell list, fused
        BiLSTM and video encoder kernels.  Anything else (dl >= D, C outside 2..4, > 8 layers, H > 256, odd widths) runs the
        same kernels from the Python host below."""
        H, ve, nl = self.lstm_hidden_size, self.backbone.videoencoder, len(self.smis)
        return (self.native_host and self.content_stream and self.dl < self.D and 2 <= self.C <= 4 and 1 <= nl <= 8 and nl * self.dl <= 2048
                and self.backbone.queryencoder.fused_lstm and H <= 256 and H % 4 == 0
                and video_features.dtype == torch.float32 and query_features.dtype == torch.float32 and ve.d0 % 4 == 0 and ve.d % 4 == 0
                and video_features.shape[1] == self.T and video_features.shape[1] <= ve.pe.weight.shape[0])

    @_hip_forward
    def forward(self, video_features, video_mask, query_features, query_mask, length_mask, moment_mask):
        # the reference's dataset pads queries and their mask to max_query_length (dataset.py:35, 173); a batch cut to its longest query is taken
        # too: the encoder pads f_w as models.py:58-59 does, and the mask is padded here (every kernel reads max_query_length mask columns)
        query_mask = query_mask.reshape(query_features.shape[0], -1)
        if query_mask.shape[1] != query_features.shape[1] or query_mask.shape[1] > self.max_query_length:
            raise ValueError(f"query_mask has {query_mask.shape[1]} columns for {query_features.shape[1]} words (max_query_length {self.max_query_length})")
        if query_mask.shape[1] < self.max_query_length:
            query_mask = torch.nn.functional.pad(query_mask, (0, self.max_query_length - query_mask.shape[1]))
        if self.grad_sync and not self._streams_allowed("torch"):
            raise RuntimeError("SMIN.grad_sync: the in-node gradient exchange runs RCCL beside the contraction kernels and is limited to the exact "
                               "fp32 mode; call distributed.wrap after set_gemm_mode (it then uses torch DDP)")
        if self.grad_sync and not (self._native_ok(video_features, query_features) and self.fused_core):
            raise RuntimeError("SMIN.grad_sync (distributed.wrap's in-node gradient exchange) needs the one-node extension path; this "
                               "call does not qualify (see SMIN._native_ok) -- wrap the model with SMIN_TORCH_DDP=1 instead")
        if self._native_ok(video_features, query_features):
            from . import _lib
            cfg = [self.T, self.L, self.C, self.D, self.dl, len(self.smis), self.max_query_length, self.lstm_hidden_size,
                   int(self.overlap_boundary), int(self.overlap_prep and (self._streams_allowed("torch") or self._prep_is_library_code())), int(self.fused_core),
                   int(self.async_weights), int(self.bf16_operand_storage), int(self.grad_sync and torch.is_grad_enabled()),
                   -1 if self.known_cell_count is None else int(self.known_cell_count), int(self.tail_split)]
            return _lib.load_torch().smin_forward(video_features, video_mask, query_features, query_mask, length_mask, moment_mask,
                                                  self._native_params(), cfg)
        pending = CellLayout.begin(moment_mask)                    # work is driven by moment_mask (SURVEY 8a-0 caveat)
        f, fs, fw = self.backbone(video_features, video_mask, query_features, query_mask)
        layout = pending.finish()                                  # the only host sync of a step; hidden behind the backbone
        if (self.content_stream and self.dl < self.D and 2 <= self.C <= 4 and layout.all_valid
                and len(self.smis) <= 8 and len(self.smis) * self.dl <= 2048):    # limits of the clip-window-means launch
            return self._forward_stream(f, fs, fw, query_mask, length_mask, layout)
        fc, fm, fb = self.pgm.forward_packed(f, layout)
        fcmean = fm                                                # mean_c fc: the map's f_m, then each layer's clip mean
        for k, smi in enumerate(self.smis):
            last = k == len(self.smis) - 1                        # the last layer's content output feeds nothing but its mean
            fc, fm, fb, fcmean = smi.forward_packed(fc, fm, fb, fw, fs, query_mask, length_mask, layout, fcmean if last else None)
        return self.localization.forward_packed(fm, fb, length_mask, layout)
