"""autograd.Function wrappers around the C ABI (include/smin_hip.h).

Each Function owns one reference module body (fwd) and the backward torch autograd would have derived
for it; they compose under ordinary autograd, so DistributedDataParallel hooks fire as usual.
All tensors are fp32, contiguous, on one HIP device; kernels launch on torch's current stream."""
import torch
from torch.autograd import Function

from . import _lib
from ._lib import call, ptr, stream


def _c(t):
    return None if t is None else t.contiguous()


def _ws(nbytes, device):
    buf = _lib.workspace(nbytes, device)
    return buf, ptr(buf), buf.numel()


def _unit_ws(layout, C, D, dl, Nq, device):
    n = _lib.load().smin_workspace_bytes(layout.N, layout.B, C, D, dl, Nq)
    return _ws(n, device)


_EVENT_TABLES = {}


def clip_event_table(device, T, L, C):
    """The geometry's clip-boundary table (offsets int32 [T+1], entries int32 [n, 2]), built once per (device, T, L, C)
    and kept -- the counterpart of the reference's cached ``self.Wc`` buffer (models.py:110), 8 bytes per clip
    boundary instead of a dense (L, L, C, T) matrix."""
    key = (device.type, device.index, T, L, C)
    tab = _EVENT_TABLES.get(key)
    if tab is None:
        counts = torch.empty(T, dtype=torch.int32, device=device)
        call("smin_clip_event_table", stream(), T, L, C, ptr(counts), None, None)
        offsets = torch.zeros(T + 1, dtype=torch.int32, device=device)
        offsets[1:] = torch.cumsum(counts, 0).to(torch.int32)
        table = torch.empty((max(1, int(offsets[-1])), 2), dtype=torch.int32, device=device)
        call("smin_clip_event_table", stream(), T, L, C, None, ptr(offsets), ptr(table))
        tab = _EVENT_TABLES[key] = (offsets, table)
    return tab


class ProposalMapFn(Function):
    """ProposalGeneration.forward (reference models.py:115-126)."""

    @staticmethod
    def forward(ctx, f, layout, T, L, C):
        f = _c(f)
        B, Tn, D = f.shape
        if Tn != T:
            raise ValueError(f"ProposalGeneration was built for T={T} but got {Tn} frames")
        fc = f.new_empty((layout.N, C, D))
        fm = f.new_empty((layout.N, D))
        fb = f.new_empty((B, L, D))
        _, wp, wn = _ws(8 * B * (T + 1) * D, f.device)
        call("smin_proposal_map_fwd", stream(), ptr(f), ptr(layout.cells), layout.N, B, T, L, C, D, ptr(fc), ptr(fm), ptr(fb), wp, wn)
        ctx.layout, ctx.dims = layout, (B, T, L, C, D)
        return fc, fm, fb

    @staticmethod
    def backward(ctx, dfc, dfm, dfb):
        layout = ctx.layout
        B, T, L, C, D = ctx.dims
        dfc, dfm, dfb = _c(dfc), _c(dfm), _c(dfb)
        ref = dfc if dfc is not None else dfm if dfm is not None else dfb
        df = ref.new_empty((B, T, D))
        _, wp, wn = _ws(4 * B * T * D, df.device)
        eo, et = clip_event_table(df.device, T, L, C)
        call("smin_proposal_map_bwd", stream(), ptr(dfc), ptr(dfm), ptr(dfb), ptr(layout.cells), ptr(layout.row_ptr), ptr(layout.cellmap),
             layout.N, B, T, L, C, D, ptr(df), wp, wn, ptr(eo), ptr(et))
        return df, None, None, None, None


class ProposalMeansFn(Function):
    """ProposalGeneration.forward without f_c: (f_m, f_b) only (content stream; f_c is never formed)."""

    @staticmethod
    def forward(ctx, f, layout, T, L, C):
        f = _c(f)
        B, Tn, D = f.shape
        if Tn != T:
            raise ValueError(f"ProposalGeneration was built for T={T} but got {Tn} frames")
        fm = f.new_empty((layout.N, D))
        fb = f.new_empty((B, L, D))
        _, wp, wn = _ws(8 * B * (T + 1) * D, f.device)
        call("smin_proposal_map_fwd", stream(), ptr(f), ptr(layout.cells), layout.N, B, T, L, C, D, None, ptr(fm), ptr(fb), wp, wn)
        ctx.layout, ctx.dims = layout, (B, T, L, C, D)
        return fm, fb

    @staticmethod
    def backward(ctx, dfm, dfb):
        layout = ctx.layout
        B, T, L, C, D = ctx.dims
        dfm, dfb = _c(dfm), _c(dfb)
        ref = dfm if dfm is not None else dfb
        df = ref.new_empty((B, T, D))
        _, wp, wn = _ws(4 * B * T * D, df.device)
        eo, et = clip_event_table(df.device, T, L, C)
        call("smin_proposal_map_bwd", stream(), None, ptr(dfm), ptr(dfb), ptr(layout.cells), ptr(layout.row_ptr), ptr(layout.cellmap),
             layout.N, B, T, L, C, D, ptr(df), wp, wn, ptr(eo), ptr(et))
        return df, None, None, None, None


class ClipWindowMeansFn(Function):
    """out[s][n, c] = mean over clip c of cell n of g[b, t, s*W:(s+1)*W] + bias[s*W:(s+1)*W], one tensor per segment s
    (bias may cover only the first k segments: k*W entries)
    -- every layer's linear_c_hat applied to ProposalGeneration's f_c (reference models.py:117, 247) through
    g = f [Wch_1; ..]^T, without forming f_c.  Mask-driven layouts only."""

    @staticmethod
    def forward(ctx, g, bias, layout, T, L, C, nseg):
        g, bias = _c(g), _c(bias)
        B, Tn, D = g.shape
        W = D // nseg
        out = g.new_empty((nseg, layout.N * C, W))
        _, wp, wn = _ws(8 * B * (T + 1) * D, g.device)
        nb = 0 if bias is None else bias.numel()                     # bias may cover only the first segments
        call("smin_clip_window_means_fwd", stream(), ptr(g), ptr(bias), nb, ptr(layout.cells), layout.N, B, T, L, C, W, nseg, ptr(out), wp, wn)
        ctx.layout, ctx.dims, ctx.nb = layout, (B, T, L, C, W, nseg), nb
        return tuple(out[s] for s in range(nseg))

    @staticmethod
    def backward(ctx, *douts):
        import ctypes
        layout = ctx.layout
        B, T, L, C, W, nseg = ctx.dims
        ref = next(d for d in douts if d is not None)
        douts = [_c(d) if d is not None else ref.new_zeros((layout.N * C, W)) for d in douts]
        dg = ref.new_empty((B, T, W * nseg))
        _, wp, wn = _ws(4 * B * T * W * nseg, ref.device)
        arr = (ctypes.c_void_p * nseg)(*[d.data_ptr() for d in douts])
        eo, et = clip_event_table(ref.device, T, L, C)
        call("smin_clip_window_means_bwd", stream(), arr, ptr(layout.cells), ptr(layout.row_ptr), ptr(layout.cellmap),
             layout.N, B, T, L, C, W, nseg, ptr(dg), wp, wn, ptr(eo), ptr(et))
        dbias = torch.cat([d.sum(dim=0) for d in douts[:ctx.nb // W]]) if (ctx.needs_input_grad[1] and ctx.nb) else None
        return dg, dbias, None, None, None, None, None


class ContentAttnFn(Function):
    """The content unit's attention core alone (reference models.py:252-267): chat [N*C, dl] -> (cc [N*C, dl],
    ccmean [N, dl] = mean_c cc).  want_rows=False (last layer) returns an empty cc."""

    @staticmethod
    def forward(ctx, chat, Mq, uq, what, shat, qmask, layout, C, want_rows):
        chat, Mq, uq, what, shat, qmask = map(_c, (chat, Mq, uq, what, shat, qmask))
        B, Nq, dl = what.shape
        N = layout.N
        cc = chat.new_empty((N * C, dl) if want_rows else (0, dl))
        ccmean = chat.new_empty((N, dl))
        call("smin_content_attn_fwd", stream(), ptr(chat), ptr(layout.cells), ptr(layout.row_ptr), N, B, layout.L, C, dl, Nq,
             ptr(Mq), ptr(uq), ptr(what), ptr(shat), ptr(qmask), ptr(cc) if want_rows else None, ptr(ccmean))
        ctx.save_for_backward(chat, Mq, uq, what, shat, qmask)
        ctx.layout, ctx.C, ctx.want_rows = layout, C, want_rows
        if not want_rows:
            ctx.mark_non_differentiable(cc)
        return cc, ccmean

    @staticmethod
    def backward(ctx, dcc, dccmean):
        chat, Mq, uq, what, shat, qmask = ctx.saved_tensors
        layout, C = ctx.layout, ctx.C
        B, Nq, dl = what.shape
        N = layout.N
        dcc = _c(dcc) if ctx.want_rows else None
        dccmean = _c(dccmean)
        if dcc is None and dccmean is None:
            dccmean = chat.new_zeros((N, dl))
        dchat = torch.empty_like(chat)
        dMq, duq, dwhat, dshat = torch.empty_like(Mq), torch.empty_like(uq), torch.empty_like(what), torch.empty_like(shat)
        if N == 0:
            for t in (dMq, duq, dwhat, dshat):
                t.zero_()
        else:
            _, wp, wn = _ws(_lib.load().smin_content_attn_bwd_workspace_bytes(N, B, C, dl), chat.device)
            call("smin_content_attn_bwd", stream(), ptr(dcc), ptr(dccmean), ptr(chat), ptr(layout.cells), ptr(layout.row_ptr),
                 N, B, layout.L, C, dl, Nq, ptr(Mq), ptr(uq), ptr(what), ptr(shat), ptr(qmask),
                 ptr(dchat), ptr(dMq), ptr(duq), ptr(dwhat), ptr(dshat), wp, wn)
        return dchat, dMq, duq, dwhat, dshat, None, None, None, None


def _ptr_array(tensors):
    import ctypes
    return (ctypes.c_void_p * len(tensors))(*[t.data_ptr() for t in tensors])


class LinearRowsFn(Function):
    """y[r] = [x_0[r] | x_1[r] | ..] W^T + bias + add_rows[r] + add_cells[r // C]
    (each x_s [R, K], W [O, nseg*K]; bias and the addends are optional).  Call as
    ``LinearRowsFn.apply(W, bias, add_rows, add_cells, C, x_0, x_1, ...)``."""

    @staticmethod
    def forward(ctx, W, bias, add_rows, add_cells, C, *xs):
        W, bias, add_rows, add_cells = map(_c, (W, bias, add_rows, add_cells))
        xs = tuple(_c(x) for x in xs)
        R, K = xs[0].shape
        O = W.shape[0]
        y = xs[0].new_empty((R, O))
        for x in xs:
            ptr(x)                                                  # device / contiguity checks
        call("smin_linear_rows_fwd", stream(), _ptr_array(xs), len(xs), ptr(W), ptr(bias), ptr(add_rows), ptr(add_cells), C, R, O, K, ptr(y))
        ctx.save_for_backward(W, *xs)
        ctx.C = C
        return y

    @staticmethod
    def backward(ctx, dy):
        W, *xs = ctx.saved_tensors
        dy = _c(dy)
        R, K = xs[0].shape
        O, nseg = W.shape[0], len(xs)
        need = ctx.needs_input_grad
        want_dx = any(need[5:])
        dxs = [torch.empty_like(x) for x in xs] if want_dx else None
        dW = torch.empty_like(W)
        dbias = dy.new_empty((O,)) if need[1] else None
        _, wp, wn = _ws(_lib.load().smin_linear_rows_bwd_workspace_bytes(R, O, nseg * K), dy.device)
        call("smin_linear_rows_bwd", stream(), ptr(dy), _ptr_array(xs), nseg, ptr(W.t().contiguous()), R, O, K,
             _ptr_array(dxs) if want_dx else None, ptr(dW), ptr(dbias), wp, wn)
        dcells = None
        if need[3]:
            if ctx.C == 1:
                dcells = dy
            else:
                dcells = dy.new_empty((R // ctx.C, O))
                call("smin_group_sum", stream(), ptr(dy), R // ctx.C, ctx.C, O, ptr(dcells))
        return (dW, dbias, (dy if need[2] else None), dcells, None) + (tuple(dxs) if want_dx else (None,) * nseg)


def _ptr_array_opt(tensors):
    import ctypes
    return (ctypes.c_void_p * len(tensors))(*[None if t is None else t.data_ptr() for t in tensors])


class WordPrepFn(Function):
    """Word-side operands of every layer's content attention in one launch (reference models.py:249-251, 209-211):
    ``WordPrepFn.apply(fw, fs, qmask, *params)`` with 8 parameters per layer (linear_w_hat.{weight,bias}, linear_s_hat.{..},
    attn_layer.W_k.{..}, attn_layer.W_q.{..}) returns (what_k, shat_k, Mq_k, uq_k) for every layer, flattened."""

    @staticmethod
    def forward(ctx, fw, fs, qmask, *params):
        fw, fs, qmask = _c(fw), _c(fs), _c(qmask)
        params = tuple(_c(p) for p in params)
        nl = len(params) // 8
        B, Nq, D = fw.shape
        dl = params[0].shape[0]
        what, kb, Mq = (fw.new_empty((nl, B, Nq, dl)) for _ in range(3))
        shat, uq = fw.new_empty((nl, B, dl)), fw.new_empty((nl, B, Nq))
        for p in params:
            ptr(p)
        call("smin_word_prep_fwd", stream(), ptr(fw), ptr(fs), ptr(qmask), _ptr_array(params), nl, B, Nq, D, dl, ptr(what), ptr(shat), ptr(kb), ptr(Mq), ptr(uq))
        ctx.save_for_backward(fw, fs, qmask, what, kb, *params)
        outs = []
        for k in range(nl):
            outs += [what[k], shat[k], Mq[k], uq[k]]
        return tuple(outs)

    @staticmethod
    def backward(ctx, *grads):
        fw, fs, qmask, what, kb, *params = ctx.saved_tensors
        nl = len(params) // 8
        B, Nq, D = fw.shape
        dl = params[0].shape[0]
        g = [None if x is None else _c(x) for x in grads]
        for x in g:
            ptr(x)
        dfw, dfs = torch.empty_like(fw), torch.empty_like(fs)
        dparams = [torch.empty_like(p) for p in params]
        _, wp, wn = _ws(_lib.load().smin_word_prep_bwd_workspace_bytes(nl, B, Nq, D, dl), fw.device)
        call("smin_word_prep_bwd", stream(), _ptr_array_opt(g[0::4]), _ptr_array_opt(g[1::4]), _ptr_array_opt(g[2::4]), _ptr_array_opt(g[3::4]),
             ptr(fw), ptr(fs), ptr(qmask), ptr(what), ptr(kb), _ptr_array(params), nl, B, Nq, D, dl, ptr(dfw), ptr(dfs), _ptr_array(dparams), wp, wn)
        return (dfw, dfs, None) + tuple(dparams)


class VideoFuseFn(Function):
    """f = ((x W^T + b + pe[t]) * vmask) * f_s  -- VideoEncoder.forward and the backbone's Hadamard product (reference
    models.py:25-36, 81-83) as one contraction with a fused epilogue.  x [B, T, Din] (no gradient), pe [T_emb, D]."""

    @staticmethod
    def forward(ctx, x, W, bias, pe, vmask, fs):
        x, W, bias, pe, vmask, fs = map(_c, (x, W, bias, pe, vmask, fs))
        B, T, Din = x.shape
        D = W.shape[0]
        fv, f = x.new_empty((B, T, D)), x.new_empty((B, T, D))
        call("smin_video_encoder_fwd", stream(), ptr(x), ptr(W), ptr(bias), ptr(pe), ptr(vmask), ptr(fs), B, T, Din, D, ptr(fv), ptr(f))
        ctx.save_for_backward(x, fv, fs, vmask)
        ctx.dims, ctx.pe_rows = (B, T, Din, D), pe.shape[0]
        return f

    @staticmethod
    def backward(ctx, df):
        x, fv, fs, vmask = ctx.saved_tensors
        B, T, Din, D = ctx.dims
        df = _c(df)
        dW, dbias = x.new_empty((D, Din)), x.new_empty((D,))
        dpe = x.new_zeros((ctx.pe_rows, D)) if ctx.pe_rows != T else x.new_empty((T, D))
        dfs = torch.empty_like(fs)
        _, wp, wn = _ws(_lib.load().smin_video_encoder_bwd_workspace_bytes(B, T, Din, D), x.device)
        call("smin_video_encoder_bwd", stream(), ptr(df), ptr(fv), ptr(fs), ptr(vmask), ptr(x), B, T, Din, D,
             ptr(dW), ptr(dbias), ptr(dpe), ptr(dfs), wp, wn)
        return None, dW, dbias, dpe, None, dfs


class BiLstmLayerFn(Function):
    """One bidirectional LSTM layer over a padded batch with per-sample lengths (reference models.py:46-58: nn.LSTM on
    pack_padded_sequence / pad_packed_sequence).  x [B, Nq, In], length int32 [B] (device) -> [B, Nq, 2H], zero at
    padded positions.  Parameter order as nn.LSTM names them: (w_ih, w_hh, b_ih, b_hh) forward, then reverse."""

    @staticmethod
    def forward(ctx, x, length, w_ih_f, w_hh_f, b_ih_f, b_hh_f, w_ih_r, w_hh_r, b_ih_r, b_hh_r):
        x = _c(x)
        B, Nq, In = x.shape
        H = w_hh_f.shape[1]
        Wih = torch.cat([w_ih_f, w_ih_r]).contiguous()                       # [8H, In]
        bias = torch.cat([b_ih_f + b_hh_f, b_ih_r + b_hh_r]).contiguous()   # [8H]
        Whh = torch.stack([w_hh_f, w_hh_r]).contiguous()                    # [2, 4H, H]
        W4 = Whh.view(2, 4, H, H).permute(0, 3, 2, 1).contiguous()          # [2, k, u, gate]
        G = x.new_empty((B, Nq, 2, 4 * H))
        Hout = x.new_empty((B, Nq, 2 * H))
        Cs = x.new_empty((B, Nq, 2, H))
        call("smin_bilstm_layer_fwd", stream(), ptr(x), ptr(Wih), ptr(bias), ptr(W4), ptr(length), B, Nq, In, H, ptr(G), ptr(Hout), ptr(Cs))
        ctx.save_for_backward(x, length, Hout, G, Cs, Wih, Whh)
        return Hout

    @staticmethod
    def backward(ctx, dH):
        x, length, Hout, G, Cs, Wih, Whh = ctx.saved_tensors
        B, Nq, In = x.shape
        H = Whh.shape[2]
        dH = _c(dH)
        dX = torch.empty_like(x) if ctx.needs_input_grad[0] else None
        dWih, dbias, dWhh = torch.empty_like(Wih), x.new_empty((8 * H,)), torch.empty_like(Whh)
        _, wp, wn = _ws(_lib.load().smin_bilstm_layer_bwd_workspace_bytes(B, Nq, In, H), x.device)
        call("smin_bilstm_layer_bwd", stream(), ptr(dH), ptr(x), ptr(Hout), ptr(G), ptr(Cs), ptr(Wih.t().contiguous()), ptr(Whh), ptr(length),
             B, Nq, In, H, ptr(dX), ptr(dWih), ptr(dbias), ptr(dWhh), wp, wn)
        H4 = 4 * H
        return (dX, None, dWih[:H4], dWhh[0], dbias[:H4], dbias[:H4], dWih[H4:], dWhh[1], dbias[H4:], dbias[H4:])


class GateFn(Function):
    """hbar = sigmoid(fm * fs) * fm -- the gated moment feature of models.py:191 and 272-274, computed once per layer.

    Returns n_hbar views of hbar followed by n_res views of fm (defaults: hbar for the content unit, hbar for the
    boundary unit, fm passed through for the moment unit's residual).  Handing every consumer its own view lets the
    one backward kernel sum their gradients instead of autograd doing it in extra full-size passes."""

    @staticmethod
    def forward(ctx, fm, fs, layout, n_hbar=2, n_res=1):
        fm, fs = _c(fm), _c(fs)
        N, D = fm.shape
        hbar = torch.empty_like(fm)
        call("smin_gate_fwd", stream(), ptr(fm), ptr(fs), ptr(layout.cells), N, D, ptr(hbar))
        ctx.save_for_backward(fm, fs)
        ctx.layout, ctx.n_hbar = layout, n_hbar
        return (hbar,) + tuple(hbar.view_as(hbar) for _ in range(n_hbar - 1)) + tuple(fm.view_as(fm) for _ in range(n_res))

    @staticmethod
    def backward(ctx, *grads):
        fm, fs = ctx.saved_tensors
        layout = ctx.layout
        N, D = fm.shape
        dh = [_c(g) for g in grads[:ctx.n_hbar] if g is not None]
        dr = [_c(g) for g in grads[ctx.n_hbar:] if g is not None]
        if not dh:
            dh = [torch.zeros_like(fm)]
        for g in dh + dr:
            ptr(g)
        dfm, dfs = torch.empty_like(fm), torch.empty_like(fs)
        _, wp, wn = _ws(4 * layout.B * 512 * D + 4096, fm.device)
        call("smin_gate_bwd", stream(), _ptr_array(dh), len(dh), _ptr_array(dr) if dr else None, len(dr), ptr(fm), ptr(fs),
             ptr(layout.row_ptr), N, layout.B, layout.L, D, ptr(dfm), ptr(dfs), wp, wn, None, None, None)
        return dfm, dfs, None, None, None


class ContentUnitFn(Function):
    """ContentUnit.forward + ContentAttention.forward (reference models.py:207-226, 242-276).

    Inputs: fc [N,C,D], hbar [N,D] (GateFn), linear_c_hat (Wch, bch), the folded word-side operands
    Mq/uq/what/shat (see smin_hip.h), qmask [B,Nq] fp32, linear_c (Wc, bc).
    Outputs: fc_out [N,C,D], fcmean [N,D]."""

    @staticmethod
    def forward(ctx, fc, hbar, Wch, bch, Mq, uq, what, shat, qmask, Wc, bc, layout, fcmean_in=None):
        """fcmean_in (= mean_c fc) marks the last SMI layer: nothing consumes fc_out there (models.py:372-375), so
        linear_c and its gradients run on the clip means (N rows instead of N*C) and fc_out is returned empty."""
        fc, hbar, Wch, bch, Mq, uq, what, shat, qmask, Wc, bc = map(_c, (fc, hbar, Wch, bch, Mq, uq, what, shat, qmask, Wc, bc))
        N, C, D = fc.shape
        B, Nq, dl = what.shape
        last = fcmean_in is not None
        fcmean_in = _c(fcmean_in)
        fc_out = fc.new_empty((0, C, D)) if last else torch.empty_like(fc)
        fcmean = fc.new_empty((N, D))
        chat = fc.new_empty((N * C, dl))
        cchat = fc.new_empty((N if last else N * C, dl))
        call("smin_content_unit_fwd", stream(), ptr(fc), ptr(hbar), ptr(layout.cells), ptr(layout.row_ptr), N, B, layout.L, C, D, dl, Nq,
             ptr(Wch), ptr(bch), ptr(Mq), ptr(uq), ptr(what), ptr(shat), ptr(qmask), ptr(Wc), ptr(bc),
             ptr(fcmean_in), int(last), None if last else ptr(fc_out), ptr(fcmean), ptr(chat), ptr(cchat))
        ctx.save_for_backward(fc, Wch, Mq, uq, what, shat, qmask, Wc, chat, cchat)
        ctx.layout, ctx.last = layout, last
        if last:
            ctx.mark_non_differentiable(fc_out)
        return fc_out, fcmean

    @staticmethod
    def backward(ctx, dfc_out, dfcmean):
        fc, Wch, Mq, uq, what, shat, qmask, Wc, chat, cchat = ctx.saved_tensors
        layout = ctx.layout
        N, C, D = fc.shape
        B, Nq, dl = what.shape
        dfc_out = None if ctx.last else _c(dfc_out)
        dfcmean = _c(dfcmean) if dfcmean is not None else fc.new_zeros((N, D))
        WchT, WcT = Wch.t().contiguous(), Wc.t().contiguous()
        dfc, dhbar = torch.empty_like(fc), fc.new_empty((N, D))
        dWch, dbch = torch.empty_like(Wch), fc.new_empty((dl,))
        dMq, duq, dwhat, dshat = torch.empty_like(Mq), torch.empty_like(uq), torch.empty_like(what), torch.empty_like(shat)
        dWc, dbc = torch.empty_like(Wc), fc.new_empty((D,))
        if N == 0:
            for t in (dWch, dbch, dMq, duq, dwhat, dshat, dWc, dbc):
                t.zero_()
        else:
            _, wp, wn = _unit_ws(layout, C, D, dl, Nq, fc.device)
            call("smin_content_unit_bwd", stream(), ptr(dfc_out), ptr(dfcmean), ptr(fc), ptr(layout.cells), ptr(layout.row_ptr),
                 N, B, layout.L, C, D, dl, Nq, ptr(WchT), ptr(Mq), ptr(uq), ptr(what), ptr(shat), ptr(qmask), ptr(WcT),
                 ptr(chat), ptr(cchat), ptr(dfc), ptr(dhbar), ptr(dWch), ptr(dbch), ptr(dMq), ptr(duq), ptr(dwhat), ptr(dshat),
                 ptr(dWc), ptr(dbc), wp, wn, int(ctx.last))
        # fcmean_in is mean_c fc by contract: its gradient (dfcmean) is already folded into dfc as dfcmean / C per clip
        return dfc, dhbar, dWch, dbch, dMq, duq, dwhat, dshat, None, dWc, dbc, None, None


class BoundaryUnitFn(Function):
    """BoundaryUnit.forward with its word attention (reference models.py:137-196), all in HIP."""

    @staticmethod
    def forward(ctx, fb, fw, fs, hbar, Wq, bq, Wk, bk, qmask, lmask, layout):
        fb, fw, fs, hbar, Wq, bq, Wk, bk, qmask, lmask = map(_c, (fb, fw, fs, hbar, Wq, bq, Wk, bk, qmask, lmask))
        B, L, D = fb.shape
        Nq = fw.shape[1]
        out = torch.empty_like(fb)
        Qb, baq, bqv = torch.empty_like(fb), torch.empty_like(fb), torch.empty_like(fb)
        Kb = torch.empty_like(fw)
        P = fb.new_empty((B, L, Nq))
        A = fb.new_empty((B, L, L))
        call("smin_boundary_unit_fwd", stream(), ptr(fb), ptr(fw), ptr(fs), ptr(hbar), ptr(layout.cells), ptr(layout.row_ptr),
             layout.N, B, L, Nq, D, ptr(Wq), ptr(bq), ptr(Wk), ptr(bk), ptr(qmask), ptr(lmask),
             ptr(out), ptr(Qb), ptr(Kb), ptr(P), ptr(baq), ptr(bqv), ptr(A))
        ctx.save_for_backward(fb, fw, fs, hbar, Wq, Wk, qmask, lmask, Qb, Kb, P, baq, bqv, A)
        ctx.layout = layout
        return out

    @staticmethod
    def backward(ctx, dout):
        fb, fw, fs, hbar, Wq, Wk, qmask, lmask, Qb, Kb, P, baq, bqv, A = ctx.saved_tensors
        layout = ctx.layout
        B, L, D = fb.shape
        Nq = fw.shape[1]
        dout = _c(dout)
        WqT, WkT = Wq.t().contiguous(), Wk.t().contiguous()
        dfb, dfw, dfs, dhbar = torch.empty_like(fb), torch.empty_like(fw), torch.empty_like(fs), torch.empty_like(hbar)
        dWq, dbq, dWk, dbk = torch.empty_like(Wq), fb.new_empty((D,)), torch.empty_like(Wk), fb.new_empty((D,))
        nbytes = 4 * (2 * B * L * L + 3 * B * L * D + B * L * Nq + B * Nq * D + 2 * 64 * (D * D + D)) + 4096
        _, wp, wn = _ws(nbytes, fb.device)
        call("smin_boundary_unit_bwd", stream(), ptr(dout), ptr(fb), ptr(fw), ptr(fs), ptr(hbar), ptr(layout.cells), ptr(layout.row_ptr),
             layout.N, B, L, Nq, D, ptr(WqT), ptr(WkT), ptr(qmask), ptr(lmask), ptr(Qb), ptr(Kb), ptr(P), ptr(baq), ptr(bqv), ptr(A),
             ptr(dfb), ptr(dfw), ptr(dfs), ptr(dhbar), ptr(dWq), ptr(dbq), ptr(dWk), ptr(dbk), wp, wn)
        return dfb, dfw, dfs, dhbar, dWq, dbq, dWk, dbk, None, None, None


class MomentUnitFn(Function):
    """MomentUnit.forward (reference models.py:288-303); Wcat = [conv_fb.W | conv_fc.W] (D, 2D)."""

    @staticmethod
    def forward(ctx, fcmean, fm, fb, Wcat, bcat, layout):
        """Returns (mu, view of fcmean): a later consumer of fcmean (the content stream's clip-mean chain) should read
        the view, so that its gradient is summed into dfcmean by the backward epilogue instead of by autograd."""
        fcmean, fm, fb, Wcat, bcat = map(_c, (fcmean, fm, fb, Wcat, bcat))
        N, D = fm.shape
        B, L, _ = fb.shape
        mu = torch.empty_like(fm)
        x1 = torch.empty_like(fm)                                     # f_b[i] * f_b[j], kept for the weight gradient
        call("smin_pair_product", stream(), ptr(fb), ptr(layout.cells), N, L, D, ptr(x1))
        call("smin_moment_unit_fwd", stream(), ptr(fcmean), ptr(fm), ptr(fb), ptr(layout.cells), N, B, L, D, ptr(Wcat), ptr(bcat), ptr(mu), ptr(x1))
        ctx.save_for_backward(fcmean, fb, Wcat, x1)
        ctx.layout = layout
        return mu, fcmean.view_as(fcmean)

    @staticmethod
    def backward(ctx, dmu, dacc):
        fcmean, fb, Wcat, x1 = ctx.saved_tensors
        layout = ctx.layout
        N, D = fcmean.shape
        B, L, _ = fb.shape
        dmu, dacc = _c(dmu), _c(dacc)
        if dmu is None:
            dmu = torch.zeros_like(fcmean)
        WcatT = Wcat.t().contiguous()
        dfcmean, dfb = torch.empty_like(fcmean), torch.empty_like(fb)
        dWcat, dbcat = torch.empty_like(Wcat), fb.new_empty((D,))
        _, wp, wn = _unit_ws(layout, 4, D, 4, 1, fb.device)
        call("smin_moment_unit_bwd", stream(), ptr(dmu), ptr(fcmean), ptr(fb), ptr(layout.cells), ptr(layout.row_ptr), ptr(layout.cellmap),
             N, B, L, D, ptr(WcatT), ptr(dfcmean), ptr(dfb), ptr(dWcat), ptr(dbcat), wp, wn, int(layout.all_valid), ptr(dacc), ptr(x1), None)
        return dfcmean, dmu, dfb, dWcat, dbcat, None


class ScoreMapFn(Function):
    """Localization.forward (reference models.py:335-344).  Returns pm (B,L,L) dense and psea (3,B,L)."""

    @staticmethod
    def forward(ctx, fm, fb, wm, bm, wb, bb, lmask, layout):
        fm, fb, wm, bm, wb, bb, lmask = map(_c, (fm, fb, wm, bm, wb, bb, lmask))
        N, D = fm.shape
        B, L, _ = fb.shape
        pm = fm.new_empty((B, L, L))
        psea = fm.new_empty((3, B, L))
        call("smin_score_map_fwd", stream(), ptr(fm), ptr(fb), ptr(layout.cells), N, B, L, D, ptr(wm), ptr(bm), ptr(wb), ptr(bb), ptr(lmask),
             ptr(pm), ptr(psea))
        ctx.save_for_backward(fm, fb, wm, wb, lmask, pm, psea)
        ctx.layout = layout
        return pm, psea

    @staticmethod
    def backward(ctx, dpm, dpsea):
        fm, fb, wm, wb, lmask, pm, psea = ctx.saved_tensors
        layout = ctx.layout
        N, D = fm.shape
        B, L, _ = fb.shape
        dpm = _c(dpm) if dpm is not None else torch.zeros_like(pm)
        dpsea = _c(dpsea) if dpsea is not None else torch.zeros_like(psea)
        dfm, dfb = torch.empty_like(fm), torch.empty_like(fb)
        dwm, dbm = torch.empty_like(wm), fm.new_empty((1,))
        dwb, dbb = torch.empty_like(wb), fm.new_empty((3,))
        _, wp, wn = _unit_ws(layout, 4, D, 4, 1, fm.device)
        call("smin_score_map_bwd", stream(), ptr(dpm), ptr(dpsea), ptr(pm), ptr(psea), ptr(fm), ptr(fb), ptr(layout.cells), N, B, L, D,
             ptr(wm), ptr(wb), ptr(lmask), ptr(dfm), ptr(dfb), ptr(dwm), ptr(dbm), ptr(dwb), ptr(dbb), wp, wn)
        return dfm, dfb, dwm, dbm, dwb, dbb, None, None


class LossFn(Function):
    """Restated loss of the reference's train loop (main.py:89-116) as one forward and one backward kernel."""

    @staticmethod
    def forward(ctx, pm, ps, pe, pa, ym, sm, mm, ys, ss, ye, se, ya, lm):
        pm, ps, pe, pa, sm, ss, se = (_c(x.float()) for x in (pm, ps, pe, pa, sm, ss, se))
        ym, mm, ys, ye, ya, lm = (_c(x if x.dtype in (torch.bool, torch.uint8) else x != 0) for x in (ym, mm, ys, ye, ya, lm))
        B, L = ps.shape
        loss = pm.new_empty((1,))
        part = pm.new_empty((B, 6))
        call("smin_loss_fwd", stream(), ptr(pm), ptr(ym), ptr(sm), ptr(mm), ptr(ps), ptr(ys), ptr(ss), ptr(pe), ptr(ye), ptr(se),
             ptr(pa), ptr(ya), ptr(lm), B, L, ptr(loss), ptr(part))
        ctx.save_for_backward(pm, ps, pe, pa, ym, sm, mm, ys, ss, ye, se, ya, lm, part)
        return loss.reshape(())

    @staticmethod
    def backward(ctx, dloss):
        pm, ps, pe, pa, ym, sm, mm, ys, ss, ye, se, ya, lm, part = ctx.saved_tensors
        B, L = ps.shape
        dloss = _c(dloss.reshape(1).float())
        dpm, dps, dpe, dpa = torch.empty_like(pm), torch.empty_like(ps), torch.empty_like(pe), torch.empty_like(pa)
        call("smin_loss_bwd", stream(), ptr(dloss), ptr(part), ptr(pm), ptr(ym), ptr(sm), ptr(mm), ptr(ps), ptr(ys), ptr(ss),
             ptr(pe), ptr(ye), ptr(se), ptr(pa), ptr(ya), ptr(lm), B, L, ptr(dpm), ptr(dps), ptr(dpe), ptr(dpa))
        return (dpm, dps, dpe, dpa) + (None,) * 9


def gemm_nt(a, b):
    """C = A @ B^T on the library's fp32 MFMA engine (tests / bench roofline probe)."""
    a, b = _c(a), _c(b)
    M, K = a.shape
    N = b.shape[0]
    c = a.new_empty((M, N))
    call("smin_gemm_nt", stream(), ptr(a), ptr(b), ptr(c), M, N, K)
    return c
