#!/usr/bin/env python3
"""Reads the per-step phase stamps of a -DSMIN_LSTM_STAMPS build of the cluster LSTM kernels (tools/lstm_stamps.sh): one layer forward
and backward at the bench workload's query shape (B = 64, Nq = 20, H = 256, In = 512), s_memtime ticks between stamps (about one per shader clock: a forward
step of 4.4 us reads ~7 700) as medians over the time steps."""
import ctypes
import statistics
import sys

import torch

FWD = ["contraction", "barrier", "gates + publish", "gather h", "barrier"]
BWD = ["gate gradients", "barrier", "contraction + publish", "barrier", "wait partial sums"]


def main():
    lib = ctypes.CDLL(sys.argv[1])
    dev = torch.device("cuda:0")
    B, Nq, In, H = 64, 20, 512, 256
    g = torch.Generator(device=dev).manual_seed(3)
    r = lambda *s: torch.randn(*s, generator=g, device=dev)
    x, Wih, bias, Whh = r(B, Nq, In), r(8 * H, In) * 0.05, r(8 * H) * 0.1, r(2, 4 * H, H) * 0.05
    W4 = Whh.view(2, 4, H, H).permute(0, 3, 2, 1).contiguous()
    length = torch.full((B,), Nq, dtype=torch.int32, device=dev)
    G, Ho, Cs = torch.empty(B, Nq, 2, 4 * H, device=dev), torch.empty(B, Nq, 2 * H, device=dev), torch.empty(B, Nq, 2, H, device=dev)
    dHo, dX = r(B, Nq, 2 * H), torch.empty(B, Nq, In, device=dev)
    WihT = Wih.t().contiguous()
    lib.smin_bilstm_layer_bwd_workspace_bytes.restype = ctypes.c_size_t
    nb = lib.smin_bilstm_layer_bwd_workspace_bytes(B, Nq, In, H)
    ws = torch.empty(nb + 64, dtype=torch.uint8, device=dev)
    vp = lambda t: ctypes.c_void_p(t.data_ptr() if t is not None else 0)
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    for _ in range(3):
        assert lib.smin_bilstm_layer_fwd(st, vp(x), vp(Wih), vp(bias), vp(W4), vp(length), B, Nq, In, H, vp(G), vp(Ho), vp(Cs)) == 0
        assert lib.smin_bilstm_layer_bwd(st, vp(dHo), vp(x), vp(Ho), vp(G), vp(Cs), vp(WihT), vp(Whh), vp(length), B, Nq, In, H, vp(dX), None, None, None,
                                         vp(ws), ctypes.c_size_t(nb + 64)) == 0
    torch.cuda.synchronize()
    n = 2 * 64 * 8
    buf = (ctypes.c_ulonglong * n)()
    assert lib.smin_debug_lstm_stamps(buf, n) == 0
    for kern, names in ((0, FWD), (1, BWD)):
        steps = [[buf[(kern * 64 + s) * 8 + k] for k in range(6)] for s in range(Nq)]
        steps = [p for p in steps[2:Nq - 1] if all(v > 0 for v in p)]
        print("== %s: median s_memtime ticks per phase over %d steps" % ("forward" if kern == 0 else "backward", len(steps)))
        for k, nm in enumerate(names):
            print("  %-24s %6.0f" % (nm, statistics.median(p[k + 1] - p[k] for p in steps)))
        per = [steps[i + 1][0] - steps[i][0] for i in range(len(steps) - 1)]
        print("  %-24s %6.0f" % ("step (start to start)", statistics.median(per)))


main()
