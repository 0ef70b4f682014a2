#!/usr/bin/env python3
"""Reads the phase stamps of a -DSMIN_ATTN_STAMPS build of content_attn_bwd_kernel (see tools/attn_stamps.sh) on the bench
workload's cell list and prints, per wave of two workgroups, the cycles between consecutive stamps (median over rounds)."""
import ctypes
import os
import statistics
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
NAMES = ["S+softmax", "wait B4", "clipattn(a)", "dA loop", "sym+dq loop+dP", "dS + tiles", "dchat loop", "prefetch chat", "wait B1", "R1",
         "wait B2", "W+prefetch g", "wait B3", "R0"]


def main():
    import bench
    import models
    from vml_amd.cells import CellLayout
    lib = ctypes.CDLL(sys.argv[1])
    dev = torch.device("cuda:0")
    B, T, L, Nq, dl, C = 64, 256, 64, 20, 128, 4
    b = bench.make_batch(B, T, L, Nq, 500, seed=1000, device=dev)
    lay = CellLayout.from_mask(b["moment_mask"])
    N = lay.N
    g = torch.Generator(device=dev).manual_seed(1)
    r = lambda *s: torch.randn(*s, generator=g, device=dev)
    chat, dcc, dccm = r(N * C, dl) * 0.5, r(N * C, dl), r(N, dl)
    Mq, uq, what, shat = r(B, Nq, dl) * 0.2, r(B, Nq) * 0.2, r(B, Nq, dl), r(B, dl)
    qmask = b["query_mask"].reshape(B, -1).float().contiguous()
    dchat = torch.empty_like(chat)
    dMq, duq, dwhat, dshat = torch.empty_like(Mq), torch.empty_like(uq), torch.empty_like(what), torch.empty_like(shat)
    lib.smin_content_attn_bwd_workspace_bytes.restype = ctypes.c_size_t
    wsb = lib.smin_content_attn_bwd_workspace_bytes(N, B, C, dl)
    ws = torch.empty(wsb + 4096, dtype=torch.uint8, device=dev)
    vp = lambda t: ctypes.c_void_p(t.data_ptr() if t is not None else 0)
    for variant, (g1, g2) in {"rows+mean": (dcc, dccm), "mean only": (None, dccm)}.items():
        for it in range(5):
            rc = lib.smin_content_attn_bwd(ctypes.c_void_p(torch.cuda.current_stream().cuda_stream), vp(g1), vp(g2), vp(chat), vp(lay.cells), vp(lay.row_ptr),
                                           N, B, L, C, dl, Nq, vp(Mq), vp(uq), vp(what), vp(shat), vp(qmask), vp(dchat), vp(dMq), vp(duq), vp(dwhat), vp(dshat),
                                           vp(ws), ctypes.c_size_t(wsb))
            assert rc == 0, rc
        torch.cuda.synchronize()
        n = 2 * 4 * 8 * 16
        buf = (ctypes.c_ulonglong * n)()
        assert lib.smin_debug_attn_stamps(buf, n) == 0
        print(f"== {variant}: cycles between stamps, median over the recorded rounds; columns = (workgroup, wave)")
        cols = []
        for blk in range(2):
            for w in range(4):
                per = [[buf[((blk * 4 + w) * 8 + rd) * 16 + k] for k in range(15)] for rd in range(8)]
                per = [p for p in per if all(x > 0 for x in p)]
                d = [[p[k + 1] - p[k] for k in range(14)] for p in per]
                rounds = [per[i + 1][0] - per[i][0] for i in range(len(per) - 1)]
                cols.append(([statistics.median(x[k] for x in d) for k in range(14)] if d else [0] * 14, statistics.median(rounds) if rounds else 0, len(per)))
        for k, nm in enumerate(NAMES):
            print(f"{nm:18s}" + "".join(f"{c[0][k]:9.0f}" for c in cols))
        print(f"{'round (start-start)':18s}" + "".join(f"{c[1]:9.0f}" for c in cols))
        print(f"{'rounds recorded':18s}" + "".join(f"{c[2]:9d}" for c in cols))


if __name__ == "__main__":
    main()
