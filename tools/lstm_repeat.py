#!/usr/bin/env python3
"""Cluster LSTM: the same layer forward + backward many times on the same inputs; every output must repeat bit for bit (a granule read
too early would show up as a rare difference).  python tools/lstm_repeat.py [reps]"""
import ctypes, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import models
lib = models.vml_amd._lib.load()
dev = torch.device("cuda:0")
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
g = torch.Generator(device=dev).manual_seed(5)
r = lambda *s: torch.randn(*s, generator=g, device=dev)
vp = lambda t: ctypes.c_void_p(t.data_ptr() if t is not None else 0)
for (B, Nq, In, H) in [(64, 20, 512, 256), (5, 9, 40, 64), (3, 20, 300, 256), (3, 9, 40, 64), (2, 9, 40, 64), (37, 5, 300, 256), (6, 9, 24, 128), (2, 14, 300, 256)]:
    x, Wih, bias, Whh = r(B, Nq, In), r(8 * H, In) * 0.05, r(8 * H) * 0.1, r(2, 4 * H, H) * 0.1
    W4 = Whh.view(2, 4, H, H).permute(0, 3, 2, 1).contiguous()
    length = torch.randint(1, Nq + 1, (B,), generator=torch.Generator().manual_seed(B)).to(torch.int32).to(dev)
    dHo = r(B, Nq, 2 * H)
    WihT = Wih.t().contiguous()
    nb = lib.smin_bilstm_layer_bwd_workspace_bytes(B, Nq, In, H)
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    ref, bad = None, 0
    for it in range(reps):
        G, Ho, Cs = torch.zeros(B, Nq, 2, 4 * H, device=dev), torch.zeros(B, Nq, 2 * H, device=dev), torch.zeros(B, Nq, 2, H, device=dev)
        dX, ws = torch.zeros(B, Nq, In, device=dev), torch.zeros(nb + 64, dtype=torch.uint8, device=dev)
        assert lib.smin_bilstm_layer_fwd(st, vp(x), vp(Wih), vp(bias), vp(W4), vp(length), B, Nq, In, H, vp(G), vp(Ho), vp(Cs)) == 0
        assert lib.smin_bilstm_layer_bwd(st, vp(dHo), vp(x), vp(Ho), vp(G), vp(Cs), vp(WihT), vp(Whh), vp(length), B, Nq, In, H, vp(dX), None, None, None, vp(ws),
                                         ctypes.c_size_t(nb + 64)) == 0
        dG = ws[: B * Nq * 8 * H * 4].view(torch.float32).clone()
        cur = (Ho.clone(), Cs.clone(), dG, dX.clone())
        if ref is None:
            ref = cur
        elif not all(torch.equal(a, b) for a, b in zip(ref, cur)):
            bad += 1
            if bad <= 2:
                for nm, a, b in zip(("Hout", "Cs", "dG", "dX"), ref, cur):
                    if not torch.equal(a, b):
                        d = (a - b).abs()
                        print("   differs:", nm, "entries", int((d > 0).sum()), "max", float(d.max()), "first index", int(torch.nonzero(d.flatten() > 0)[0]))
    print(f"B={B} Nq={Nq} In={In} H={H}: {bad} of {reps - 1} repetitions differ; poll expired: {lib.smin_lstm_cluster_error()}")
