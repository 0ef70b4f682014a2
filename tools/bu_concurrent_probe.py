"""smin_boundary_unit_bwd repeated on one stream while another stream runs large contractions: are its outputs bit-identical
from repetition to repetition?  MODE=f32e L=64 BG=nt|tn|ntk16|mm|copy [ZERO_WQ=1]
Finding (round 2): with a bf16-core contraction kernel of this library as the neighbour (BG=nt / tn, MODE != f32) a handful of dfb
entries lose one addend (a 16-byte global load of the accumulation loop comes back as zeros); inputs stay intact, every
intermediate in the workspace is identical; never with MODE=f32, rocBLAS (mm), copies or a one-tile K loop (ntk16) as neighbour.
Cause: packed fp32 arithmetic (v_pk_fma_f32 ...) in the victim beside v_mfma_f32_32x32x16_bf16 of the neighbour; the library is
now built without packed fp32 arithmetic (csrc/Makefile NOPACK) and this probe reports 0 differing entries."""
import os, sys, ctypes, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import models
from vml_amd._lib import call, ptr
V = models.vml_amd
lib = V._lib.load()
dev = torch.device("cuda:0")
V.set_gemm_mode(os.environ.get("MODE", "f32e"))
g = torch.Generator().manual_seed(0)
arr = lambda ts: (ctypes.c_void_p * len(ts))(*[t.data_ptr() for t in ts])
side, main = torch.cuda.Stream(), torch.cuda.current_stream()
sp = lambda s: ctypes.c_void_p(s.cuda_stream)
B, L, Nq, D = int(os.environ.get("B", 2)), int(os.environ.get("L", 64)), 20, 512
lens = torch.tensor([L, max(3, L // 2)] + [L] * (B - 2))[:B]
lm = torch.arange(L).unsqueeze(0) < lens.unsqueeze(1)
mm = torch.triu(lm.unsqueeze(2) & lm.unsqueeze(1)).to(dev)
lay = V.CellLayout.begin(mm).finish()
N = lay.N
r = lambda *s: torch.randn(*s, generator=g).to(dev)
fb, fw, fs, hbar = r(B, L, D), r(B, Nq, D), r(B, D), r(N, D)
Wq, bq, Wk, bk = r(D, D) * 0.05, r(D), r(D, D) * 0.05, r(D)
qmask = torch.ones(B, Nq, device=dev); qmask[1, 15:] = 0
lmask = lm.float().to(dev)
out, Qb, baq, bqv, Kb = (torch.empty(B, L, D, device=dev) for _ in range(4)).__iter__().__next__(), None, None, None, None
out, Qb, baq, bqv = [torch.empty(B, L, D, device=dev) for _ in range(4)]
Kb, P, A = torch.empty(B, Nq, D, device=dev), torch.empty(B, L, Nq, device=dev), torch.empty(B, L, L, device=dev)
call("smin_boundary_unit_fwd", sp(main), ptr(fb), ptr(fw), ptr(fs), ptr(hbar), ptr(lay.cells), ptr(lay.row_ptr), N, B, L, Nq, D, ptr(Wq), ptr(bq), ptr(Wk), ptr(bk),
     ptr(qmask), ptr(lmask), ptr(out), ptr(Qb), ptr(Kb), ptr(P), ptr(baq), ptr(bqv), ptr(A))
WqT, WkT = Wq.t().contiguous(), Wk.t().contiguous()
if os.environ.get('ZERO_WQ'):
    WqT.zero_()
dout = r(B, L, D)
nbytes = 4 * (2 * B * L * L + 3 * B * L * D + B * L * Nq + B * Nq * D + 2 * 64 * (D * D + D)) + 4096
ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
Rb, Ob, Kb_ = 100000, 512, 1024
xb, Wb, yb = r(Rb, Kb_), r(Ob, Kb_), torch.empty(Rb, Ob, device=dev)
dWbg = torch.empty(Ob, Kb_, device=dev)
wsbg = torch.empty(lib.smin_linear_rows_bwd_workspace_bytes(Rb, Ob, Kb_) + 64, dtype=torch.uint8, device=dev)
xb16, Wb16 = r(Rb, 16), r(Ob, 16)
yb.normal_()
names = ["dfb", "dfw", "dfs", "dhbar", "dWq", "dbq", "dWk", "dbk"]
inputs = dict(dout=dout, fb=fb, fw=fw, fs=fs, hbar=hbar, WqT=WqT, WkT=WkT, qmask=qmask, lmask=lmask, Qb=Qb, Kb=Kb, P=P, baq=baq, bqv=bqv, A=A)
in_ref = {k: v.clone() for k, v in inputs.items()}
outs = []
torch.cuda.synchronize()
for rep in range(16):
    if rep % 2 == 1:
        bg = os.environ.get("BG", "nt")
        for _ in range(2):
            if bg == "nt":
                call("smin_linear_rows_fwd", sp(side), arr([xb]), 1, ptr(Wb), None, None, None, 1, Rb, Ob, Kb_, ptr(yb))
            elif bg == "tn":
                call("smin_linear_rows_bwd", sp(side), ptr(yb), arr([xb]), 1, None, Rb, Ob, Kb_, None, ptr(dWbg), None, ptr(wsbg), wsbg.numel())
            elif bg == "ntk16":
                call("smin_linear_rows_fwd", sp(side), arr([xb16]), 1, ptr(Wb16), None, None, None, 1, Rb, Ob, 16, ptr(yb))
            elif bg == "mm":
                with torch.cuda.stream(side):
                    torch.mm(xb, Wb.t(), out=yb)
            elif bg == "copy":
                with torch.cuda.stream(side):
                    yb.copy_(xb[:, :Ob]); yb.mul_(1.0001)
    o = [torch.empty(B, L, D, device=dev), torch.empty(B, Nq, D, device=dev), torch.empty(B, D, device=dev), torch.empty(N, D, device=dev),
         torch.empty(D, D, device=dev), torch.empty(D, device=dev), torch.empty(D, D, device=dev), torch.empty(D, device=dev)]
    call("smin_boundary_unit_bwd", sp(main), ptr(dout), ptr(fb), ptr(fw), ptr(fs), ptr(hbar), ptr(lay.cells), ptr(lay.row_ptr), N, B, L, Nq, D, ptr(WqT), ptr(WkT),
         ptr(qmask), ptr(lmask), ptr(Qb), ptr(Kb), ptr(P), ptr(baq), ptr(bqv), ptr(A), *[ptr(t) for t in o], ptr(ws), ws.numel())
    torch.cuda.synchronize()
    outs.append([t.clone() for t in o])
    for k, v in inputs.items():
        if not torch.equal(v, in_ref[k]):
            print("INPUT", k, "changed in rep", rep)
for i, n in enumerate(names):
    d = [float((outs[0][i] - o[i]).abs().max()) for o in outs[1:]]
    print(f"{n:6s} max diff over repetitions {max(d):.3e}  differing reps {[k + 1 for k, v in enumerate(d) if v > 0]}")
d = (outs[0][0] - outs[1][0]).abs()                      # dfb: quiet repetition against a concurrent one
idx = (d > 0).nonzero()
print("dfb differing entries:", idx.shape[0], "of", d.numel())
if idx.shape[0]:
    bs, rows, cols = idx[:, 0], idx[:, 1], idx[:, 2]
    print("  samples", sorted(set(bs.tolist())), "rows min/max", int(rows.min()), int(rows.max()), "distinct rows", len(set(rows.tolist())),
          "cols min/max", int(cols.min()), int(cols.max()), "distinct cols", len(set(cols.tolist())))
    print("  rows:", sorted(set((bs * L + rows).tolist()))[:40])
    ref = outs[0][0]
    print("  a few (value quiet, value concurrent):", [(float(ref[tuple(i)]), float(outs[1][0][tuple(i)])) for i in idx[:5]])
    d2 = (outs[1][0] - outs[3][0]).abs()
    print("  concurrent rep 1 vs concurrent rep 3 differ in", int((d2 > 0).sum()), "entries")
# the same call in exact mode, alone: which of the two is right?
V.set_gemm_mode("f32")
o = [torch.empty(B, L, D, device=dev), torch.empty(B, Nq, D, device=dev), torch.empty(B, D, device=dev), torch.empty(N, D, device=dev),
     torch.empty(D, D, device=dev), torch.empty(D, device=dev), torch.empty(D, D, device=dev), torch.empty(D, device=dev)]
call("smin_boundary_unit_bwd", sp(main), ptr(dout), ptr(fb), ptr(fw), ptr(fs), ptr(hbar), ptr(lay.cells), ptr(lay.row_ptr), N, B, L, Nq, D, ptr(WqT), ptr(WkT),
     ptr(qmask), ptr(lmask), ptr(Qb), ptr(Kb), ptr(P), ptr(baq), ptr(bqv), ptr(A), *[ptr(t) for t in o], ptr(ws), ws.numel())
torch.cuda.synchronize()
ex = o[0]
print("exact-mode dfb vs f32e quiet: max diff", float((ex - outs[0][0]).abs().max()), " vs f32e concurrent:", float((ex - outs[1][0]).abs().max()))
if idx.shape[0]:
    print("  exact values at the differing entries:", [float(ex[tuple(i)]) for i in idx[:5]])
# which intermediate of the backward differs between a quiet and a concurrent repetition?  (workspace layout of smin_boundary_unit_bwd)
V.set_gemm_mode(os.environ.get("MODE", "f32e"))
al = lambda n: (n + 3) & ~3
sizes = [("dAbm", B * L * L), ("draw", B * L * L), ("dbaq_lm", B * L * D), ("dfs_part", B * L * D), ("dQK", B * L * Nq), ("dQb", B * L * D), ("dKb", B * Nq * D)]
snaps = []
for rep in range(6):
    if rep % 2 == 1:
        for _ in range(2):
            call("smin_linear_rows_fwd", sp(side), arr([xb]), 1, ptr(Wb), None, None, None, 1, Rb, Ob, Kb_, ptr(yb))
    ws.zero_()
    torch.cuda.synchronize() if rep % 2 == 0 else None
    o = [torch.empty(B, L, D, device=dev), torch.empty(B, Nq, D, device=dev), torch.empty(B, D, device=dev), torch.empty(N, D, device=dev),
         torch.empty(D, D, device=dev), torch.empty(D, device=dev), torch.empty(D, D, device=dev), torch.empty(D, device=dev)]
    call("smin_boundary_unit_bwd", sp(main), ptr(dout), ptr(fb), ptr(fw), ptr(fs), ptr(hbar), ptr(lay.cells), ptr(lay.row_ptr), N, B, L, Nq, D, ptr(WqT), ptr(WkT),
         ptr(qmask), ptr(lmask), ptr(Qb), ptr(Kb), ptr(P), ptr(baq), ptr(bqv), ptr(A), *[ptr(t) for t in o], ptr(ws), ws.numel())
    torch.cuda.synchronize()
    wf = ws.view(torch.float32)
    off, parts = 0, {}
    for nme, cnt in sizes:
        parts[nme] = wf[off:off + cnt].clone(); off += al(cnt)
    parts["dfb"] = o[0].clone()
    snaps.append(parts)
for nme in [n for n, _ in sizes] + ["dfb"]:
    print(f"  {nme:9s} quiet0 vs concurrent1: {int((snaps[0][nme] != snaps[1][nme]).sum()):7d} entries differ;  quiet0 vs quiet2: {int((snaps[0][nme] != snaps[2][nme]).sum())}")
# which term of dfb[r] = dout[r] + sum_j A[j][i] lm[j] dout[j] + dq * tq is missing in the concurrent result?
q, c = snaps[0]["dfb"], snaps[1]["dfb"]
idx = (q != c).nonzero()
for e in idx[:6]:
    b, i, d = [int(v) for v in e]
    miss = float(q[b, i, d] - c[b, i, d])
    contrib = (A[b, :, i] * lmask[b, :] * dout[b, :, d])
    j = int((contrib - miss).abs().argmin())
    print(f"  (b={b}, i={i}, d={d}): quiet {float(q[b,i,d]):+.5f} concurrent {float(c[b,i,d]):+.5f} missing {miss:+.5f}; closest single-j contribution: j={j} -> {float(contrib[j]):+.5f}"
          f" (A[j][i]={float(A[b,j,i]):.4f});  sum of all contributions {float(contrib.sum()):+.5f}")
