import ctypes, os, sys, torch
sys.path.insert(0, os.environ.get("SMIN_ROOT") or os.getcwd())
import models
lib = models.vml_amd._lib.load()
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
nb = lib.smin_bilstm_layer_bwd_workspace_bytes(B, Nq, In, H)
ws = torch.empty(nb + 64, dtype=torch.uint8, device=dev)
vp = lambda t: ctypes.c_void_p(t.data_ptr() if t is not None else 0)
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
def f(): lib.smin_bilstm_layer_fwd(st, vp(x), vp(Wih), vp(bias), vp(W4), vp(length), B, Nq, In, H, vp(G), vp(Ho), vp(Cs))
def b(): lib.smin_bilstm_layer_bwd(st, vp(dHo), vp(x), vp(Ho), vp(G), vp(Cs), vp(WihT), vp(Whh), vp(length), B, Nq, In, H, vp(dX), None, None, None, vp(ws), ctypes.c_size_t(nb + 64))
for name, fn in (("fwd layer (input GEMM + recurrence)", f), ("bwd layer inputs half", b)):
    for _ in range(10): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(50): fn()
    e.record(); torch.cuda.synchronize()
    print(name, round(s.elapsed_time(e) / 50 * 1e3, 1), "us")
