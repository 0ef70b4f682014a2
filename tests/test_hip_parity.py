"""GPU parity tests (run with -m gpu on an MI355X): the HIP path, called through the C ABI via the
nn.Module mirror, against (a) the golden fixtures produced from the reference and (b) the CPU oracle on
the same seeded inputs.  fp32 tolerance demanded by BASELINE.json's north_star: 1e-3 on scores; the
tests hold the path to 2e-5 (scores) and ~1e-3 relative (gradients)."""
import numpy as np
import pytest
import torch

from tests import helpers as H

pytestmark = pytest.mark.gpu

SCORE_TOL = 2e-5


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    import models
    models.vml_amd._lib.load()          # fail loudly if libsmin_hip.so is missing
    return torch.device("cuda:0")


def build_model(cfg, sd, dev):
    import models
    m = models.SMIN(cfg["T"], cfg["L"], cfg["C"], cfg["D"], cfg["dl"], cfg["layers"], cfg["Din"], cfg["Nq"], cfg["H"], dev)
    missing = m.load_state_dict(sd, strict=True)
    assert not missing.missing_keys and not missing.unexpected_keys
    return m.to(dev)


def rel_err(got, ref):
    return (got - ref).abs().max().item() / max(ref.abs().max().item(), 1e-12)


# ---------------------------------------------------------------- engine
@pytest.mark.parametrize("M,N,K", [(128, 128, 32), (1000, 128, 512), (515, 512, 128), (300, 16, 32), (4096, 512, 1024), (37, 20, 16)])
def test_gemm_nt_engine(dev, M, N, K):
    from vml_amd.functional import gemm_nt
    g = torch.Generator().manual_seed(M + N + K)
    a = torch.randn(M, K, generator=g)
    b = torch.randn(N, K, generator=g)
    ref = (a.double() @ b.double().t())
    got = gemm_nt(a.to(dev), b.to(dev)).cpu().double()
    assert (got - ref).abs().max().item() <= 2e-6 * K ** 0.5 * 8


def test_gemm_nt_identity_asymmetric(dev):
    """A = I against an asymmetric B catches a transposed C write (guide: always A=I-check with asymmetric B)."""
    from vml_amd.functional import gemm_nt
    K = 128
    a = torch.eye(K)
    b = torch.arange(96 * K, dtype=torch.float32).reshape(96, K) * 0.25 - 1000.0
    got = gemm_nt(a.to(dev), b.to(dev)).cpu()
    assert torch.equal(got, b.t().contiguous())


# ---------------------------------------------------------------- module seams vs golden seam tensors
@pytest.mark.parametrize("name", H.TINY)
def test_seams_against_golden(dev, name):
    import models
    cfg, sd, batch, out, _, _ = H.split_tiny(H.load_npz(name))
    m = build_model(cfg, sd, dev)
    g = {k: v.to(dev) for k, v in out.items()}
    b = {k: v.to(dev) for k, v in batch.items()}
    with torch.no_grad():
        f, fs, fw = m.backbone(b["video_features"], b["video_mask"], b["query_features"], b["query_mask"])
        for k, v in (("f", f), ("fs", fs), ("fw", fw)):
            assert rel_err(v, g[k]) < 1e-5, k
        fc, fm, fb = m.pgm(g["f"], b["moment_mask"])
        assert rel_err(fc, g["fc"]) < 2e-6 and rel_err(fm, g["fm"]) < 2e-6 and rel_err(fb, g["fb"]) < 2e-6
        prev = ("fc", "fm", "fb")
        for k in range(cfg["layers"]):
            smi = m.smis[k]
            cu = smi.content_unit(g[prev[0]], g["fw"], g["fs"], g[prev[1]], b["query_mask"], b["moment_mask"])
            bu = smi.boundary_unit(g[prev[2]], g["fw"], g["fs"], g[prev[1]], b["query_mask"], b["length_mask"])
            mu = smi.moment_unit(g[f"cu{k}"], g[prev[1]], g[f"bu{k}"], b["moment_mask"])
            assert rel_err(cu, g[f"cu{k}"]) < 1e-5, f"cu{k}"
            assert rel_err(bu, g[f"bu{k}"]) < 1e-5, f"bu{k}"
            assert rel_err(mu, g[f"mu{k}"]) < 1e-5, f"mu{k}"
            cu2, mu2, bu2 = smi(g[prev[0]], g[prev[1]], g[prev[2]], g["fw"], g["fs"], b["query_mask"], b["length_mask"], b["moment_mask"])
            assert rel_err(cu2, g[f"cu{k}"]) < 1e-5 and rel_err(mu2, g[f"mu{k}"]) < 1e-5 and rel_err(bu2, g[f"bu{k}"]) < 1e-5
            prev = (f"cu{k}", f"mu{k}", f"bu{k}")
        pm, ps, pe, pa = m.localization(g[prev[1]], g[prev[2]], b["length_mask"], b["moment_mask"])
        for k, v in (("pm", pm), ("ps", ps), ("pe", pe), ("pa", pa)):
            assert (v - g[k]).abs().max().item() < SCORE_TOL, k


@pytest.mark.parametrize("name", H.TINY)
def test_forward_backward_against_golden(dev, name):
    from vml_amd import loss_fn
    cfg, sd, batch, out, grads, loss_ref = H.split_tiny(H.load_npz(name))
    m = build_model(cfg, sd, dev)
    b = {k: v.to(dev) for k, v in batch.items()}
    pm, ps, pe, pa = m(*H.model_inputs(b))
    assert pm.is_contiguous() and pm.shape == out["pm"].shape
    for k, v in (("pm", pm), ("ps", ps), ("pe", pe), ("pa", pa)):
        assert (v.detach().cpu() - out[k]).abs().max().item() < SCORE_TOL, k
    assert pm.detach()[~b["moment_mask"]].abs().max().item() == 0.0
    loss = loss_fn(pm, b["ym"], b["sm"], b["moment_mask"], ps, b["ys"], b["ss"], pe, b["ye"], b["se"], pa, b["ya"], b["length_mask"])
    assert abs(loss.item() - loss_ref) < 2e-5
    loss.backward()
    worst = 0.0
    for k, p in m.named_parameters():
        ref = grads[k]
        err = (p.grad.cpu() - ref).abs().max().item() / max(ref.abs().max().item(), 1e-6)
        worst = max(worst, err)
        assert err < 2e-3, (k, err)
    print(name, "worst relative grad error", worst)


def test_float_masks_take_the_torch_prologue(dev):
    """Masks that are not one byte per element (fp32 0/1 here) leave the one-launch prologue (smin_step_prologue) for torch calls:
    scores and gradients equal the byte-mask run bit for bit."""
    from oracle import smin_oracle as O
    from vml_amd import loss_fn
    T, L, C, D, dl, layers, Din, Nq, Hh, B = 64, 16, 4, 128, 32, 3, 40, 9, 64, 4
    sd = O.formula_state_dict(H.smin_shapes(T, L, C, D, dl, layers, Din, Nq, Hh), gain=1.2)
    b = {k: v.to(dev) for k, v in O.synthetic_batch(B, T, L, Nq, Din, seed=5).items()}
    res = []
    for as_float in (False, True):
        m = build_model(dict(T=T, L=L, C=C, D=D, dl=dl, layers=layers, Din=Din, Nq=Nq, H=Hh), sd, dev)
        inp = list(H.model_inputs(b))
        if as_float:
            inp = [t.float() if i in (1, 3, 4, 5) else t for i, t in enumerate(inp)]          # video / query / length / moment masks
        out = m(*inp)
        loss_fn(out[0], b["ym"], b["sm"], b["moment_mask"], out[1], b["ys"], b["ss"], out[2], b["ye"], b["se"], out[3], b["ya"], b["length_mask"]).backward()
        res.append(([o.detach().clone() for o in out], {k: p.grad.clone() for k, p in m.named_parameters()}))
    for a, c in zip(res[0][0], res[1][0]):
        assert torch.equal(a, c)
    for k in res[0][1]:
        assert torch.equal(res[0][1][k], res[1][1][k]), k


def test_fused_step_repeats_bit_for_bit(dev):
    """The one-node step, thirty times on the same weights and batch with the weight stream on and off in turn: every gradient repeats
    bit for bit.  (What this caught: an asm store of a 16-byte granule without the wait state of the store-data hazard -- a wrong value
    under a right tag in 3 % of the runs, and only beside the other streams' kernels.)"""
    from oracle import smin_oracle as O
    from vml_amd import loss_fn
    T, L, C, D, dl, layers, Din, Nq, Hh, B = 128, 64, 4, 512, 128, 3, 500, 20, 256, 3
    sd = O.formula_state_dict(H.smin_shapes(T, L, C, D, dl, layers, Din, Nq, Hh), gain=1.2)
    b = {k: v.to(dev) for k, v in O.synthetic_batch(B, T, L, Nq, Din, seed=21).items()}
    ref = None
    for it in range(30):
        m = build_model(dict(T=T, L=L, C=C, D=D, dl=dl, layers=layers, Din=Din, Nq=Nq, H=Hh), sd, dev)
        m.async_weights = it % 2 == 0
        out = m(*H.model_inputs(b))
        loss_fn(out[0], b["ym"], b["sm"], b["moment_mask"], out[1], b["ys"], b["ss"], out[2], b["ye"], b["se"], out[3], b["ya"], b["length_mask"]).backward()
        cur = {k: p.grad.clone() for k, p in m.named_parameters()}
        if ref is None:
            ref = cur
        for k in ref:
            assert torch.equal(ref[k], cur[k]), (it, k)


# ---------------------------------------------------------------- full BASELINE shapes vs golden (formula weights)
@pytest.mark.parametrize("name", ["tacos_yml", "tacos_d500", "charades", "anet_yml", "anet_t256"])
def test_full_size_against_golden(dev, name):
    from oracle import smin_oracle as O           # test infrastructure: input / weight generators only
    from vml_amd import loss_fn
    z = H.load_npz("g5_" + name)
    T, L, C, D, dl, layers, Din, Nq, Hh = H.FULL[name]
    B, seed = int(z["cfg"][-2]), int(z["cfg"][-1])
    sd = O.formula_state_dict(H.smin_shapes(T, L, C, D, dl, layers, Din, Nq, Hh), gain=1.3)
    m = build_model(dict(T=T, L=L, C=C, D=D, dl=dl, layers=layers, Din=Din, Nq=Nq, H=Hh), sd, dev)
    batch = O.synthetic_batch(B, T, L, Nq, Din, seed=seed)
    b = {k: v.to(dev) for k, v in batch.items()}
    pm, ps, pe, pa = m(*H.model_inputs(b))
    for k, v in (("pm", pm), ("ps", ps), ("pe", pe), ("pa", pa)):
        err = (v.detach().cpu() - torch.from_numpy(z["out/" + k])).abs().max().item()
        print(name, k, "max abs err", err)
        assert err < 1e-4, (k, err)               # north_star demands 1e-3
    loss = loss_fn(pm, b["ym"], b["sm"], b["moment_mask"], ps, b["ys"], b["ss"], pe, b["ye"], b["se"], pa, b["ya"], b["length_mask"])
    assert abs(loss.item() - float(z["loss"])) < 1e-4
    loss.backward()
    norms = dict(zip([str(s) for s in z["grad_names"]], z["grad_norms"]))
    for k, p in m.named_parameters():
        got = p.grad.double().norm().item()
        assert abs(got - norms[k]) <= 2e-3 * norms[k] + 1e-7, (k, got, norms[k])
    for k in [f for f in z.files if f.startswith("grad/")]:
        ref = torch.from_numpy(z[k])
        got = dict(m.named_parameters())[k[5:]].grad.cpu()
        assert (got - ref).abs().max().item() <= 2e-3 * ref.abs().max().item() + 1e-7, k


# ---------------------------------------------------------------- oracle on fresh seeded inputs
@pytest.mark.parametrize("T,L,C,D,dl,layers,Din,Nq,Hh,B", [
    (64, 16, 4, 64, 32, 2, 40, 9, 32, 5),
    (48, 24, 4, 128, 64, 1, 32, 20, 64, 3),      # r = 2 < C: empty clips and dropped frames
    (64, 32, 2, 64, 16, 2, 16, 3, 32, 2),
    (96, 32, 3, 128, 128, 3, 24, 32, 64, 1),     # B = 1 (the reference raises here), Nq = 32, dl = 128
    (16, 16, 4, 64, 32, 2, 24, 6, 32, 3),        # r = 1: one frame per snippet (general proposal-map backward path)
    (32, 8, 4, 64, 16, 5, 24, 6, 32, 2),         # 5 layers: the content stream's history spills into a second 4-wide partition
    (32, 8, 4, 64, 16, 9, 24, 6, 32, 2),         # 9 layers: beyond the clip-window-means launch limit -> unit as written
])
def test_against_oracle_random(dev, T, L, C, D, dl, layers, Din, Nq, Hh, B):
    from oracle import smin_oracle as O
    from vml_amd import loss_fn
    sd = O.formula_state_dict(H.smin_shapes(T, L, C, D, dl, layers, Din, Nq, Hh), gain=1.2)
    batch = O.synthetic_batch(B, T, L, Nq, Din, seed=T + L + D)
    sdg = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    pm0, ps0, pe0, pa0 = O.smin_forward(sdg, dict(T=T, L=L, C=C), *H.model_inputs(batch))
    l0 = O.loss_fn(pm0, batch["ym"], batch["sm"], batch["moment_mask"], ps0, batch["ys"], batch["ss"], pe0, batch["ye"], batch["se"],
                   pa0, batch["ya"], batch["length_mask"])
    l0.backward()
    m = build_model(dict(T=T, L=L, C=C, D=D, dl=dl, layers=layers, Din=Din, Nq=Nq, H=Hh), sd, dev)
    b = {k: v.to(dev) for k, v in batch.items()}
    pm, ps, pe, pa = m(*H.model_inputs(b))
    for got, ref in ((pm, pm0), (ps, ps0), (pe, pe0), (pa, pa0)):
        assert (got.detach().cpu() - ref.detach()).abs().max().item() < SCORE_TOL
    loss = loss_fn(pm, b["ym"], b["sm"], b["moment_mask"], ps, b["ys"], b["ss"], pe, b["ye"], b["se"], pa, b["ya"], b["length_mask"])
    loss.backward()
    for k, p in m.named_parameters():
        ref = sdg[k].grad
        assert (p.grad.cpu() - ref).abs().max().item() <= 2e-3 * ref.abs().max().item() + 1e-7, k


def test_dense_seams_arbitrary_inputs(dev):
    """Sub-module seams must reproduce the reference for inputs whose masked cells are NOT zero, and for a
    non-triangular moment mask (SURVEY 8a-0 caveat): work is driven by moment_mask, not by j >= i."""
    import models
    from oracle import smin_oracle as O
    B, L, C, D, dl, Nq = 2, 8, 4, 32, 16, 6
    g = torch.Generator().manual_seed(5)
    smi = models.SMI(D, dl)
    sd = {"smis.0." + k: v.detach().clone() for k, v in smi.state_dict().items()}
    f_c, f_m, f_b = torch.randn(B, L, L, C, D, generator=g), torch.randn(B, L, L, D, generator=g), torch.randn(B, L, D, generator=g)
    f_w, f_s = torch.randn(B, Nq, D, generator=g), torch.randn(B, D, generator=g)
    qmask = torch.ones(B, Nq, 1, dtype=torch.uint8); qmask[1, 4:] = 0
    lmask = torch.ones(B, L, dtype=torch.bool); lmask[1, 6:] = False
    mmask = torch.rand(B, L, L, generator=g) > 0.4
    cu0, mu0, bu0 = O.smi_layer(sd, 0, f_c, f_m, f_b, f_w, f_s, qmask, lmask, mmask)
    smi = smi.to(dev)
    with torch.no_grad():
        cu, mu, bu = smi(*(t.to(dev) for t in (f_c, f_m, f_b, f_w, f_s, qmask, lmask, mmask)))
    assert rel_err(cu.cpu(), cu0) < 1e-5 and rel_err(mu.cpu(), mu0) < 1e-5 and rel_err(bu.cpu(), bu0) < 1e-5


def test_smin_all_ones_moment_mask(dev):
    """SURVEY 8a-0 caveat at the SMIN level: a caller that passes a non-triangular moment_mask (all ones; random) gets what the
    reference computes -- lower-triangle cells have empty clips but biases make them non-zero.  Work follows the mask."""
    from oracle import smin_oracle as O
    from vml_amd import loss_fn
    T, L, C, D, dl, layers, Din, Nq, Hh, B = 32, 8, 4, 64, 16, 2, 24, 6, 32, 3
    sd = O.formula_state_dict(H.smin_shapes(T, L, C, D, dl, layers, Din, Nq, Hh), gain=1.2)
    g = torch.Generator().manual_seed(8)
    for kind in ("ones", "random"):
        batch = O.synthetic_batch(B, T, L, Nq, Din, seed=12)
        batch["moment_mask"] = torch.ones(B, L, L, dtype=torch.bool) if kind == "ones" else torch.rand(B, L, L, generator=g) > 0.4
        batch["sm"] = batch["sm"] * 0 + torch.rand(B, L, L, generator=g) * batch["moment_mask"]
        batch["ym"] = batch["sm"] > 0.5
        sdg = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
        ref = O.smin_forward(sdg, dict(T=T, L=L, C=C), *H.model_inputs(batch))
        O.loss_fn(ref[0], batch["ym"], batch["sm"], batch["moment_mask"], ref[1], batch["ys"], batch["ss"], ref[2], batch["ye"],
                  batch["se"], ref[3], batch["ya"], batch["length_mask"]).backward()
        assert ref[0].detach()[:, 5, 2].abs().max().item() > 0 or kind == "random"      # lower triangle is live in the reference
        m = build_model(dict(T=T, L=L, C=C, D=D, dl=dl, layers=layers, Din=Din, Nq=Nq, H=Hh), sd, dev)
        b = {k: v.to(dev) for k, v in batch.items()}
        out = m(*H.model_inputs(b))
        for got, want in zip(out, ref):
            assert (got.detach().cpu() - want.detach()).abs().max().item() < SCORE_TOL, kind
        loss_fn(out[0], b["ym"], b["sm"], b["moment_mask"], out[1], b["ys"], b["ss"], out[2], b["ye"], b["se"], out[3], b["ya"],
                b["length_mask"]).backward()
        for k, p in m.named_parameters():
            g0 = sdg[k].grad
            assert (p.grad.cpu() - g0).abs().max().item() <= 2e-3 * g0.abs().max().item() + 1e-7, (kind, k)


def test_batch_targets_on_device(dev):
    """build_targets on the HIP device against the per-sample restatement of dataset.py:95-155.
    PARITY UNPINNED (dataset.py is not importable: torchtext / h5py absent; the reference holds no fixtures for it)."""
    import models
    from oracle import labels_oracle as LO
    g = torch.Generator().manual_seed(3)
    for (T, L, B) in [(256, 64, 7), (1024, 512, 3), (64, 16, 6)]:
        dur = torch.rand(B, generator=g) * 200 + 5
        ts = torch.rand(B, generator=g) * dur * 0.6
        te = ts + torch.rand(B, generator=g) * (dur - ts) * 0.9 + 0.5
        nf = torch.randint(3, 2 * T, (B,), generator=g)
        out = models.vml_amd.build_targets(torch.stack([ts, te], 1).to(dev), dur.to(dev), nf.to(dev), T, L, device=dev)
        for b in range(B):
            ref = LO.sample_targets(float(ts[b]), float(te[b]), float(dur[b]), int(nf[b]), T, L)
            for k, v in ref.items():
                a = out[k][b].cpu()
                if v.dtype.is_floating_point:
                    assert torch.allclose(a, v, rtol=1e-5, atol=1e-6, equal_nan=True), (T, L, b, k)
                else:
                    assert torch.equal(a.to(v.dtype).reshape(v.shape), v), (T, L, b, k)


def test_native_host_equals_python_host(dev):
    """The torch-extension host (csrc/torch_binding.cpp) with a node per module and the Python host (functional.py / modules.py)
    drive the same kernels in the same order: scores and every parameter gradient must agree bit for bit.  The default host runs
    whole model as ONE node (SminCore): the same kernels except for the parameter products (its own kernel instead of hipBLASLt: scores
    within 2e-6); gradients of tensors with several consumers are summed in one launch instead of by the autograd engine (a different
    but fixed order: tight tolerance)."""
    from oracle import smin_oracle as O
    import vml_amd.training as TR
    from vml_amd import loss_fn
    import models
    for (T, L, C, D, dl, layers, Din, Nq, Hh, B) in [(64, 16, 4, 128, 32, 3, 40, 9, 64, 5), (128, 64, 4, 512, 128, 3, 500, 20, 256, 3),
                                                     (64, 16, 2, 128, 32, 5, 40, 9, 64, 3), (64, 16, 4, 128, 32, 1, 40, 9, 64, 2)]:
        sd = O.formula_state_dict(H.smin_shapes(T, L, C, D, dl, layers, Din, Nq, Hh), gain=1.2)
        batch = O.synthetic_batch(B, T, L, Nq, Din, seed=21)
        b = {k: v.to(dev) for k, v in batch.items()}
        res = []
        for native, fused in ((True, True), (True, False), (False, False)):
            m = build_model(dict(T=T, L=L, C=C, D=D, dl=dl, layers=layers, Din=Din, Nq=Nq, H=Hh), sd, dev)
            m.native_host = native
            m.fused_core = fused
            TR.NATIVE_LOSS = native
            try:
                out = m(*H.model_inputs(b))
                loss = loss_fn(out[0], b["ym"], b["sm"], b["moment_mask"], out[1], b["ys"], b["ss"], out[2], b["ye"], b["se"], out[3], b["ya"], b["length_mask"])
                loss.backward()
            finally:
                TR.NATIVE_LOSS = True
            res.append(([o.detach().clone() for o in out], loss.detach().clone(), {k: p.grad.clone() for k, p in m.named_parameters()}))
        (o0, l0, g0), (o1, l1, g1), (o2, l2, g2) = res
        assert torch.equal(l1, l2) and abs(float(l0) - float(l1)) <= 2e-6 * max(1.0, abs(float(l1)))
        for x, y, z in zip(o0, o1, o2):
            assert torch.equal(y, z)
            assert float((x - y).abs().max()) <= 2e-6          # (the one-node step forms the weight products with its own kernel: last-bit differences)
        gmax = max(float(v.abs().max()) for v in g1.values())
        for k in g1:
            assert torch.equal(g1[k], g2[k]), k
            assert g0[k].shape == g1[k].shape, k
            err = float((g0[k] - g1[k]).abs().max())
            assert err <= 2e-5 * float(g1[k].abs().max()) + 1e-6 * gmax + 1e-9, (k, err, float(g1[k].abs().max()), gmax)
        # the fused node is deterministic: a second run reproduces its gradients bit for bit
        m = build_model(dict(T=T, L=L, C=C, D=D, dl=dl, layers=layers, Din=Din, Nq=Nq, H=Hh), sd, dev)
        out = m(*H.model_inputs(b))
        loss_fn(out[0], b["ym"], b["sm"], b["moment_mask"], out[1], b["ys"], b["ss"], out[2], b["ye"], b["se"], out[3], b["ya"], b["length_mask"]).backward()
        for k, p in m.named_parameters():
            assert torch.equal(p.grad, g0[k]), k
        # ... and where its weight-side work is queued (weight stream or main stream) changes no bit
        for async_weights in (False,):
            m = build_model(dict(T=T, L=L, C=C, D=D, dl=dl, layers=layers, Din=Din, Nq=Nq, H=Hh), sd, dev)
            m.async_weights = async_weights
            out = m(*H.model_inputs(b))
            loss_fn(out[0], b["ym"], b["sm"], b["moment_mask"], out[1], b["ys"], b["ss"], out[2], b["ye"], b["se"], out[3], b["ya"], b["length_mask"]).backward()
            for k, p in m.named_parameters():
                assert torch.equal(p.grad, g0[k]), (k, async_weights)


def test_target_kernel_and_feeder(dev):
    """csrc/labels.hip (one launch for every mask / target of a batch) against the per-sample restatement of dataset.py:95-155
    and the torch form; then the double-buffered BatchFeeder: what it yields equals direct construction, batch after batch, and
    feeds SMIN.  PARITY UNPINNED for the targets (dataset.py is not importable here; the reference holds no fixtures for them)."""
    import models
    from oracle import labels_oracle as LO
    V = models.vml_amd
    g = torch.Generator().manual_seed(5)
    for (T, L, Nq, B) in [(256, 64, 20, 7), (1024, 512, 20, 2), (64, 16, 13, 6)]:
        dur = torch.rand(B, generator=g) * 200 + 5
        ts = torch.rand(B, generator=g) * dur * 0.6
        te = ts + torch.rand(B, generator=g) * (dur - ts) * 0.9 + 0.5
        nf = torch.randint(3, 2 * T, (B,), generator=g)
        ql = torch.randint(1, Nq + 1, (B,), generator=g)
        times = torch.stack([ts, te], 1)
        got = V.build_targets_hip(times.to(dev), dur.to(dev), nf.to(dev), ql.to(dev), T, L, Nq)
        ref_t = V.build_targets(times.to(dev), dur.to(dev), nf.to(dev), T, L, device=dev)
        assert got["video_mask"].dtype == torch.uint8 and got["video_mask"].shape == (B, T, 1) and got["query_mask"].shape == (B, Nq, 1)
        assert got["moment_mask"].dtype == torch.bool and got["length_mask"].dtype == torch.bool
        assert torch.equal(got["query_mask"].reshape(B, Nq).cpu(), (torch.arange(Nq).unsqueeze(0) < ql.unsqueeze(1)).to(torch.uint8))
        for k, v in ref_t.items():
            if v.dtype.is_floating_point:
                assert torch.allclose(got[k], v, rtol=1e-5, atol=1e-6, equal_nan=True), (T, L, k)
        for b in range(B):
            ref = LO.sample_targets(float(ts[b]), float(te[b]), float(dur[b]), int(nf[b]), T, L)
            for k, v in ref.items():
                a = got[k][b].cpu()
                if v.dtype.is_floating_point:
                    assert torch.allclose(a, v, rtol=1e-5, atol=1e-6, equal_nan=True), (T, L, b, k)
                else:                                     # labels are thresholds of fp32 scores: allow a flip only where the score sits on 0.5
                    a, v = a.to(v.dtype).reshape(v.shape), v
                    if not torch.equal(a, v):
                        src = {"ym": "sm", "ys": "ss", "ye": "se"}.get(k)
                        assert src is not None, (T, L, b, k)
                        assert ((ref[src][a != v] - 0.5).abs() < 1e-5).all(), (T, L, b, k)
    # annotation times in double (what a loader has): the Gaussian denominators are formed in double and rounded once, as
    # dataset.py:116-119 does in Python floats -- tighter agreement than through fp32 times
    T, L, Nq, B = 64, 16, 9, 5
    dur = torch.rand(B, generator=g) * 100 + 5
    ts64 = torch.rand(B, generator=g, dtype=torch.float64) * dur.double() * 0.5
    te64 = ts64 + 0.37 + torch.rand(B, generator=g, dtype=torch.float64) * 3.0
    got = V.build_targets_hip(torch.stack([ts64, te64], 1).to(dev), dur.to(dev), torch.full((B,), T).to(dev), None, T, L, Nq)
    for b in range(B):
        ref = LO.sample_targets(float(ts64[b]), float(te64[b]), float(dur[b]), T, T, L)
        for k in ("ss", "se"):
            assert torch.allclose(got[k][b].cpu(), ref[k], rtol=2e-6, atol=1e-30), (b, k)
    # the feeder
    T, L, Nq, Din, B = 64, 16, 9, 40, 4
    hbs = []
    for s in range(5):
        dur = torch.rand(B, generator=g) * 100 + 5
        ts = torch.rand(B, generator=g) * dur * 0.5
        te = ts + 1.0 + torch.rand(B, generator=g) * (dur - ts - 1.0).clamp(min=0)
        hbs.append(dict(video_features=torch.randn(B, T, Din, generator=g).numpy(), query_features=torch.randn(B, Nq, 300, generator=g),
                        nfeats=torch.randint(T // 2, T + 1, (B,), generator=g), qlen=torch.randint(2, Nq + 1, (B,), generator=g),
                        times=torch.stack([ts, te], 1), duration=dur))
    feeder = V.BatchFeeder(T, L, Nq, dev)
    m = models.SMIN(T, L, 4, 64, 32, 2, Din, Nq, 32, dev).to(dev)
    n = 0
    for hb, fed in zip(hbs, feeder.feed(hbs)):
        n += 1
        assert list(fed.keys()) == ["video_features", "video_mask", "query_features", "query_mask", "length_mask", "moment_mask", "sm", "ym", "ss", "ys", "se", "ye", "ya"]
        want = V.build_targets_hip(torch.as_tensor(hb["times"]).to(dev), hb["duration"].to(dev), hb["nfeats"].to(dev), hb["qlen"].to(dev), T, L, Nq)
        assert torch.equal(fed["video_features"].cpu(), torch.as_tensor(hb["video_features"]))
        assert torch.equal(fed["query_features"].cpu(), hb["query_features"])
        for k, v in want.items():
            assert torch.equal(torch.nan_to_num(fed[k].float()), torch.nan_to_num(v.float())), k
        out = m(*H.model_inputs(fed))
        loss = V.loss_fn(out[0], fed["ym"], torch.nan_to_num(fed["sm"]), fed["moment_mask"], out[1], fed["ys"], fed["ss"], out[2], fed["ye"], fed["se"], out[3],
                         fed["ya"], fed["length_mask"])
        assert torch.isfinite(loss)
    assert n == 5


# ---------------------------------------------------------------- size-independent properties at BASELINE size
def test_properties_full_size(dev):
    from oracle import smin_oracle as O
    import models
    T, L, C, D, dl, layers, Din, Nq, Hh = H.FULL["anet_t256"]
    B = 64                                                              # BASELINE configs[2]: the bench's per-GPU batch
    torch.manual_seed(43)
    m = models.SMIN(T, L, C, D, dl, layers, Din, Nq, Hh, dev).to(dev)
    batch = O.synthetic_batch(B, T, L, Nq, Din, seed=1)
    b = {k: v.to(dev) for k, v in batch.items()}
    with torch.no_grad():
        out1 = m(*H.model_inputs(b))
        out2 = m(*H.model_inputs(b))
        for x, y in zip(out1, out2):
            assert torch.equal(x, y)                                   # deterministic
        pm = out1[0]
        assert pm[~b["moment_mask"]].abs().max().item() == 0.0          # masked proposals are exactly 0
        assert out1[1][~b["length_mask"]].abs().max().item() == 0.0
        assert ((pm >= 0) & (pm <= 1)).all()
        # batch independence: a sample scored alone equals the sample scored inside the batch
        for s in (0, 3, 63):
            one = m(*[x[s:s + 1] for x in H.model_inputs(b)])
            for x, y in zip(one, out1):
                assert (x[0] - y[s]).abs().max().item() < 1e-5


def test_gradients_deterministic(dev):
    from oracle import smin_oracle as O
    from vml_amd import loss_fn
    import models
    T, L, C, D, dl, layers, Din, Nq, Hh = H.FULL["charades"]
    torch.manual_seed(1)
    m = models.SMIN(T, L, C, D, dl, layers, Din, Nq, Hh, dev).to(dev)
    batch = O.synthetic_batch(4, T, L, Nq, Din, seed=2)
    b = {k: v.to(dev) for k, v in batch.items()}
    snaps = []
    for _ in range(2):
        m.zero_grad(set_to_none=True)
        pm, ps, pe, pa = m(*H.model_inputs(b))
        loss_fn(pm, b["ym"], b["sm"], b["moment_mask"], ps, b["ys"], b["ss"], pe, b["ye"], b["se"], pa, b["ya"], b["length_mask"]).backward()
        snaps.append({k: p.grad.clone() for k, p in m.named_parameters()})      # backbone (bilstm / video encoder kernels) included
    for k in snaps[0]:
        assert torch.equal(snaps[0][k], snaps[1][k]), k          # fixed-order reductions: bitwise reproducible


# ---------------------------------------------------------------- split-bf16 ("bf16x3") contraction mode
@pytest.fixture()
def bf16x3():
    import models
    models.vml_amd.set_gemm_mode("bf16x3")
    yield
    models.vml_amd.set_gemm_mode(models.vml_amd._lib.DEFAULT_GEMM_MODE)


@pytest.mark.parametrize("M,N,K", [(1000, 128, 512), (515, 512, 128), (4096, 512, 1024), (37, 20, 16)])
def test_gemm_nt_engine_bf16x3(dev, bf16x3, M, N, K):
    from vml_amd.functional import gemm_nt
    g = torch.Generator().manual_seed(M + N + K)
    a = torch.randn(M, K, generator=g)
    b = torch.randn(N, K, generator=g)
    ref = (a.double() @ b.double().t())
    got = gemm_nt(a.to(dev), b.to(dev)).cpu().double()
    rel = ((got - ref).abs().max() / ref.abs().max()).item()
    print("bf16x3 gemm", (M, N, K), "max err / max |C|", rel)
    assert rel < 2e-5            # ~2^-16 per product; fp32 mode reaches ~1e-7 here


@pytest.mark.parametrize("name", ["charades", "anet_t256"])
def test_full_size_against_golden_bf16x3(dev, bf16x3, name):
    """Same golden check as the fp32 mode with the tolerance this arithmetic earns: scores 1e-4 (required 1e-3)."""
    from oracle import smin_oracle as O
    from vml_amd import loss_fn
    z = H.load_npz("g5_" + name)
    T, L, C, D, dl, layers, Din, Nq, Hh = H.FULL[name]
    B, seed = int(z["cfg"][-2]), int(z["cfg"][-1])
    sd = O.formula_state_dict(H.smin_shapes(T, L, C, D, dl, layers, Din, Nq, Hh), gain=1.3)
    m = build_model(dict(T=T, L=L, C=C, D=D, dl=dl, layers=layers, Din=Din, Nq=Nq, H=Hh), sd, dev)
    batch = O.synthetic_batch(B, T, L, Nq, Din, seed=seed)
    b = {k: v.to(dev) for k, v in batch.items()}
    pm, ps, pe, pa = m(*H.model_inputs(b))
    for k, v in (("pm", pm), ("ps", ps), ("pe", pe), ("pa", pa)):
        err = (v.detach().cpu() - torch.from_numpy(z["out/" + k])).abs().max().item()
        print("bf16x3", name, k, "max abs err", err)
        assert err < 1e-4, (k, err)
    loss = loss_fn(pm, b["ym"], b["sm"], b["moment_mask"], ps, b["ys"], b["ss"], pe, b["ye"], b["se"], pa, b["ya"], b["length_mask"])
    assert abs(loss.item() - float(z["loss"])) < 1e-4
    loss.backward()
    norms = dict(zip([str(s) for s in z["grad_names"]], z["grad_norms"]))
    worst = 0.0
    for k, p in m.named_parameters():
        got = p.grad.double().norm().item()
        worst = max(worst, abs(got - norms[k]) / (norms[k] + 1e-12))
        assert abs(got - norms[k]) <= 5e-3 * norms[k] + 1e-7, (k, got, norms[k])
    print("bf16x3", name, "worst grad-norm deviation", worst)


# ---------------------------------------------------------------- fp32 emulated on the bf16 matrix cores ("f32e")
@pytest.fixture()
def f32e():
    import models
    models.vml_amd.set_gemm_mode("f32e")
    yield
    models.vml_amd.set_gemm_mode(models.vml_amd._lib.DEFAULT_GEMM_MODE)


@pytest.mark.parametrize("M,N,K", [(1000, 128, 512), (515, 512, 128), (4096, 512, 1024), (37, 20, 16), (300, 64, 2052)])
def test_gemm_engines_f32e(dev, M, N, K):
    """Three-way bf16 split, six products, fp32 accumulation: the error against fp64 must sit where the exact fp32 MFMA's does
    (both are dominated by fp32 accumulation), for the NT engine and -- through the linear-rows weight gradient -- the TN engine."""
    import models
    from vml_amd.functional import gemm_nt, LinearRowsFn
    g = torch.Generator().manual_seed(M + N + K)
    scale = torch.exp(torch.randn(M, 1, generator=g) * 2)              # rows of very different magnitude: the split is per element
    a = torch.randn(M, K, generator=g) * scale
    b = torch.randn(N, K, generator=g)
    ref = a.double() @ b.double().t()
    errs = {}
    for mode in ("f32", "f32e"):
        models.vml_amd.set_gemm_mode(mode)
        try:
            got = gemm_nt(a.to(dev), b.to(dev)).cpu().double()
            errs[mode] = ((got - ref).abs() / (a.double().abs() @ b.double().abs().t() + 1e-30)).max().item()
            if K % 4 == 0 and N % 4 == 0:
                x = a.to(dev).clone().requires_grad_(True)
                w = b.to(dev).clone().requires_grad_(True)
                y = LinearRowsFn.apply(w, None, None, None, 1, x)
                dy = torch.randn(M, N, generator=g).to(dev)
                y.backward(dy)
                dw_ref = dy.cpu().double().t() @ a.double()
                dx_ref = dy.cpu().double() @ b.double()
                errs[mode + "/dW"] = ((w.grad.cpu().double() - dw_ref).abs() / (dy.cpu().double().abs().t() @ a.double().abs() + 1e-30)).max().item()
                errs[mode + "/dX"] = ((x.grad.cpu().double() - dx_ref).abs() / (dy.cpu().double().abs() @ b.double().abs() + 1e-30)).max().item()
        finally:
            models.vml_amd.set_gemm_mode(models.vml_amd._lib.DEFAULT_GEMM_MODE)
    print("f32e gemm", (M, N, K), errs)
    for k, v in errs.items():
        assert v < 1e-6, (k, v)                                         # componentwise: |err| <= 4e-7 * sum |a||b|  (fp32 unit roundoff 6e-8)
    for k in [k for k in errs if k.startswith("f32e")]:
        assert errs[k] <= 4 * errs[k.replace("f32e", "f32")] + 6e-8, (k, errs)


@pytest.mark.parametrize("name", ["charades", "anet_t256", "tacos_d500"])
def test_full_size_against_golden_f32e(dev, f32e, name):
    """The fp32 golden check at the fp32 mode's own tolerances, with every contraction emulated on the bf16 cores."""
    from oracle import smin_oracle as O
    from vml_amd import loss_fn
    z = H.load_npz("g5_" + name)
    T, L, C, D, dl, layers, Din, Nq, Hh = H.FULL[name]
    B, seed = int(z["cfg"][-2]), int(z["cfg"][-1])
    sd = O.formula_state_dict(H.smin_shapes(T, L, C, D, dl, layers, Din, Nq, Hh), gain=1.3)
    m = build_model(dict(T=T, L=L, C=C, D=D, dl=dl, layers=layers, Din=Din, Nq=Nq, H=Hh), sd, dev)
    batch = O.synthetic_batch(B, T, L, Nq, Din, seed=seed)
    b = {k: v.to(dev) for k, v in batch.items()}
    pm, ps, pe, pa = m(*H.model_inputs(b))
    for k, v in (("pm", pm), ("ps", ps), ("pe", pe), ("pa", pa)):
        err = (v.detach().cpu() - torch.from_numpy(z["out/" + k])).abs().max().item()
        print("f32e", name, k, "max abs err", err)
        assert err < 5e-6, (k, err)               # fp32 mode: 6e-8 .. 8e-7; its test allows 1e-4, north_star 1e-3
    loss = loss_fn(pm, b["ym"], b["sm"], b["moment_mask"], ps, b["ys"], b["ss"], pe, b["ye"], b["se"], pa, b["ya"], b["length_mask"])
    assert abs(loss.item() - float(z["loss"])) < 1e-5
    loss.backward()
    norms = dict(zip([str(s) for s in z["grad_names"]], z["grad_norms"]))
    for k, p in m.named_parameters():
        got = p.grad.double().norm().item()
        assert abs(got - norms[k]) <= 2e-3 * norms[k] + 1e-7, (k, got, norms[k])
    for k in [f for f in z.files if f.startswith("grad/")]:
        ref = torch.from_numpy(z[k])
        got = dict(m.named_parameters())[k[5:]].grad.cpu()
        assert (got - ref).abs().max().item() <= 2e-3 * ref.abs().max().item() + 1e-7, k


# ---------------------------------------------------------------- plain bf16 contraction mode (BASELINE.json configs[1])
@pytest.fixture()
def bf16_mode():
    import models
    models.vml_amd.set_gemm_mode("bf16")
    yield
    models.vml_amd.set_gemm_mode(models.vml_amd._lib.DEFAULT_GEMM_MODE)


@pytest.mark.parametrize("M,N,K", [(1000, 128, 512), (515, 512, 128), (4096, 512, 1024), (37, 20, 16)])
def test_gemm_nt_engine_bf16(dev, bf16_mode, M, N, K):
    from vml_amd.functional import gemm_nt
    g = torch.Generator().manual_seed(M + N + K)
    a = torch.randn(M, K, generator=g)
    b = torch.randn(N, K, generator=g)
    got = gemm_nt(a.to(dev), b.to(dev)).cpu().double()
    # exactly the product of the bf16-rounded operands, accumulated in fp32
    ref = a.bfloat16().double() @ b.bfloat16().double().t()
    assert ((got - ref).abs().max() / ref.abs().max()).item() < 2e-6
    full = a.double() @ b.double().t()
    print("bf16 gemm", (M, N, K), "error vs fp64 / max |C|", ((got - full).abs().max() / full.abs().max()).item())


@pytest.mark.parametrize("name", ["charades", "anet_t256"])
def test_full_size_against_golden_bf16(dev, bf16_mode, name):
    """BASELINE.json configs[1] (bf16): every contraction -- forward, input gradients and weight gradients -- with operands
    rounded to bf16 and fp32 accumulation.  The reference has no bf16 path (casting it raises), so the gate is the
    fp32 golden output at SURVEY's tolerance for this mode: 2e-2 on the scores; the achieved error is printed."""
    from oracle import smin_oracle as O
    from vml_amd import loss_fn
    z = H.load_npz("g5_" + name)
    T, L, C, D, dl, layers, Din, Nq, Hh = H.FULL[name]
    B, seed = int(z["cfg"][-2]), int(z["cfg"][-1])
    sd = O.formula_state_dict(H.smin_shapes(T, L, C, D, dl, layers, Din, Nq, Hh), gain=1.3)
    m = build_model(dict(T=T, L=L, C=C, D=D, dl=dl, layers=layers, Din=Din, Nq=Nq, H=Hh), sd, dev)
    batch = O.synthetic_batch(B, T, L, Nq, Din, seed=seed)
    b = {k: v.to(dev) for k, v in batch.items()}
    pm, ps, pe, pa = m(*H.model_inputs(b))
    worst = 0.0
    for k, v in (("pm", pm), ("ps", ps), ("pe", pe), ("pa", pa)):
        err = (v.detach().cpu() - torch.from_numpy(z["out/" + k])).abs().max().item()
        worst = max(worst, err)
        print("bf16", name, k, "max abs err", err)
        assert err < 2e-2, (k, err)
    loss = loss_fn(pm, b["ym"], b["sm"], b["moment_mask"], ps, b["ys"], b["ss"], pe, b["ye"], b["se"], pa, b["ya"], b["length_mask"])
    assert abs(loss.item() - float(z["loss"])) < 2e-2 * max(1.0, abs(float(z["loss"])))
    loss.backward()
    norms = dict(zip([str(s) for s in z["grad_names"]], z["grad_norms"]))
    dev_w = 0.0
    gmax = max(norms.values())
    for k, p in m.named_parameters():
        got = p.grad.double().norm().item()
        assert torch.isfinite(p.grad).all(), k
        if norms[k] > 1e-3 * gmax:                       # gradients that matter: norm within 5 %
            dev_w = max(dev_w, abs(got - norms[k]) / norms[k])
            assert abs(got - norms[k]) <= 5e-2 * norms[k], (k, got, norms[k])
    print("bf16", name, "worst score error", worst, "worst gradient-norm deviation", dev_w)


def test_bf16_operand_storage_changes_no_bit(dev, bf16_mode):
    """Under plain bf16 contractions the pair product f_b[i]*f_b[j] is stored as bf16 (it only ever feeds contractions, whose loaders round
    it to bf16 anyway): scores and every gradient equal the fp32-storage step bit for bit; and csrc's producer writes exactly torch's
    round-to-nearest-even bf16 of the fp32 product."""
    import ctypes
    from oracle import smin_oracle as O
    from vml_amd import loss_fn
    from vml_amd._lib import call, ptr, stream
    T, L, C, D, dl, layers, Din, Nq, Hh, B = 128, 64, 4, 512, 128, 3, 500, 20, 256, 3
    sd = O.formula_state_dict(H.smin_shapes(T, L, C, D, dl, layers, Din, Nq, Hh), gain=1.2)
    batch = O.synthetic_batch(B, T, L, Nq, Din, seed=5)
    b = {k: v.to(dev) for k, v in batch.items()}
    res = []
    for store in (True, False):
        m = build_model(dict(T=T, L=L, C=C, D=D, dl=dl, layers=layers, Din=Din, Nq=Nq, H=Hh), sd, dev)
        m.bf16_operand_storage = store
        out = m(*H.model_inputs(b))
        loss_fn(out[0], b["ym"], b["sm"], b["moment_mask"], out[1], b["ys"], b["ss"], out[2], b["ye"], b["se"], out[3], b["ya"], b["length_mask"]).backward()
        res.append(([o.detach().clone() for o in out], {k: p.grad.clone() for k, p in m.named_parameters()}))
    for x, y in zip(res[0][0], res[1][0]):
        assert torch.equal(x, y)
    for k in res[0][1]:
        assert torch.equal(res[0][1][k], res[1][1][k]), k
    g = torch.Generator().manual_seed(9)
    Bn, Ln, Dn = 2, 7, 64
    fb = torch.randn(Bn, Ln, Dn, generator=g).to(dev)
    cells = torch.tensor([[bb, i, j, 1] for bb in range(Bn) for i in range(Ln) for j in range(i, Ln)], dtype=torch.int32, device=dev)
    x1h = torch.empty(cells.shape[0], Dn, dtype=torch.bfloat16, device=dev)
    call("smin_pair_product_bf16", stream(), ptr(fb), ptr(cells), cells.shape[0], Ln, Dn, ctypes.c_void_p(x1h.data_ptr()))
    ref = (fb[cells[:, 0].long(), cells[:, 1].long()] * fb[cells[:, 0].long(), cells[:, 2].long()]).to(torch.bfloat16)
    assert torch.equal(x1h.view(torch.int16), ref.view(torch.int16))


def test_bf16_mode_with_no_valid_cell(dev, bf16_mode):
    """bf16-stored operands and an empty cell list (every length mask false): zero scores, finite gradients, no rejected argument."""
    from oracle import smin_oracle as O
    T, L, C, D, dl, layers, Din, Nq, Hh, B = 64, 16, 4, 128, 32, 3, 40, 9, 64, 3
    sd = O.formula_state_dict(H.smin_shapes(T, L, C, D, dl, layers, Din, Nq, Hh), gain=1.2)
    m = build_model(dict(T=T, L=L, C=C, D=D, dl=dl, layers=layers, Din=Din, Nq=Nq, H=Hh), sd, dev)
    b = {k: v.to(dev) for k, v in O.synthetic_batch(B, T, L, Nq, Din, seed=4).items()}
    b["length_mask"][:] = False
    b["moment_mask"][:] = False
    out = m(*H.model_inputs(b))
    assert all(float(o.detach().abs().max()) == 0.0 for o in out)
    sum(o.sum() for o in out).backward()
    assert all(torch.isfinite(p.grad).all() for p in m.parameters() if p.grad is not None)


def test_fused_loss_matches_torch_restatement(dev):
    """vml_amd.loss_fn on device (two fused kernels) against the torch restatement: value and the four score gradients."""
    from oracle import smin_oracle as O
    from vml_amd import loss_fn, loss_fn_torch
    cfg, sd, batch, out, _, loss_ref = H.split_tiny(H.load_npz("g2_r2_ragged"))
    b = {k: v.to(dev) for k, v in batch.items()}
    leaves = [out[k].to(dev).clone().requires_grad_(True) for k in ("pm", "ps", "pe", "pa")]
    ref_leaves = [out[k].to(dev).clone().requires_grad_(True) for k in ("pm", "ps", "pe", "pa")]
    args = lambda p: (p[0], b["ym"], b["sm"], b["moment_mask"], p[1], b["ys"], b["ss"], p[2], b["ye"], b["se"], p[3], b["ya"], b["length_mask"])
    l1 = loss_fn(*args(leaves))
    l0 = loss_fn_torch(*args(ref_leaves))
    assert abs(l1.item() - l0.item()) < 1e-5 and abs(l1.item() - loss_ref) < 1e-5
    (l1 * 1.7).backward()
    (l0 * 1.7).backward()
    for a, r in zip(leaves, ref_leaves):
        assert (a.grad - r.grad).abs().max().item() <= 1e-5 * r.grad.abs().max().item() + 1e-9


def test_compute_ious_on_device(dev):
    """utils.py:10-31 metric: the one-kernel device version against the reference's own fixture and the torch restatement."""
    import models
    z = H.load_npz("g6_ious")
    args = [torch.from_numpy(z[k]).to(dev) for k in ("pm", "ps", "pe", "mm", "sm")]
    got = models.vml_amd.compute_ious(*args)
    for k, v in zip(z["keys"], z["vals"]):
        assert got[str(k)] == float(v), k
    g = torch.Generator().manual_seed(11)
    B, L = 9, 64
    lm = torch.ones(B, L, dtype=torch.bool); lm[2, 40:] = False
    mm = torch.triu(lm.unsqueeze(2) & lm.unsqueeze(1))
    pm, ps, pe, sm = torch.rand(B, L, L, generator=g) * mm, torch.rand(B, L, generator=g), torch.rand(B, L, generator=g), torch.rand(B, L, L, generator=g) * mm
    want = models.vml_amd.compute_ious_torch(pm, ps, pe, mm, sm)                # the torch restatement, on the host
    got = models.vml_amd.compute_ious(*(x.to(dev) for x in (pm, ps, pe, mm, sm)))
    assert got == want


# ---------------------------------------------------------------- C ABI entry points not reached through the module mirror
def test_c_abi_layout_helpers_and_boundary_reduce(dev):
    import models
    from vml_amd._lib import call, ptr, stream
    g = torch.Generator().manual_seed(21)
    B, L, D = 3, 12, 32
    mm = (torch.rand(B, L, L, generator=g) > 0.5).to(dev)
    lay = models.vml_amd.CellLayout.from_mask(mm)
    dense = torch.randn(B, L, L, D, generator=g).to(dev)
    packed = torch.empty(lay.N, D, device=dev)
    call("smin_pack_cells", stream(), ptr(dense), ptr(lay.cells), lay.N, L, D, ptr(packed))
    assert torch.equal(packed, lay.pack(dense))
    back = torch.zeros_like(dense)
    call("smin_unpack_cells", stream(), ptr(packed), ptr(lay.cells), lay.N, L, D, ptr(back))
    assert torch.equal(back, dense * mm.unsqueeze(-1))
    # fbm[b,i] = sum_j Ab[b,i,j] hbar[(b,i,j)]  and its two gradients
    Ab = torch.randn(B, L, L, generator=g).to(dev)
    hbar = torch.randn(lay.N, D, generator=g).to(dev)
    fbm = torch.empty(B, L, D, device=dev)
    call("smin_boundary_reduce_fwd", stream(), ptr(Ab), ptr(hbar), ptr(lay.cells), ptr(lay.row_ptr), lay.N, B, L, D, ptr(fbm))
    hd = lay.unpack(hbar)
    ref = (Ab.unsqueeze(-1) * hd).sum(dim=2)
    assert (fbm - ref).abs().max().item() < 1e-5
    dy = torch.randn(B, L, D, generator=g).to(dev)
    dAb, dh = torch.empty_like(Ab), torch.empty_like(hbar)
    call("smin_boundary_reduce_bwd", stream(), ptr(dy), ptr(Ab), ptr(hbar), ptr(lay.cells), ptr(lay.row_ptr), lay.N, B, L, D, ptr(dAb), ptr(dh))
    assert (dAb - (dy.unsqueeze(2) * hd).sum(-1) * mm).abs().max().item() < 1e-4
    assert (dh - lay.pack(Ab.unsqueeze(-1) * dy.unsqueeze(2))).abs().max().item() < 1e-5


# ---------------------------------------------------------------- content stream: pieces and whole
def test_clip_event_table_matches_content_matrix(dev):
    """The clip-boundary table is the sparsity pattern of the reference's content matrix: integrating its +-1/cs
    entries over time must give back compute_content_matrix (golden g4_wc, produced from the reference) exactly."""
    import models
    from vml_amd.functional import clip_event_table
    z = H.load_npz("g4_wc")
    for key in [k[6:] for k in z.files if k.startswith("start/")]:
        T, L, C = (int(v) for v in key.split("_"))
        size = torch.from_numpy(z["size/" + key]).long()
        start = torch.from_numpy(z["start/" + key]).long()
        val = torch.from_numpy(z["val/" + key])
        t = torch.arange(T)
        Wc = ((t >= start.unsqueeze(-1)) & (t < (start + size).unsqueeze(-1))).float() * val.unsqueeze(-1)   # (L, L, C, T)
        off, tab = (x.cpu() for x in clip_event_table(dev, T, L, C))
        diff = torch.zeros(L, L, C, T + 1)
        nclip = (size > 0).sum(-1)                                           # clips present per cell
        for tt in range(T):
            e = tab[off[tt]:off[tt + 1]]
            key_, sc = e[:, 0], e[:, 1].contiguous().view(torch.float32)
            i, j, c, wm = key_ >> 18, (key_ >> 6) & 4095, key_ & 7, (key_ >> 3) & 1
            diff[i.long(), j.long(), c.long(), tt] += sc
            # the mean-path flag marks the first clip's start and the last clip's end of a cell
            want = torch.where(sc > 0, c == 0, c + 1 == nclip[i.long(), j.long()])
            assert torch.equal(wm.bool(), want), key
        assert torch.equal(diff.cumsum(-1)[..., :T], Wc), key


@pytest.mark.parametrize("R,O,K,nseg,C", [(1000, 128, 128, 1, 4), (515, 64, 32, 2, 1), (2048, 512, 128, 1, 1), (999, 128, 128, 3, 3), (64, 24, 16, 4, 2)])
def test_linear_rows_forward_backward(dev, R, O, K, nseg, C):
    from vml_amd.functional import LinearRowsFn
    g = torch.Generator().manual_seed(R + O + K)
    R = (R // C) * C
    xs = [torch.randn(R, K, generator=g, dtype=torch.float64).requires_grad_(True) for _ in range(nseg)]
    W = torch.randn(O, nseg * K, generator=g, dtype=torch.float64).requires_grad_(True)
    bias = torch.randn(O, generator=g, dtype=torch.float64).requires_grad_(True)
    ar = torch.randn(R, O, generator=g, dtype=torch.float64).requires_grad_(True)
    ac = torch.randn(R // C, O, generator=g, dtype=torch.float64).requires_grad_(True)
    ref = torch.cat(xs, 1) @ W.t() + bias + ar + ac.repeat_interleave(C, 0)
    dy = torch.randn(R, O, generator=g, dtype=torch.float64)
    ref.backward(dy)
    leaves = [W, bias, ar, ac] + xs
    got_in = [t.detach().float().to(dev).requires_grad_(True) for t in leaves]
    y = LinearRowsFn.apply(got_in[0], got_in[1], got_in[2], got_in[3], C, *got_in[4:])
    tol = 4e-6 * (nseg * K) ** 0.5 * 8
    assert (y.detach().cpu().double() - ref.detach()).abs().max().item() <= tol
    y.backward(dy.float().to(dev))
    for a, b in zip(got_in, leaves):
        scale = max(1.0, b.grad.abs().max().item())
        assert (a.grad.cpu().double() - b.grad).abs().max().item() <= 2e-5 * scale, tuple(b.shape)


def test_clip_window_means_forward_backward(dev):
    """Clip means of a projected frame feature, one tensor per segment, against the dense content-matrix einsum
    (models.py:117) and its autograd gradient."""
    import models
    from vml_amd.functional import ClipWindowMeansFn
    g = torch.Generator().manual_seed(5)
    for (T, L, C, W, nseg, B) in [(32, 8, 4, 16, 3, 3), (16, 16, 4, 8, 2, 2), (24, 12, 3, 24, 1, 2)]:
        lm = torch.zeros(B, L, dtype=torch.bool)
        for b in range(B):
            lm[b, : L - 2 * b] = True
        mm = torch.triu(lm.unsqueeze(2) & lm.unsqueeze(1))
        lay = models.vml_amd.CellLayout.from_mask(mm.to(dev))
        gfeat = torch.randn(B, T, nseg * W, generator=g, dtype=torch.float64).requires_grad_(True)
        bias = torch.randn(nseg * W, generator=g, dtype=torch.float64).requires_grad_(True)
        Wc = models.compute_content_matrix(T, L, C).double()
        dense = torch.einsum("ijct,btd->bijcd", Wc, gfeat) + bias                     # (B, L, L, C, nseg*W)
        rows = dense[mm]                                                              # (N, C, nseg*W), sorted by (b, i, j)
        refs = [rows[..., s * W:(s + 1) * W].reshape(-1, W) for s in range(nseg)]
        dys = [torch.randn(r.shape, generator=g, dtype=torch.float64) for r in refs]
        sum((r * d).sum() for r, d in zip(refs, dys)).backward()
        gi = gfeat.detach().float().to(dev).requires_grad_(True)
        bi = bias.detach().float().to(dev).requires_grad_(True)
        outs = ClipWindowMeansFn.apply(gi, bi, lay, T, L, C, nseg)
        for o, r in zip(outs, refs):
            assert (o.detach().cpu().double() - r.detach()).abs().max().item() < 2e-6
        sum((o * d.float().to(dev)).sum() for o, d in zip(outs, dys)).backward()
        assert (gi.grad.cpu().double() - gfeat.grad).abs().max().item() < 2e-5 * max(1.0, gfeat.grad.abs().max().item())
        assert (bi.grad.cpu().double() - bias.grad).abs().max().item() < 2e-5 * max(1.0, bias.grad.abs().max().item())


@pytest.mark.parametrize("name", ["g1_r4", "g2_r2_ragged"])
def test_content_stream_equals_unit_as_written(dev, name):
    """SMIN with the content unit re-associated into the dl-dimensional space (content stream) against the same model
    running every unit as the reference writes it: scores, loss and every parameter gradient."""
    from vml_amd import loss_fn
    cfg, sd, batch, out, grads, loss_ref = H.split_tiny(H.load_npz(name))
    b = {k: v.to(dev) for k, v in batch.items()}
    res = []
    for stream_on in (True, False):
        m = build_model(cfg, sd, dev)
        m.content_stream = stream_on
        pm, ps, pe, pa = m(*H.model_inputs(b))
        loss = loss_fn(pm, b["ym"], b["sm"], b["moment_mask"], ps, b["ys"], b["ss"], pe, b["ye"], b["se"], pa, b["ya"], b["length_mask"])
        loss.backward()
        res.append(((pm, ps, pe, pa), loss.item(), {k: p.grad.clone() for k, p in m.named_parameters()}))
    for a, c in zip(res[0][0], res[1][0]):
        assert (a - c).abs().max().item() < 5e-6
    assert abs(res[0][1] - res[1][1]) < 5e-6
    for k in res[0][2]:
        ga, gc = res[0][2][k], res[1][2][k]
        assert (ga - gc).abs().max().item() <= 5e-4 * gc.abs().max().item() + 1e-7, k


# ---------------------------------------------------------------- query encoder: fused BiLSTM layer kernels
def test_cluster_recurrences_from_two_streams_take_turns(dev):
    """Two query encoders stepped on two streams at once: the cluster recurrence (W_hh resident in LDS, one workgroup per CU,
    granule exchange through one buffer per device) must not run twice at a time -- the library orders the launches behind each
    other.  Results equal the one-stream results bit for bit and no bounded poll expires."""
    import models
    B, Nq, H = 64, 20, 256
    g = torch.Generator().manual_seed(77)
    qes = []
    for _ in range(2):
        qe = models.QueryEncoder(Nq, H)
        for p in qe.parameters():
            p.data = (torch.rand(p.shape, generator=g) - 0.5) * 0.6
        qes.append(qe.to(dev))
    xs = [torch.randn(B, Nq, 300, generator=g).to(dev) for _ in range(2)]
    mask = torch.ones(B, Nq, 1, dtype=torch.uint8, device=dev)

    def step(qe, x):
        qe.zero_grad(set_to_none=True)
        fs, fw = qe(x, mask)
        (fw.sum() + fs.sum()).backward()
        return [fs.detach().clone(), fw.detach().clone()] + [p.grad.clone() for p in qe.parameters()]

    ref = [step(qe, x) for qe, x in zip(qes, xs)]
    torch.cuda.synchronize()
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    for rep in range(3):
        got = []
        for s in streams:
            s.wait_stream(torch.cuda.current_stream())
        for qe, x, s in zip(qes, xs, streams):
            with torch.cuda.stream(s):
                got.append(step(qe, x))
        torch.cuda.synchronize()
        for r, o in zip(ref, got):
            for a, b in zip(r, o):
                assert torch.equal(a, b)
    assert models.vml_amd._lib.load().smin_lstm_cluster_error() == 0, "a bounded poll of the cluster recurrence expired"


@pytest.mark.parametrize("B,Nq,H", [(5, 7, 16), (64, 20, 256), (3, 32, 64), (1, 4, 32), (6, 9, 128), (37, 5, 256)])
def test_query_encoder_matches_packed_lstm(dev, B, Nq, H):
    """QueryEncoder on the HIP BiLSTM kernels against the reference formulation (models.py:46-63): nn.LSTM over
    pack_padded_sequence / pad_packed_sequence in fp64 on the CPU -- outputs and every parameter gradient."""
    import models
    from torch.nn.utils.rnn import pack_padded_sequence, pad_packed_sequence
    g = torch.Generator().manual_seed(B * 100 + Nq + H)
    qe = models.QueryEncoder(Nq, H)
    for p in qe.parameters():
        p.data = (torch.rand(p.shape, generator=g) - 0.5) * 0.6
    x = torch.randn(B, Nq, 300, generator=g)
    lens = torch.randint(1, Nq + 1, (B,), generator=g)
    lens[0] = Nq
    mask = (torch.arange(Nq).unsqueeze(0) < lens.unsqueeze(1)).to(torch.uint8).unsqueeze(-1)
    x = x * mask
    # reference: packed nn.LSTM, fp64
    ref = torch.nn.LSTM(300, H, num_layers=2, bidirectional=True, batch_first=True).double()
    ref.load_state_dict({k: v.double() for k, v in qe.lstm.state_dict().items()})
    packed = pack_padded_sequence(x.double(), lens, batch_first=True, enforce_sorted=False)
    out, _ = ref(packed)
    fw_ref, _ = pad_packed_sequence(out, batch_first=True, total_length=Nq)
    fs_ref = torch.cat([fw_ref[torch.arange(B), lens - 1, :H], fw_ref[:, 0, H:]], dim=1)
    wf, ws = torch.randn(B, Nq, 2 * H, generator=g).double(), torch.randn(B, 2 * H, generator=g).double()
    ((fw_ref * wf).sum() + (fs_ref * ws).sum()).backward()
    # HIP path
    qd = qe.to(dev)
    fs, fw = qd(x.to(dev), mask.to(dev))
    assert (fw.detach().cpu().double() - fw_ref.detach()).abs().max().item() < 5e-6
    assert (fs.detach().cpu().double() - fs_ref.detach()).abs().max().item() < 5e-6
    ((fw * wf.float().to(dev)).sum() + (fs * ws.float().to(dev)).sum()).backward()
    for (k, p), (_, r) in zip(qd.lstm.named_parameters(), ref.named_parameters()):
        assert (p.grad.cpu().double() - r.grad).abs().max().item() <= 2e-4 * r.grad.abs().max().item() + 1e-6, k
    assert models.vml_amd._lib.load().smin_lstm_cluster_error() == 0, "a bounded poll of the cluster recurrence expired"
    # and the library path gives the same
    qd.fused_lstm = False
    fs2, fw2 = qd(x.to(dev), mask.to(dev))
    assert (fw2 - fw).abs().max().item() < 2e-5 and (fs2 - fs).abs().max().item() < 2e-5


def test_gate_gradient_lists_and_pair_product(dev):
    """GateFn with several views of hbar / f_m (their gradients are summed by the backward kernel from pointer lists)
    against torch autograd on the formula of models.py:191, 272-274; and the moment unit with the materialised pair
    product against the on-the-fly operand (same C entry point, x1 given / NULL)."""
    import models
    from vml_amd._lib import call, ptr, stream
    from vml_amd.functional import GateFn
    g = torch.Generator().manual_seed(11)
    B, L, D = 3, 10, 48
    mm = torch.triu(torch.ones(L, L, dtype=torch.bool)).unsqueeze(0).repeat(B, 1, 1)
    mm[1, :, 7:] = False
    lay = models.vml_amd.CellLayout.from_mask(mm.to(dev))
    N = lay.N
    fm = torch.randn(N, D, generator=g, dtype=torch.float64)
    fs = torch.randn(B, D, generator=g, dtype=torch.float64)
    ws = [torch.randn(N, D, generator=g, dtype=torch.float64) for _ in range(6)]
    fm_r, fs_r = fm.clone().requires_grad_(True), fs.clone().requires_grad_(True)
    b_idx = lay.cells[:, 0].long().cpu()
    hbar = torch.sigmoid(fm_r * fs_r[b_idx]) * fm_r
    (sum((hbar * w).sum() for w in ws[:4]) + sum((fm_r * w).sum() for w in ws[4:])).backward()
    fm_d, fs_d = fm.float().to(dev).requires_grad_(True), fs.float().to(dev).requires_grad_(True)
    outs = GateFn.apply(fm_d, fs_d, lay, 4, 2)
    assert len(outs) == 6 and (outs[0].detach().cpu().double() - hbar.detach()).abs().max().item() < 1e-5
    sum((o * w.float().to(dev)).sum() for o, w in zip(outs, ws)).backward()
    assert (fm_d.grad.cpu().double() - fm_r.grad).abs().max().item() < 2e-5 * max(1.0, fm_r.grad.abs().max().item())
    assert (fs_d.grad.cpu().double() - fs_r.grad).abs().max().item() < 2e-4 * max(1.0, fs_r.grad.abs().max().item())
    # moment unit forward: x1 given vs formed on the fly
    fb = torch.randn(B, L, D, generator=g).to(dev)
    fcm = torch.randn(N, D, generator=g).to(dev)
    fmf = fm.float().to(dev)
    Wcat = (torch.randn(D, 2 * D, generator=g) * 0.1).to(dev)
    bcat = torch.randn(D, generator=g).to(dev)
    x1 = torch.empty(N, D, device=dev)
    call("smin_pair_product", stream(), ptr(fb), ptr(lay.cells), N, L, D, ptr(x1))
    ci, cj = lay.cells[:, 1].long(), lay.cells[:, 2].long()
    assert torch.equal(x1, fb[lay.cells[:, 0].long(), ci] * fb[lay.cells[:, 0].long(), cj])
    mu_a, mu_b = torch.empty(N, D, device=dev), torch.empty(N, D, device=dev)
    call("smin_moment_unit_fwd", stream(), ptr(fcm), ptr(fmf), ptr(fb), ptr(lay.cells), N, B, L, D, ptr(Wcat), ptr(bcat), ptr(mu_a), ptr(x1))
    call("smin_moment_unit_fwd", stream(), ptr(fcm), ptr(fmf), ptr(fb), ptr(lay.cells), N, B, L, D, ptr(Wcat), ptr(bcat), ptr(mu_b), None)
    ref = torch.cat([x1, fcm], 1).double() @ Wcat.double().t() + bcat.double() + fmf.double()
    assert (mu_a.double() - ref).abs().max().item() < 1e-4 and (mu_b.double() - ref).abs().max().item() < 1e-4


def test_step_helpers_through_the_c_abi(dev):
    """The entry points the one-node step added: batched transposes, list sums, and the two halves of the backward contractions
    run separately (as the step does on two streams) against the combined call -- bit for bit."""
    import ctypes
    import models
    from vml_amd._lib import call, ptr, stream
    lib = models.vml_amd._lib.load()
    g = torch.Generator().manual_seed(3)
    # smin_transpose_batch: odd shapes, more than one 32x32 tile, 33 matrices need two calls (SMIN_BATCH_MAX = 32)
    shapes = [(128, 512), (512, 128), (37, 5), (1, 70), (64, 64), (33, 129)] + [(8, 12)] * 26
    mats = [torch.randn(r, c, generator=g).to(dev) for r, c in shapes]
    outs = [torch.empty(c, r, device=dev) for r, c in shapes]
    arr = lambda ts: (ctypes.c_void_p * len(ts))(*[t.data_ptr() for t in ts])
    iarr = lambda v: (ctypes.c_int32 * len(v))(*v)
    call("smin_transpose_batch", stream(), arr(mats), arr(outs), iarr([s[0] for s in shapes]), iarr([s[1] for s in shapes]), len(mats))
    for m, o in zip(mats, outs):
        assert torch.equal(o, m.t().contiguous())
    assert lib.smin_transpose_batch(stream(), arr(mats + mats[:1]), arr(outs + outs[:1]), iarr([1] * 33), iarr([1] * 33), 33) < 0      # rejected, not truncated
    # smin_sum_lists: list order is the summation order
    for n, numel in [(2, 4096), (4, 1000), (7, 13), (3, 5 * 64 * 20)]:
        ts = [torch.randn(numel, generator=g).to(dev) for _ in range(n)]
        out = torch.empty(numel, device=dev)
        call("smin_sum_lists", stream(), arr(ts), n, numel, ptr(out))
        ref = ts[0].clone()
        for t in ts[1:]:
            ref = ref + t
        assert torch.equal(out, ref), (n, numel)
    # halves of the moment-unit backward and of the linear-rows backward
    B, L, D = 3, 12, 64
    mm = torch.triu(torch.ones(L, L, dtype=torch.bool)).unsqueeze(0).repeat(B, 1, 1)
    mm[2, :, 9:] = False
    lay = models.vml_amd.CellLayout.from_mask(mm.to(dev))
    N = lay.N
    r = lambda *s: torch.randn(*s, generator=g).to(dev)
    dmu, fcm, fb, x1, acc = r(N, D), r(N, D), r(B, L, D), r(N, D), r(N, D)
    WT = (r(2 * D, D) * 0.1).contiguous()
    nb = lib.smin_workspace_bytes(N, B, 4, D, 4, 1)
    ws = torch.empty(nb, dtype=torch.uint8, device=dev)

    def mu_bwd(want_in, want_w):
        dfc, dfb, dW, db = torch.full((N, D), 7.0, device=dev), torch.full((B, L, D), 7.0, device=dev), torch.full((D, 2 * D), 7.0, device=dev), torch.full((D,), 7.0, device=dev)
        call("smin_moment_unit_bwd", stream(), ptr(dmu), ptr(fcm), ptr(fb), ptr(lay.cells), ptr(lay.row_ptr), ptr(lay.cellmap), N, B, L, D, ptr(WT),
             ptr(dfc) if want_in else None, ptr(dfb) if want_in else None, ptr(dW) if want_w else None, ptr(db) if want_w else None, ptr(ws), ws.numel(), 1,
             ptr(acc) if want_in else None, ptr(x1), None)
        return dfc, dfb, dW, db
    full, ins, wts = mu_bwd(True, True), mu_bwd(True, False), mu_bwd(False, True)
    assert torch.equal(full[0], ins[0]) and torch.equal(full[1], ins[1]) and torch.equal(full[2], wts[2]) and torch.equal(full[3], wts[3])
    assert float(ins[2].min()) == 7.0 and float(wts[0].min()) == 7.0          # the skipped half's outputs are untouched
    # ... with another consumer's gradient of f_b accumulated into dfb (dfb_acc)
    dfb_acc = r(B, L, D)
    dfc2, dfb2 = torch.empty(N, D, device=dev), torch.empty(B, L, D, device=dev)
    call("smin_moment_unit_bwd", stream(), ptr(dmu), ptr(fcm), ptr(fb), ptr(lay.cells), ptr(lay.row_ptr), ptr(lay.cellmap), N, B, L, D, ptr(WT), ptr(dfc2), ptr(dfb2),
         None, None, ptr(ws), ws.numel(), 1, ptr(acc), ptr(x1), ptr(dfb_acc))
    assert torch.equal(dfc2, full[0]) and (dfb2 - (full[1] + dfb_acc)).abs().max().item() <= 1e-6 * max(1.0, float(full[1].abs().max()))
    # the two halves of the localization backward (map score / boundary heads) against the single call
    pm, psea, dpm, dpsea = torch.sigmoid(r(B, L, L)), torch.sigmoid(r(3, B, L)), r(B, L, L), r(3, B, L)
    fm_, wm, wb3, lmk = r(N, D), r(D), r(3, D), (torch.rand(B, L, generator=g) > 0.2).float().to(dev)

    def loc_bwd(map_half, heads_half):
        dfm_, dfb_, dwm, dbm, dwb, dbb = (torch.full(sh, 7.0, device=dev) for sh in ((N, D), (B, L, D), (D,), (1,), (3, D), (3,)))
        call("smin_score_map_bwd", stream(), ptr(dpm) if map_half else None, ptr(dpsea) if heads_half else None, ptr(pm), ptr(psea), ptr(fm_), ptr(fb), ptr(lay.cells),
             N, B, L, D, ptr(wm), ptr(wb3), ptr(lmk), ptr(dfm_) if map_half else None, ptr(dfb_) if heads_half else None, ptr(dwm) if map_half else None,
             ptr(dbm) if map_half else None, ptr(dwb) if heads_half else None, ptr(dbb) if heads_half else None, ptr(ws), ws.numel())
        return dfm_, dwm, dbm, dfb_, dwb, dbb
    both, mp, hd = loc_bwd(True, True), loc_bwd(True, False), loc_bwd(False, True)
    for q in range(3):
        assert torch.equal(both[q], mp[q]) and torch.equal(both[3 + q], hd[3 + q])
        assert float(mp[3 + q].min()) == 7.0 and float(hd[q].min()) == 7.0
    assert lib.smin_score_map_bwd(stream(), None, None, ptr(pm), ptr(psea), ptr(fm_), ptr(fb), ptr(lay.cells), N, B, L, D, ptr(wm), ptr(wb3), ptr(lmk), None, None, None,
                                  None, None, None, ptr(ws), ws.numel()) < 0                       # neither half
    R, K, O = 150, 32, 48
    dy, xa, xb = r(R, O), r(R, K), r(R, K)
    WTl = r(2 * K, O)
    nbl = lib.smin_linear_rows_bwd_workspace_bytes(R, O, 2 * K)
    wsl = torch.empty(nbl + 64, dtype=torch.uint8, device=dev)

    def lr_bwd(want_in, want_w):
        dxa, dxb, dW, db = torch.full((R, K), 7.0, device=dev), torch.full((R, K), 7.0, device=dev), torch.full((O, 2 * K), 7.0, device=dev), torch.full((O,), 7.0, device=dev)
        call("smin_linear_rows_bwd", stream(), ptr(dy), arr([xa, xb]), 2, ptr(WTl), R, O, K, arr([dxa, dxb]) if want_in else None, ptr(dW) if want_w else None,
             ptr(db) if want_w else None, ptr(wsl), wsl.numel())
        return dxa, dxb, dW, db
    full, ins, wts = lr_bwd(True, True), lr_bwd(True, False), lr_bwd(False, True)
    assert torch.equal(full[0], ins[0]) and torch.equal(full[1], ins[1]) and torch.equal(full[2], wts[2]) and torch.equal(full[3], wts[3])
    assert (full[2].double() - dy.double().t() @ torch.cat([xa, xb], 1).double()).abs().max().item() < 1e-3
    assert float(ins[2].min()) == 7.0 and float(wts[0].min()) == 7.0
    # smin_col_sum: one block, many blocks, a width that does not divide the workgroup, no rows
    for R_, W_ in [(5, 8), (300, 128), (70000, 128), (4099, 384), (1500, 1024), (0, 64)]:
        xs_ = r(R_, W_) if R_ else torch.empty(0, W_, device=dev)
        nb = lib.smin_col_sum_workspace_bytes(R_, W_)
        wsc = torch.empty(nb + 64, dtype=torch.uint8, device=dev)
        out = torch.full((W_,), 7.0, device=dev)
        call("smin_col_sum", stream(), ptr(xs_), R_, W_, ptr(out), ptr(wsc), wsc.numel())
        ref = xs_.double().sum(0)
        assert (out.double() - ref).abs().max().item() <= 1e-6 * max(1.0, float(xs_.abs().sum(0).max()) if R_ else 1.0), (R_, W_)
        out2 = torch.empty_like(out)
        call("smin_col_sum", stream(), ptr(xs_), R_, W_, ptr(out2), ptr(wsc), wsc.numel())
        assert torch.equal(out, out2)
    assert lib.smin_col_sum(stream(), ptr(r(4, 6)), 4, 6, ptr(torch.empty(6, device=dev)), None, 0) < 0           # W % 4
    # video encoder and LSTM layer backward: inputs half, then weights half on the same workspace (the step runs the second on its weight stream)
    B, T, Din, D = 3, 50, 24, 64
    df, fv, fs, x = r(B * T, D), r(B * T, D), r(B, D), r(B * T, Din)
    vm = (torch.rand(B * T, generator=g) > 0.2).float().to(dev)
    nbv = lib.smin_video_encoder_bwd_workspace_bytes(B, T, Din, D)
    # forward: the fused call against projection alone + smin_video_encoder_gate (the step runs the projection beside the query encoder)
    Wv, bv, pe = r(D, Din), r(D), r(T + 3, D)
    fv1, f1, fv2, f2 = (torch.full((B * T, D), 7.0, device=dev) for _ in range(4))
    call("smin_video_encoder_fwd", stream(), ptr(x), ptr(Wv), ptr(bv), ptr(pe), ptr(vm), ptr(fs), B, T, Din, D, ptr(fv1), ptr(f1))
    call("smin_video_encoder_fwd", stream(), ptr(x), ptr(Wv), ptr(bv), ptr(pe), ptr(vm), None, B, T, Din, D, ptr(fv2), None)
    call("smin_video_encoder_gate", stream(), ptr(fv2), ptr(fs), B, T, D, ptr(f2))
    assert torch.equal(fv1, fv2) and torch.equal(f1, f2)
    ref_fv = ((x.double() @ Wv.double().t() + bv.double()).view(B, T, D) + pe[:T].double()) * vm.view(B, T, 1).double()
    assert (fv1.view(B, T, D).double() - ref_fv).abs().max().item() < 1e-4
    assert lib.smin_video_encoder_fwd(stream(), ptr(x), ptr(Wv), ptr(bv), ptr(pe), ptr(vm), ptr(fs), B, T, Din, D, ptr(fv1), None) < 0     # fs without f

    def ve_bwd(split):
        ws = torch.zeros(nbv + 64, dtype=torch.uint8, device=dev)
        dW, db, dpe, dfs = torch.full((D, Din), 7.0, device=dev), torch.full((D,), 7.0, device=dev), torch.full((T, D), 7.0, device=dev), torch.full((B, D), 7.0, device=dev)
        if split:
            call("smin_video_encoder_bwd", stream(), ptr(df), ptr(fv), ptr(fs), ptr(vm), ptr(x), B, T, Din, D, None, None, None, ptr(dfs), ptr(ws), ws.numel())
            assert float(dW.min()) == 7.0 and float(dpe.min()) == 7.0
            call("smin_video_encoder_bwd", stream(), None, ptr(fv), ptr(fs), ptr(vm), ptr(x), B, T, Din, D, ptr(dW), ptr(db), ptr(dpe), None, ptr(ws), ws.numel())
        else:
            call("smin_video_encoder_bwd", stream(), ptr(df), ptr(fv), ptr(fs), ptr(vm), ptr(x), B, T, Din, D, ptr(dW), ptr(db), ptr(dpe), ptr(dfs), ptr(ws), ws.numel())
        return dW, db, dpe, dfs
    for a, b in zip(ve_bwd(False), ve_bwd(True)):
        assert torch.equal(a, b)
    dv = df.view(B, T, D) * fs.view(B, 1, D) * vm.view(B, T, 1)
    full = ve_bwd(False)
    assert (full[3].double() - (df * fv).view(B, T, D).double().sum(1)).abs().max().item() < 1e-4
    assert (full[0].double() - dv.view(B * T, D).double().t() @ x.double()).abs().max().item() < 1e-3
    assert (full[2].double() - dv.double().sum(0)).abs().max().item() < 1e-4
    wsn = torch.zeros(nbv + 64, dtype=torch.uint8, device=dev)
    assert lib.smin_video_encoder_bwd(stream(), None, ptr(fv), ptr(fs), ptr(vm), ptr(x), B, T, Din, D, None, None, None, None, ptr(wsn), wsn.numel()) < 0     # neither half
    B, Nq, In, Hh = 5, 7, 24, 32
    xq, dHo = r(B, Nq, In), r(B, Nq, 2 * Hh)
    Wih, bias, Whh = r(8 * Hh, In) * 0.3, r(8 * Hh) * 0.1, r(2, 4 * Hh, Hh) * 0.3
    W4 = Whh.view(2, 4, Hh, Hh).permute(0, 3, 2, 1).contiguous()
    length = torch.tensor([7, 3, 1, 5, 7], dtype=torch.int32, device=dev)
    G, Ho, Cs = torch.empty(B, Nq, 2, 4 * Hh, device=dev), torch.empty(B, Nq, 2 * Hh, device=dev), torch.empty(B, Nq, 2, Hh, device=dev)
    call("smin_bilstm_layer_fwd", stream(), ptr(xq), ptr(Wih), ptr(bias), ptr(W4), ptr(length), B, Nq, In, Hh, ptr(G), ptr(Ho), ptr(Cs))
    WihT = Wih.t().contiguous()
    nbl2 = lib.smin_bilstm_layer_bwd_workspace_bytes(B, Nq, In, Hh)

    def lstm_bwd(split):
        ws = torch.zeros(nbl2 + 64, dtype=torch.uint8, device=dev)
        dX, dWih, dbi, dWhh = torch.full((B, Nq, In), 7.0, device=dev), torch.full((8 * Hh, In), 7.0, device=dev), torch.full((8 * Hh,), 7.0, device=dev), torch.full((2, 4 * Hh, Hh), 7.0, device=dev)
        args = (ptr(xq), ptr(Ho), ptr(G), ptr(Cs), ptr(WihT), ptr(Whh), ptr(length), B, Nq, In, Hh)
        if split == "pieces":                                    # inputs half, then the weights half piece by piece in another order
            call("smin_bilstm_layer_bwd", stream(), ptr(dHo), *args, ptr(dX), None, None, None, ptr(ws), ws.numel())
            dbi2 = torch.full_like(dbi, 7.0)
            for which in (4, 1, 2):
                call("smin_bilstm_layer_bwd_weights", stream(), which, ptr(xq), ptr(Ho), B, Nq, In, Hh, ptr(dWih), ptr(dbi), ptr(dbi2), ptr(dWhh), ptr(ws), ws.numel())
            assert torch.equal(dbi, dbi2)
        elif split:
            call("smin_bilstm_layer_bwd", stream(), ptr(dHo), *args, ptr(dX), None, None, None, ptr(ws), ws.numel())
            assert float(dWih.min()) == 7.0 and float(dWhh.min()) == 7.0
            call("smin_bilstm_layer_bwd", stream(), None, *args, None, ptr(dWih), ptr(dbi), ptr(dWhh), ptr(ws), ws.numel())
        else:
            call("smin_bilstm_layer_bwd", stream(), ptr(dHo), *args, ptr(dX), ptr(dWih), ptr(dbi), ptr(dWhh), ptr(ws), ws.numel())
        return dX, dWih, dbi, dWhh
    for a, b, c in zip(lstm_bwd(False), lstm_bwd(True), lstm_bwd("pieces")):
        assert torch.equal(a, b) and torch.equal(a, c)
    # operand layouts: every layer in one launch (smin_lstm_pack_layers) against a launch per layer (smin_lstm_pack)
    import ctypes
    ins, raws, one, many = [24, 2 * Hh], [], [], []
    for In_ in ins:
        raws.append([r(4 * Hh, In_), r(4 * Hh, Hh), r(4 * Hh), r(4 * Hh), r(4 * Hh, In_), r(4 * Hh, Hh), r(4 * Hh), r(4 * Hh)])
        one.append([torch.full(sh, 7.0, device=dev) for sh in ((8 * Hh, In_), (8 * Hh,), (2, 4 * Hh, Hh), (2, Hh, Hh, 4))])
        many.append([torch.full_like(t, 9.0) for t in one[-1]])
        call("smin_lstm_pack", stream(), (ctypes.c_void_p * 8)(*[t.data_ptr() for t in raws[-1]]), In_, Hh, *[ptr(t) for t in one[-1]])
    call("smin_lstm_pack_layers", stream(), 2, (ctypes.c_void_p * 16)(*[t.data_ptr() for l in raws for t in l]), (ctypes.c_int * 2)(*ins), Hh,
         *[(ctypes.c_void_p * 2)(many[0][q].data_ptr(), many[1][q].data_ptr()) for q in range(4)])
    for l in range(2):
        for a, b in zip(one[l], many[l]):
            assert torch.equal(a, b)
        assert torch.equal(one[l][1], torch.cat([raws[l][2] + raws[l][3], raws[l][6] + raws[l][7]]))


@pytest.mark.parametrize("cut", [False, True])
def test_short_query_batch_is_padded_to_max_query_length(dev, cut):
    """A batch whose longest query has fewer words than max_query_length (the reference pads f_w to max_query_length, models.py:58-59):
    the one-node step (library kernels for the sentence feature and its gradient) against the Python host.  cut=False is the reference's
    case (the dataset pads tokens and mask to max_query_length, dataset.py:35, 173); cut=True hands over the batch cut to its longest query,
    which both hosts pad themselves."""
    from oracle import smin_oracle as O
    from vml_amd import loss_fn
    T, L, C, D, dl, layers, Din, Nq, Hh, B = 64, 16, 4, 128, 32, 3, 40, 12, 64, 4
    sd = O.formula_state_dict(H.smin_shapes(T, L, C, D, dl, layers, Din, Nq, Hh), gain=1.2)
    batch = O.synthetic_batch(B, T, L, Nq, Din, seed=8)
    short = 8
    batch["query_mask"][:, short:] = 0
    batch["query_mask"][0, :short] = 1                          # one query of full (short) length, the others as generated
    if cut:
        batch["query_features"] = batch["query_features"][:, :short].contiguous()
        batch["query_mask"] = batch["query_mask"][:, :short].contiguous()
    b = {k: v.to(dev) for k, v in batch.items()}
    res = []
    for native in (True, False):
        m = build_model(dict(T=T, L=L, C=C, D=D, dl=dl, layers=layers, Din=Din, Nq=Nq, H=Hh), sd, dev)
        m.native_host = native
        out = m(*H.model_inputs(b))
        loss_fn(out[0], b["ym"], b["sm"], b["moment_mask"], out[1], b["ys"], b["ss"], out[2], b["ye"], b["se"], out[3], b["ya"], b["length_mask"]).backward()
        res.append(([o.detach().clone() for o in out], {k: p.grad.clone() for k, p in m.named_parameters()}))
    (o0, g0), (o1, g1) = res
    for x, y in zip(o0, o1):
        assert float((x - y).abs().max()) <= 1e-5             # (the two hosts form the weight products with different kernels)
    if cut:                                                   # the oracle (like the reference) wants the mask as wide as the padded f_w
        batch["query_features"] = torch.nn.functional.pad(batch["query_features"], (0, 0, 0, Nq - short))
        batch["query_mask"] = torch.nn.functional.pad(batch["query_mask"], (0, 0, 0, Nq - short))
    ref = O.smin_forward({k: v.clone() for k, v in sd.items()}, dict(T=T, L=L, C=C), *H.model_inputs(batch))
    for x, y in zip(o0, ref):
        assert (x.cpu() - y).abs().max().item() < SCORE_TOL
    gmax = max(float(v.abs().max()) for v in g1.values())
    for k in g1:
        assert float((g0[k] - g1[k]).abs().max()) <= 2e-5 * float(g1[k].abs().max()) + 3e-6 * gmax + 1e-9, k     # (some gradients are zero in exact arithmetic)


@pytest.mark.parametrize("nl,D,dl", [(3, 512, 128), (1, 64, 32), (6, 96, 32), (8, 64, 64)])
def test_param_prep_kernels(dev, nl, D, dl):
    """csrc/param_prep.hip (every layer's parameter-only operands of the content stream and the moment unit in one launch per
    direction) against the torch formulas it replaces and their autograd gradients, in fp64."""
    import ctypes
    import models
    from vml_amd._lib import call, ptr, stream
    g = torch.Generator().manual_seed(nl * 1000 + D + dl)
    r = lambda *s: torch.randn(*s, generator=g)
    P = [[r(dl, D) * 0.1, r(dl), r(D, dl) * 0.1, r(D), r(D, D) * 0.1, r(D), r(D, D) * 0.1, r(D)] for _ in range(nl)]
    Pd = [[t.to(dev) for t in row] for row in P]
    arr = lambda ts: (ctypes.c_void_p * len(ts))(*[(t.data_ptr() if t is not None else None) for t in ts])
    nseg = lambda k, part: min(4, k - 4 * part)
    Pcat = [[(torch.empty(dl, nseg(k, part) * dl, device=dev) if k > 4 * part else None) for part in range(2)] for k in range(nl)]
    consts, Wcat, bcat, Wch_all = torch.empty(nl, dl, device=dev), torch.empty(nl, D, 2 * D, device=dev), torch.empty(nl, D, device=dev), torch.empty(nl * dl, D, device=dev)
    call("smin_param_prep_fwd", stream(), arr([t for row in Pd for t in row]), nl, D, dl, arr([t for row in Pcat for t in row]), ptr(consts), ptr(Wcat), ptr(bcat),
         ptr(Wch_all))
    # reference (fp64, autograd)
    R = [[t.double().requires_grad_(True) for t in row] for row in P]
    ref_P, ref_c = [], []
    bsum = None
    for k in range(nl):
        ref_c.append(R[k][1] + (R[k][0] @ bsum if bsum is not None else 0))
        bsum = R[k][3] if bsum is None else bsum + R[k][3]
        ref_P.append([torch.cat([R[k][0] @ R[l][2] for l in range(4 * part, min(4 * part + 4, k))], 1) if k > 4 * part else None for part in range(2)])
    ref_Wcat = torch.stack([torch.cat([R[k][4], R[k][6]], 1) for k in range(nl)])
    ref_bcat = torch.stack([R[k][5] + R[k][7] for k in range(nl)])
    ref_all = torch.cat([R[k][0] for k in range(nl)])
    tol = lambda ref: 2e-5 * max(1.0, float(ref.detach().abs().max()))
    for k in range(nl):
        assert (consts[k].cpu().double() - ref_c[k].detach()).abs().max().item() < tol(ref_c[k])
        for part in range(2):
            if Pcat[k][part] is not None:
                assert (Pcat[k][part].cpu().double() - ref_P[k][part].detach()).abs().max().item() < tol(ref_P[k][part]), (k, part)
    assert torch.equal(Wcat.cpu().double(), ref_Wcat.detach()) and torch.equal(Wch_all.cpu().double(), ref_all.detach())
    assert (bcat.cpu().double() - ref_bcat.detach()).abs().max().item() < 1e-6
    # backward
    dP = [[(r(*Pcat[k][part].shape) if Pcat[k][part] is not None else None) for part in range(2)] for k in range(nl)]
    dconsts, dWcat, dbcat, dall = r(nl, dl), r(nl, D, 2 * D), r(nl, D), r(nl * dl, D)
    base_ch = [(r(dl, D) if k > 0 else None) for k in range(nl)]
    base_c, base_bc = [r(D, dl) for _ in range(nl)], [r(D) for _ in range(nl)]
    loss = (ref_Wcat * dWcat.double()).sum() + (ref_bcat * dbcat.double()).sum() + (ref_all * dall.double()).sum()
    for k in range(nl):
        loss = loss + (ref_c[k] * dconsts[k].double()).sum() + (R[k][2] * base_c[k].double()).sum() + (R[k][3] * base_bc[k].double()).sum()
        if base_ch[k] is not None:
            loss = loss + (R[k][0] * base_ch[k].double()).sum()
        for part in range(2):
            if dP[k][part] is not None:
                loss = loss + (ref_P[k][part] * dP[k][part].double()).sum()
    loss.backward()
    grads = [[torch.full_like(t, 7.0) for t in row] for row in Pd]
    dv = lambda t: None if t is None else t.to(dev)
    dPd = [[dv(t) for t in row] for row in dP]
    bch_d, bc_d, bbc_d = [dv(t) for t in base_ch], [dv(t) for t in base_c], [dv(t) for t in base_bc]
    dcd, dWd, dbd, dad = dconsts.to(dev), dWcat.to(dev), dbcat.to(dev), dall.to(dev)
    call("smin_param_prep_bwd", stream(), arr([t for row in Pd for t in row]), nl, D, dl, arr([t for row in dPd for t in row]), ptr(dcd), ptr(dWd), ptr(dbd), ptr(dad),
         arr(bch_d), arr(bc_d), arr(bbc_d), arr([t for row in grads for t in row]))
    for k in range(nl):
        for q in range(8):
            ref = R[k][q].grad
            err = (grads[k][q].cpu().double() - ref).abs().max().item()
            assert err < 5e-5 * max(1.0, float(ref.abs().max())), (k, q, err)


def test_degenerate_samples_and_empty_batch(dev):
    """A sample with a single valid snippet, a sample with none, inference under no_grad, and a batch without any valid
    cell (N = 0) with its backward -- the edge cases of the packed layout."""
    import models
    from oracle import smin_oracle as O
    T, L, C, D, dl, layers, Din, Nq, Hh = 32, 8, 4, 64, 32, 3, 24, 6, 32
    sd = O.formula_state_dict(H.smin_shapes(T, L, C, D, dl, layers, Din, Nq, Hh), gain=1.2)
    m = build_model(dict(T=T, L=L, C=C, D=D, dl=dl, layers=layers, Din=Din, Nq=Nq, H=Hh), sd, dev)
    batch = O.synthetic_batch(3, T, L, Nq, Din, seed=5)
    batch["length_mask"][1] = False
    batch["length_mask"][1, 0] = True
    batch["length_mask"][2] = False
    batch["moment_mask"] = torch.triu(batch["length_mask"].unsqueeze(2) & batch["length_mask"].unsqueeze(1))
    batch["video_mask"][2] = 0
    batch["video_features"][2] = 0
    ref = O.smin_forward({k: v.clone() for k, v in sd.items()}, dict(T=T, L=L, C=C), *H.model_inputs(batch))
    with torch.no_grad():
        out = m(*H.model_inputs(batch, dev))
    for a, b in zip(out, ref):
        assert (a.cpu() - b).abs().max().item() < SCORE_TOL
    batch["length_mask"][:] = False
    batch["moment_mask"][:] = False
    out = m(*H.model_inputs(batch, dev))
    assert all(float(o.detach().abs().max()) == 0.0 for o in out)
    (out[0].sum() + out[1].sum()).backward()
    assert all(p.grad is None or bool(torch.isfinite(p.grad).all()) for p in m.parameters())


def test_bench_contract_line(dev):
    """bench.py prints one JSON line with the keys the driver reads (short run of a small workload)."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "2", "--warmup", "1", "--no-cpu-baseline",
                          "--workload", "charadessta"], capture_output=True, text=True, timeout=300, cwd=root)
    assert out.returncode == 0, out.stderr[-2000:]
    line = [ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1]
    j = json.loads(line)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline"):
        assert k in j, k
    default_mode = os.environ.get("SMIN_GEMM_MODE", "f32")
    assert j["n_gpus"] == 1 and j["steps"] == 2 and j["dtype"].startswith("f32" if default_mode != "bf16" else "bf16") and j["higher_is_better"] is True
    assert set(("bound", "achieved", "peak", "unit", "frac", "traffic")) <= set(j["roofline"])
    assert "workload" in j["config"] and j["value"] > 0
    # the bf16-core contraction modes are timed beside the headline, never as it; the emulated-fp32 scores sit on the exact ones
    if default_mode == "f32":
        oa = j["other_arithmetic"]
        assert set(oa) == {"f32e", "bf16x3"} and oa["f32e"]["max_score_deviation_from_f32"] < 5e-6 and oa["bf16x3"]["max_score_deviation_from_f32"] < 1e-4


def test_layout_build_kernels(dev):
    """smin_build_cells (two launches) against the torch formulation of the packed layout: mask-driven and all-cells
    lists, ragged / empty rows, L not a multiple of 64, more than 1024 rows."""
    import models
    CellLayout = models.vml_amd.CellLayout
    g = torch.Generator().manual_seed(9)
    for (B, L) in [(3, 6), (5, 70), (40, 33), (2, 130)]:
        mm = torch.rand(B, L, L, generator=g) > 0.6
        mm[0, 1] = False                                              # an empty row
        if B > 2:
            mm[2] = False                                             # an empty sample
        ref = CellLayout.from_mask(mm)                                # CPU tensors: torch formulation
        got = CellLayout.begin(mm.to(dev)).finish()                   # HIP kernels (count copied asynchronously)
        assert got.N == ref.N and got.all_valid
        for name in ("cells", "row_ptr", "cellmap"):
            assert torch.equal(getattr(got, name).cpu(), getattr(ref, name)), (B, L, name)
        assert torch.equal(got.bidx.cpu(), ref.bidx) and torch.equal(got.jidx.cpu(), ref.jidx)
        ref_all, got_all = CellLayout.all_cells(mm), CellLayout.all_cells(mm.to(dev))
        assert got_all.N == B * L * L and not got_all.all_valid
        for name in ("cells", "row_ptr", "cellmap"):
            assert torch.equal(getattr(got_all, name).cpu(), getattr(ref_all, name)), (B, L, name, "all")


def test_step_prologue_kernel(dev):
    """smin_step_prologue (the masks as fp32, query lengths, valid-cell count, the boundary heads' parameters side by side: one
    launch) against the torch formulation, twice in a row (the counting words must come back to zero), ragged sizes."""
    import models
    from vml_amd._lib import call, ptr, stream
    g = torch.Generator().manual_seed(21)
    acc = torch.zeros(2, dtype=torch.int64, device=dev)
    for (B, Nq, T, L, D) in [(3, 5, 7, 6, 12), (64, 20, 256, 64, 512), (2, 1, 1030, 130, 4)]:
        qm = (torch.rand(B, Nq, generator=g) > 0.3).to(torch.uint8).to(dev)
        vm = (torch.rand(B, T, 1, generator=g) > 0.3).to(torch.uint8).to(dev)
        lm = (torch.rand(B, L, generator=g) > 0.3).to(dev)
        mm = (torch.rand(B, L, L, generator=g) > 0.6).to(dev)
        w = [torch.randn(D, generator=g).to(dev) for _ in range(3)]
        b = [torch.randn(1, generator=g).to(dev) for _ in range(3)]
        for rep in range(2):
            len32 = torch.full((B,), -1, dtype=torch.int32, device=dev)
            qmf, vmf, lmf = torch.full((B, Nq), 7.0, device=dev), torch.full((B * T,), 7.0, device=dev), torch.full((B, L), 7.0, device=dev)
            wb, bb, count = torch.full((3, D), 7.0, device=dev), torch.full((3,), 7.0, device=dev), torch.full((1,), -1, dtype=torch.int64, device=dev)
            import ctypes
            arr = lambda ts: (ctypes.c_void_p * 3)(*[t.data_ptr() for t in ts])
            call("smin_step_prologue", stream(), ptr(qm), ptr(vm), ptr(lm), ptr(mm), arr(w), arr(b), B, Nq, T, L, D, ptr(len32), ptr(qmf), ptr(vmf), ptr(lmf), ptr(wb),
                 ptr(bb), ptr(count), ptr(acc))
            assert torch.equal(len32, qm.sum(1).to(torch.int32)) and int(count) == int(mm.sum())
            assert torch.equal(qmf, qm.float()) and torch.equal(vmf, vm.reshape(-1).float()) and torch.equal(lmf, lm.float())
            assert torch.equal(wb, torch.stack(w)) and torch.equal(bb, torch.cat(b))
            assert int(acc.abs().sum()) == 0


# ---------------------------------------------------------------- the captured step (training.CapturedStep)
def test_captured_step_bit_identical_to_eager(dev):
    """One HIP graph per (shapes, valid-cell count): scores, loss and every gradient of a replay equal the eager step's bit for bit,
    a second batch with the same count replays the same graph, a batch with another count gets its own, and a wrong
    known_cell_count is flagged instead of writing out of bounds."""
    import models
    from oracle import smin_oracle as O                         # seeded inputs and weights only
    from vml_amd import CapturedStep, loss_fn
    T, L, C, D, dl, layers, Din, Nq, Hh, B = 64, 16, 4, 128, 32, 3, 40, 9, 64, 4
    sd = O.formula_state_dict(H.smin_shapes(T, L, C, D, dl, layers, Din, Nq, Hh), gain=1.2)
    m = build_model(dict(T=T, L=L, C=C, D=D, dl=dl, layers=layers, Din=Din, Nq=Nq, H=Hh), sd, dev)

    def eager(b):
        for p in m.parameters():
            p.grad = None
        out = m(*H.model_inputs(b))
        loss = loss_fn(out[0], b["ym"], b["sm"], b["moment_mask"], out[1], b["ys"], b["ss"], out[2], b["ye"], b["se"], out[3], b["ya"], b["length_mask"])
        loss.backward()
        return loss.detach().clone(), [o.detach().clone() for o in out], [p.grad.clone() for p in m.parameters()]

    b1 = {k: v.to(dev) for k, v in O.synthetic_batch(B, T, L, Nq, Din, seed=11).items()}
    # a second batch with the same masks (hence the same count) and other values
    b2 = dict(b1)
    g = torch.Generator().manual_seed(5)
    b2["video_features"] = (torch.randn(b1["video_features"].shape, generator=g).to(dev) * b1["video_mask"].float())
    b2["sm"] = (torch.rand(b1["sm"].shape, generator=g).to(dev) * b1["moment_mask"])
    b3 = {k: v.to(dev) for k, v in O.synthetic_batch(B, T, L, Nq, Din, seed=12).items()}     # other lengths: another count
    assert int(b3["moment_mask"].sum()) != int(b1["moment_mask"].sum())
    step = CapturedStep(m, None)
    for b in (b1, b2, b3, b1):
        want = eager(b)
        loss, out = step(b)
        torch.cuda.synchronize()
        assert torch.equal(loss.detach(), want[0])
        for a, w in zip(out, want[1]):
            assert torch.equal(a.detach(), w)
        for p, w in zip(m.parameters(), want[2]):
            assert torch.equal(p.grad, w)
    assert len(step.entries) == 2                                  # b1 / b2 share a graph
    # a wrong count: flagged, in bounds
    ops = models.vml_amd._lib.load_torch()
    status = ops.layout_status(dev)
    m.known_cell_count = int(b1["moment_mask"].sum()) + 7
    try:
        m(*H.model_inputs(b1))
        torch.cuda.synchronize()
        assert int(status.item()) == 1
    finally:
        m.known_cell_count = None
        status.zero_()


# ---------------------------------------------------------------- the two attention classes on their own
def test_standalone_attention_classes_against_oracle(dev):
    """Attention (reference models.py:128-154) and ContentAttention (models.py:198-226) as stand-alone modules: outputs and the
    gradients of inputs and parameters against the oracle's word attention on the same seeded inputs, with a partly masked
    query (including B = 1, where the reference's mask.squeeze() mis-shapes)."""
    import models
    from oracle import smin_oracle as O
    g = torch.Generator().manual_seed(21)
    for B in (3, 1):
        L, C, D, Nq = 6, 4, 32, 7
        qmask = torch.ones(B, Nq, 1, dtype=torch.uint8)
        qmask[B - 1, 4:] = 0
        key = torch.randn(B, Nq, D, generator=g) * qmask.float()
        for cls, query in ((models.Attention, torch.randn(B, L, D, generator=g)), (models.ContentAttention, torch.randn(B, L, L, C, D, generator=g))):
            mod = cls(D)
            sd = {"x." + k: v.detach().clone().requires_grad_(True) for k, v in mod.state_dict().items()}
            q0, k0 = query.clone().requires_grad_(True), key.clone().requires_grad_(True)
            want = O._word_attention(sd, "x.", q0.reshape(B, -1, D), k0, k0, qmask.reshape(B, 1, Nq).float(), D).reshape(query.shape)
            w = torch.randn(want.shape, generator=g)
            (want * w).sum().backward()
            mod = mod.to(dev)
            q1, k1 = query.to(dev).requires_grad_(True), key.to(dev).requires_grad_(True)
            got = mod(q1, k1, k1, qmask.to(dev))
            (got * w.to(dev)).sum().backward()
            assert rel_err(got.detach().cpu(), want.detach()) < 1e-5, cls.__name__
            eq, ek = rel_err(q1.grad.cpu(), q0.grad), rel_err(k1.grad.cpu(), k0.grad)
            assert eq < 1e-4 and ek < 1e-4, (cls.__name__, B, eq, ek)
            # (W_k.bias shifts every score of a row equally: softmax-invariant, its true gradient is zero and what is computed is
            #  rounding noise -- held to the scale of the whole gradient)
            gmax = max(sd["x." + name].grad.abs().max().item() for name, _ in mod.named_parameters())
            for name, p in mod.named_parameters():
                ref = sd["x." + name].grad
                assert (p.grad.cpu() - ref).abs().max().item() <= 1e-4 * ref.abs().max().item() + 1e-5 * gmax, (cls.__name__, name)


def test_captured_step_with_optimizer_matches_eager_training(dev):
    """The captured step including the (capturable, fused) Adam update: after its three warm-up steps and one replay the
    parameters equal those of four eager steps on the same batch, bit for bit; further replays keep matching."""
    import models
    from oracle import smin_oracle as O
    from vml_amd import CapturedStep, loss_fn
    T, L, C, D, dl, layers, Din, Nq, Hh, B = 32, 8, 4, 64, 32, 2, 24, 6, 32, 3
    sd = O.formula_state_dict(H.smin_shapes(T, L, C, D, dl, layers, Din, Nq, Hh), gain=1.2)
    b = {k: v.to(dev) for k, v in O.synthetic_batch(B, T, L, Nq, Din, seed=21).items()}
    cfg = dict(T=T, L=L, C=C, D=D, dl=dl, layers=layers, Din=Din, Nq=Nq, H=Hh)
    m_e, m_c = build_model(cfg, sd, dev), build_model(cfg, sd, dev)
    opt_e = torch.optim.Adam(m_e.parameters(), lr=1e-3, fused=True, capturable=True)
    opt_c = torch.optim.Adam(m_c.parameters(), lr=1e-3, fused=True, capturable=True)

    def eager_step():
        opt_e.zero_grad(set_to_none=True)
        out = m_e(*H.model_inputs(b))
        loss = loss_fn(out[0], b["ym"], b["sm"], b["moment_mask"], out[1], b["ys"], b["ss"], out[2], b["ye"], b["se"], out[3], b["ya"], b["length_mask"])
        loss.backward()
        opt_e.step()
        return loss.detach().clone()

    step = CapturedStep(m_c, opt_c)
    for it in range(3):
        loss_c, _ = step(b)                                         # first call: three eager warm-up steps + capture + one replay
        n_eager = 4 if it == 0 else 1
        for _ in range(n_eager):
            loss_e = eager_step()
        torch.cuda.synchronize()
        assert torch.equal(loss_c.detach(), loss_e), it
        for (k, pe), (_, pc) in zip(m_e.named_parameters(), m_c.named_parameters()):
            assert torch.equal(pe, pc), (it, k)
