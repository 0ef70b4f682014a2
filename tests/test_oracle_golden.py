"""CPU: the oracle (oracle/smin_oracle.py) must reproduce every golden fixture that
tests/golden/make_golden.py produced from the reference implementation."""
import numpy as np
import pytest
import torch

from oracle import smin_oracle as O
from tests import helpers as H


@pytest.mark.parametrize("name", H.TINY)
def test_tiny_forward_seams_and_grads(name):
    cfg, sd, batch, out, grads, loss_ref = H.split_tiny(H.load_npz(name))
    sd = {k: v.clone().requires_grad_(v.dtype.is_floating_point) for k, v in sd.items()}
    (pm, ps, pe, pa), seams = O.smin_forward(sd, cfg, *H.model_inputs(batch), return_seams=True)
    seams.update(pm=pm, ps=ps, pe=pe, pa=pa)
    for k, ref in out.items():
        assert seams[k].shape == ref.shape, k
        err = (seams[k].detach() - ref).abs().max().item()
        assert err <= 2e-6 * max(1.0, ref.abs().max().item()), (k, err)
    loss = O.loss_fn(pm, batch["ym"], batch["sm"], batch["moment_mask"], ps, batch["ys"], batch["ss"],
                     pe, batch["ye"], batch["se"], pa, batch["ya"], batch["length_mask"])
    assert abs(loss.item() - loss_ref) <= 1e-5
    loss.backward()
    for k, g in grads.items():
        got = sd[k].grad
        scale = max(g.abs().max().item(), 1e-6)
        assert (got - g).abs().max().item() <= 2e-4 * scale + 1e-7, k


def test_structural_invariant_masked_cells_are_zero():
    """masked cells stay exactly 0 through every layer (basis of the packed layout)."""
    cfg, sd, batch, out, _, _ = H.split_tiny(H.load_npz("g2_r2_ragged"))
    mm = batch["moment_mask"]
    lm = batch["length_mask"]
    for k in range(cfg["layers"]):
        assert out[f"cu{k}"][~mm].abs().max() == 0
        assert out[f"mu{k}"][~mm].abs().max() == 0
        assert out[f"bu{k}"][~lm].abs().max() == 0
    assert out["pm"][~mm].abs().max() == 0


def test_content_windows_match_reference_matrix():
    z = H.load_npz("g4_wc")
    for key in [k[6:] for k in z.files if k.startswith("start/")]:
        T, L, C = (int(v) for v in key.split("_"))
        start, size = O.content_windows(T, L, C)
        assert np.array_equal(size.numpy(), z["size/" + key]), key
        assert np.array_equal(np.where(z["size/" + key] > 0, start.numpy(), 0), z["start/" + key]), key
        val = torch.where(size > 0, 1.0 / size.clamp(min=1).float(), torch.zeros(1))
        assert np.array_equal(val.numpy(), z["val/" + key]), key
        Wc = O.content_matrix(T, L, C)
        assert Wc.shape == (L, L, C, T)
        rows = Wc.sum(-1)
        assert torch.all(((rows - 1).abs() < 1e-6) | (rows == 0))


@pytest.mark.parametrize("name", ["tacos_d500", "charades", "anet_yml"])
def test_full_size_outputs(name):
    z = H.load_npz("g5_" + name)
    T, L, C, D, dl, layers, Din, Nq, Hh = H.FULL[name]
    B, seed = int(z["cfg"][-2]), int(z["cfg"][-1])
    sd = O.formula_state_dict(H.smin_shapes(T, L, C, D, dl, layers, Din, Nq, Hh), gain=1.3)
    assert abs(float(sum(v.double().abs().sum() for v in sd.values())) - float(z["sd_checksum"][0])) < 1e-6 * float(z["sd_checksum"][0])
    batch = O.synthetic_batch(B, T, L, Nq, Din, seed=seed)
    chk = z["in_checksum"]
    assert abs(float(batch["video_features"].double().sum()) - chk[0]) < 1e-6 * max(1.0, abs(chk[0]))
    assert float(batch["moment_mask"].sum()) == chk[2]
    with torch.no_grad():
        pm, ps, pe, pa = O.smin_forward(sd, dict(T=T, L=L, C=C), *H.model_inputs(batch))
    for k, got in dict(pm=pm, ps=ps, pe=pe, pa=pa).items():
        assert (got - torch.from_numpy(z["out/" + k])).abs().max().item() <= 1e-5, k


def test_compute_ious_fixture():
    z = H.load_npz("g6_ious")
    got = O.compute_ious(*(torch.from_numpy(z[k]) for k in ("pm", "ps", "pe", "mm", "sm")))
    for k, v in zip(z["keys"], z["vals"]):
        assert got[str(k)] == float(v)


def test_oracle_fp64_noise_floor():
    """fp32 oracle vs fp64 oracle: the reference's own rounding noise (~1e-7)."""
    cfg, sd, batch, out, _, _ = H.split_tiny(H.load_npz("g1_r4"))
    sd64 = {k: v.double() if v.dtype.is_floating_point else v for k, v in sd.items()}
    xs = H.model_inputs(batch)
    xs64 = [x.double() if x.dtype == torch.float32 else x for x in xs]
    with torch.no_grad():
        pm64 = O.smin_forward(sd64, cfg, *xs64)[0]
    assert (pm64.float() - out["pm"]).abs().max().item() < 5e-6
