"""Generate the golden fixtures under tests/golden/ from the REFERENCE implementation.

Run only in the build container (``/root/reference`` is read-only and absent on
the GPU box):  ``python -B tests/golden/make_golden.py``

The reference's ``models.py`` (imports only math/torch) and ``utils.py`` are
imported as-is on CPU; nothing from them is copied into this repo -- only
inputs and expected outputs are stored.  The training loss in the reference's
``main.py`` cannot run (``reduction=None`` raises) and ``main.py``
cannot be imported offline (torchtext), so gradients are produced by applying
the restated loss ``oracle.smin_oracle.loss_fn`` to the *reference model's*
outputs.

Fixtures:
  g1_r4.npz / g2_r2_ragged.npz / g3_c3.npz   tiny configs: state_dict, inputs,
        every seam tensor, outputs, loss, all parameter grads.
  g4_wc.npz          compute_content_matrix structure for three (T, L, C).
  g5_<cfg>.npz       full-size BASELINE configs at B=2 with formula-initialised
        weights: outputs, loss, grad norms (inputs/weights are regenerated from
        closed forms + seed; checksums stored).
  g6_ious.npz        utils.compute_ious on a fixed input.
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.dont_write_bytecode = True
sys.path.insert(0, "/root/reference")

import models as ref_models          # noqa: E402  (the reference)
import utils as ref_utils            # noqa: E402
from oracle import smin_oracle as O  # noqa: E402

torch.set_num_threads(8)


def np_sd(sd):
    return {"sd/" + k: v.detach().cpu().numpy() for k, v in sd.items()}


def run_reference(model, batch, seams=False):
    out = {}
    f, fs, fw = model.backbone(batch["video_features"], batch["video_mask"], batch["query_features"], batch["query_mask"])
    fc, fm, fb = model.pgm(f, batch["moment_mask"])
    if seams:
        out.update(f=f, fs=fs, fw=fw, fc=fc, fm=fm, fb=fb)
    for k, layer in enumerate(model.smis):
        fc, fm, fb = layer(fc, fm, fb, fw, fs, batch["query_mask"], batch["length_mask"], batch["moment_mask"])
        if seams:
            out[f"cu{k}"], out[f"mu{k}"], out[f"bu{k}"] = fc, fm, fb
    pm, ps, pe, pa = model.localization(fm, fb, batch["length_mask"], batch["moment_mask"])
    # cross-check the staged call against the reference's own forward()
    pm2, ps2, pe2, pa2 = model(batch["video_features"], batch["video_mask"], batch["query_features"],
                               batch["query_mask"], batch["length_mask"], batch["moment_mask"])
    assert torch.equal(pm, pm2) and torch.equal(ps, ps2) and torch.equal(pe, pe2) and torch.equal(pa, pa2)
    out.update(pm=pm, ps=ps, pe=pe, pa=pa)
    return out


def loss_of(out, batch):
    return O.loss_fn(out["pm"], batch["ym"], batch["sm"], batch["moment_mask"], out["ps"], batch["ys"], batch["ss"],
                     out["pe"], batch["ye"], batch["se"], out["pa"], batch["ya"], batch["length_mask"])


def tiny_case(name, T, L, C, D, dl, layers, Din, Nq, H, B, seed, boost):
    torch.manual_seed(seed)
    model = ref_models.SMIN(T, L, C, D, dl, layers, Din, Nq, H)
    if boost != 1.0:   # push activations away from the untrained 0.5 plateau
        with torch.no_grad():
            for n_, p in model.named_parameters():
                if "lstm" not in n_:
                    p.mul_(boost)
    batch = O.synthetic_batch(B, T, L, Nq, Din, seed=seed + 100)
    out = run_reference(model, batch, seams=True)
    loss = loss_of(out, batch)
    loss.backward()
    blob = {"cfg": np.array([T, L, C, D, dl, layers, Din, Nq, H, B], dtype=np.int64)}
    blob.update(np_sd(model.state_dict()))
    blob.update({"in/" + k: v.numpy() for k, v in batch.items()})
    blob.update({"out/" + k: v.detach().numpy() for k, v in out.items()})
    blob["loss"] = loss.detach().numpy()
    blob.update({"grad/" + k: p.grad.numpy() for k, p in model.named_parameters()})
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **blob)
    print(name, "loss", float(loss), "pm range", float(out["pm"][batch["moment_mask"]].min()),
          float(out["pm"][batch["moment_mask"]].max()))


def wc_case():
    blob = {}
    for (T, L, C) in [(16, 8, 4), (128, 32, 4), (128, 64, 4), (32, 8, 3), (64, 16, 4)]:
        Wc = ref_models.compute_content_matrix(T, L, C)
        nz = Wc != 0
        size = nz.sum(-1)
        start = torch.where(size > 0, nz.float().argmax(-1), torch.zeros(1, dtype=torch.long))
        val = Wc.max(-1).values
        # contiguity check: non-zeros of every row are one run of equal values
        t = torch.arange(T)
        run = (t >= start.unsqueeze(-1)) & (t < (start + size).unsqueeze(-1))
        assert torch.equal(run, nz)
        assert torch.equal(torch.where(nz, val.unsqueeze(-1).expand_as(Wc), torch.zeros(1)), Wc)
        key = f"{T}_{L}_{C}"
        blob["start/" + key] = start.numpy().astype(np.int32)
        blob["size/" + key] = size.numpy().astype(np.int32)
        blob["val/" + key] = val.numpy()
    np.savez_compressed(os.path.join(HERE, "g4_wc.npz"), **blob)
    print("g4_wc ok")


FULL = {
    # name: (T, L, C, D, dl, layers, Din, Nq, H)     BASELINE.json configs
    "tacos_yml": (128, 32, 4, 512, 128, 3, 4096, 14, 256),
    "tacos_d500": (128, 32, 4, 512, 128, 3, 500, 14, 256),
    "charades": (64, 16, 4, 512, 128, 3, 1024, 13, 256),
    "anet_yml": (128, 64, 4, 512, 128, 3, 500, 20, 256),
    "anet_t256": (256, 64, 4, 512, 128, 3, 500, 20, 256),
}


def full_case(name, cfg, B=2, seed=7):
    T, L, C, D, dl, layers, Din, Nq, H = cfg
    model = ref_models.SMIN(T, L, C, D, dl, layers, Din, Nq, H)
    shapes = {k: tuple(v.shape) for k, v in model.state_dict().items()}
    sd = O.formula_state_dict(shapes, gain=1.3)
    model.load_state_dict(sd)
    batch = O.synthetic_batch(B, T, L, Nq, Din, seed=seed)
    out = run_reference(model, batch)
    loss = loss_of(out, batch)
    loss.backward()
    blob = {"cfg": np.array(list(cfg) + [B, seed], dtype=np.int64)}
    blob.update({"out/" + k: v.detach().numpy() for k, v in out.items()})
    blob["loss"] = loss.detach().numpy()
    blob["in_checksum"] = np.array([float(batch["video_features"].double().sum()), float(batch["query_features"].double().sum()),
                                    float(batch["moment_mask"].sum()), float(batch["sm"].double().sum())])
    blob["sd_checksum"] = np.array([float(sum(v.double().abs().sum() for v in sd.values()))])
    names, norms = [], []
    for k, p in model.named_parameters():
        names.append(k)
        norms.append(float(p.grad.double().norm()))
    blob["grad_names"] = np.array(names)
    blob["grad_norms"] = np.array(norms)
    # a few complete small gradients for element-wise comparison
    for k in ["localization.conv_layer_pm.weight", "smis.0.content_unit.attn_layer.W_q.bias",
              "smis.2.moment_unit.conv_layer_fb.bias", "smis.1.content_unit.linear_s_hat.bias",
              "smis.0.boundary_unit.attn_layer.W_k.bias"]:
        blob["grad/" + k] = dict(model.named_parameters())[k].grad.numpy()
    np.savez_compressed(os.path.join(HERE, "g5_" + name + ".npz"), **blob)
    pmv = out["pm"][batch["moment_mask"]]
    print("g5", name, "loss", float(loss), "pm min/max/std", float(pmv.min()), float(pmv.max()), float(pmv.std()))


def ious_case():
    g = torch.Generator().manual_seed(3)
    B, L = 6, 16
    lm = torch.ones(B, L, dtype=torch.bool)
    lm[1, 11:] = False
    lm[4, 5:] = False
    mm = torch.triu(lm.unsqueeze(2) & lm.unsqueeze(1))
    pm = torch.rand(B, L, L, generator=g) * mm
    ps = torch.rand(B, L, generator=g) * lm
    pe = torch.rand(B, L, generator=g) * lm
    sm = torch.rand(B, L, L, generator=g) * mm
    met = ref_utils.compute_ious(pm, ps, pe, mm, sm)
    keys = sorted(met.keys())
    np.savez_compressed(os.path.join(HERE, "g6_ious.npz"), pm=pm.numpy(), ps=ps.numpy(), pe=pe.numpy(), mm=mm.numpy(),
                        sm=sm.numpy(), keys=np.array(keys), vals=np.array([met[k] for k in keys]))
    print("g6 ok", dict(met))


if __name__ == "__main__":
    #            name            T   L  C  D   dl lay Din Nq H   B seed boost
    tiny_case("g1_r4",          32, 8, 4, 32, 16, 2, 24, 5, 16, 3, 11, 2.0)
    tiny_case("g2_r2_ragged",   16, 8, 4, 32, 16, 2, 24, 6, 16, 4, 12, 2.0)
    tiny_case("g3_c3",          32, 8, 3, 32, 16, 3, 20, 7, 16, 5, 13, 1.6)
    wc_case()
    for name, cfg in FULL.items():
        full_case(name, cfg)
    ious_case()
