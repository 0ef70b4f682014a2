"""Shared helpers for the test-suite (fixture loading, batch plumbing)."""
import os

import numpy as np
import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

TINY = ["g1_r4", "g2_r2_ragged", "g3_c3"]
FULL = {
    # name: (T, L, C, D, dl, layers, Din, Nq, H)  -- BASELINE.json configs
    "tacos_yml": (128, 32, 4, 512, 128, 3, 4096, 14, 256),
    "tacos_d500": (128, 32, 4, 512, 128, 3, 500, 14, 256),
    "charades": (64, 16, 4, 512, 128, 3, 1024, 13, 256),
    "anet_yml": (128, 64, 4, 512, 128, 3, 500, 20, 256),
    "anet_t256": (256, 64, 4, 512, 128, 3, 500, 20, 256),
}
IN_KEYS = ["video_features", "video_mask", "query_features", "query_mask", "length_mask", "moment_mask"]


def load_npz(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)


def split_tiny(z):
    cfg = dict(zip(["T", "L", "C", "D", "dl", "layers", "Din", "Nq", "H", "B"], [int(v) for v in z["cfg"]]))
    sd = {k[3:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("sd/")}
    batch = {k[3:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("in/")}
    out = {k[4:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("out/")}
    grads = {k[5:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("grad/")}
    return cfg, sd, batch, out, grads, float(z["loss"])


def model_inputs(batch, device=None):
    xs = [batch[k] for k in IN_KEYS]
    return [x.to(device) for x in xs] if device is not None else xs


def smin_shapes(T, L, C, D, dl, layers, Din, Nq, H):
    """state_dict shapes of the reference SMIN, derived without the reference."""
    s = {"backbone.videoencoder.ve.weight": (D, Din), "backbone.videoencoder.ve.bias": (D,),
         "backbone.videoencoder.pe.weight": (T, D)}
    for layer in range(2):
        for sfx in ("", "_reverse"):
            inp = 300 if layer == 0 else 2 * H
            s[f"backbone.queryencoder.lstm.weight_ih_l{layer}{sfx}"] = (4 * H, inp)
            s[f"backbone.queryencoder.lstm.weight_hh_l{layer}{sfx}"] = (4 * H, H)
            s[f"backbone.queryencoder.lstm.bias_ih_l{layer}{sfx}"] = (4 * H,)
            s[f"backbone.queryencoder.lstm.bias_hh_l{layer}{sfx}"] = (4 * H,)
    for k in range(layers):
        p = f"smis.{k}."
        for n, shp in (("linear_c_hat", (dl, D)), ("linear_w_hat", (dl, D)), ("linear_s_hat", (dl, D)), ("linear_c", (D, dl))):
            s[p + f"content_unit.{n}.weight"] = shp
            s[p + f"content_unit.{n}.bias"] = (shp[0],)
        for n in ("W_q", "W_k"):
            s[p + f"content_unit.attn_layer.{n}.weight"] = (dl, dl)
            s[p + f"content_unit.attn_layer.{n}.bias"] = (dl,)
        for n in ("W_q", "W_k"):
            s[p + f"boundary_unit.attn_layer.{n}.weight"] = (D, D)
            s[p + f"boundary_unit.attn_layer.{n}.bias"] = (D,)
        for n in ("conv_layer_fb", "conv_layer_fc"):
            s[p + f"moment_unit.{n}.weight"] = (D, D, 1, 1)
            s[p + f"moment_unit.{n}.bias"] = (D,)
    s["localization.conv_layer_pm.weight"] = (1, D, 1, 1)
    s["localization.conv_layer_pm.bias"] = (1,)
    for n in ("ps", "pe", "pa"):
        s[f"localization.conv_layer_{n}.weight"] = (1, D, 1)
        s[f"localization.conv_layer_{n}.bias"] = (1,)
    return s
