"""ctypes binding of libsmin_hip.so (C ABI: include/smin_hip.h).  No CPU fallback: every entry point
raises if the library is missing or a tensor is not on a HIP device."""
import ctypes
import os
import subprocess

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libsmin_hip.so")
TORCH_LIB_PATH = os.path.join(_HERE, "libsmin_torch.so")        # TORCH_LIBRARY(smin_hip, ...): csrc/torch_binding.cpp
CSRC = os.path.join(_HERE, "csrc")

_vp, _i, _sz = ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t
ABI_VERSION = 2                                                 # include/smin_hip.h SMIN_HIP_ABI_VERSION

# name -> argtypes (restype is int unless listed in _RESTYPE); mirrors include/smin_hip.h one to one
SIGNATURES = {
    "smin_abi_version": [],
    "smin_target_arch": [],
    "smin_set_gemm_mode": [_i],
    "smin_get_gemm_mode": [],
    "smin_prof_enable": [_i],
    "smin_prof_read": [_vp, _vp, _i],
    "smin_workspace_bytes": [_i] * 6,
    "smin_transpose_batch": [_vp] * 5 + [_i],
    "smin_param_prep_fwd": [_vp, _vp, _i, _i, _i] + [_vp] * 5,
    "smin_param_prep_bwd": [_vp, _vp, _i, _i, _i] + [_vp] * 9,
    "smin_sum_lists": [_vp, _vp, _i, _sz, _vp],
    "smin_col_sum_workspace_bytes": [_i, _i],
    "smin_col_sum": [_vp, _vp, _i, _i, _vp, _vp, _sz],
    "smin_proposal_map_fwd": [_vp, _vp, _vp] + [_i] * 6 + [_vp] * 3 + [_vp, _sz],
    "smin_proposal_map_bwd": [_vp] * 7 + [_i] * 6 + [_vp, _vp, _sz, _vp, _vp],
    "smin_clip_event_table": [_vp] + [_i] * 3 + [_vp] * 3,
    "smin_gate_fwd": [_vp] * 4 + [_i] * 2 + [_vp],
    "smin_gate_fwd_sum": [_vp] * 4 + [_i] * 2 + [_vp] * 3,
    "smin_gate_bwd": [_vp, _vp, _i, _vp, _i, _vp, _vp, _vp] + [_i] * 4 + [_vp] * 2 + [_vp, _sz] + [_vp] * 3,
    "smin_content_unit_fwd": [_vp] * 5 + [_i] * 7 + [_vp] * 9 + [_vp, _i] + [_vp] * 4,
    "smin_content_unit_bwd": [_vp] * 6 + [_i] * 7 + [_vp] * 9 + [_vp] * 10 + [_vp, _sz, _i],
    "smin_boundary_reduce_fwd": [_vp] * 5 + [_i] * 4 + [_vp],
    "smin_boundary_reduce_bwd": [_vp] * 6 + [_i] * 4 + [_vp] * 2,
    "smin_boundary_unit_fwd": [_vp] * 7 + [_i] * 5 + [_vp] * 6 + [_vp] * 7,
    "smin_boundary_unit_bwd": [_vp] * 8 + [_i] * 5 + [_vp] * 4 + [_vp] * 6 + [_vp] * 8 + [_vp, _sz],
    "smin_moment_unit_fwd": [_vp] * 5 + [_i] * 4 + [_vp] * 3 + [_vp],
    "smin_pair_product": [_vp] * 3 + [_i] * 3 + [_vp],
    "smin_moment_unit_bwd": [_vp] * 7 + [_i] * 4 + [_vp] * 5 + [_vp, _sz, _i, _vp, _vp, _vp],
    "smin_pair_product_bf16": [_vp] * 3 + [_i] * 3 + [_vp],
    "smin_moment_unit_fwd_x1h": [_vp] * 5 + [_i] * 4 + [_vp] * 3 + [_vp],
    "smin_moment_unit_bwd_x1h": [_vp] * 7 + [_i] * 4 + [_vp] * 5 + [_vp, _sz, _i, _vp, _vp, _vp],
    "smin_score_map_fwd": [_vp] * 4 + [_i] * 4 + [_vp] * 7,
    "smin_score_map_bwd": [_vp] * 8 + [_i] * 4 + [_vp] * 3 + [_vp] * 6 + [_vp, _sz],
    "smin_loss_fwd": [_vp] * 14 + [_i] * 2 + [_vp] * 2,
    "smin_loss_bwd": [_vp] * 16 + [_i] * 2 + [_vp] * 4,
    "smin_compute_ious": [_vp] * 6 + [_i] * 2 + [_vp] * 2,
    "smin_build_targets": [_vp] * 5 + [_i] * 4 + [_vp] * 12,
    "smin_word_prep_fwd": [_vp] * 5 + [_i] * 5 + [_vp] * 5,
    "smin_word_prep_bwd_workspace_bytes": [_i] * 5,
    "smin_word_prep_bwd": [_vp] * 11 + [_i] * 5 + [_vp] * 3 + [_vp, _sz],
    "smin_build_cells": [_vp, _vp, _i, _i, _i, _vp, _vp, _vp],
    "smin_build_cells_n": [_vp, _vp, _i, _i, _i, _i, _vp, _vp, _vp, _vp],
    "smin_pack_cells": [_vp, _vp, _vp, _i, _i, _i, _vp],
    "smin_unpack_cells": [_vp, _vp, _vp, _i, _i, _i, _vp],
    "smin_gemm_nt": [_vp] * 4 + [_i] * 3,
    "smin_gemm_nt_acc": [_vp] * 4 + [_i] * 3,
    "smin_clip_window_means_fwd": [_vp] * 3 + [_i, _vp] + [_i] * 7 + [_vp, _vp, _sz],
    "smin_clip_window_means_bwd": [_vp] * 5 + [_i] * 7 + [_vp, _vp, _sz, _vp, _vp],
    "smin_content_attn_fwd": [_vp] * 4 + [_i] * 6 + [_vp] * 7,
    "smin_content_attn_fwd_cch": [_vp] * 4 + [_i] * 6 + [_vp] * 7,
    "smin_content_attn_bwd_workspace_bytes": [_i] * 4,
    "smin_content_attn_bwd": [_vp] * 6 + [_i] * 6 + [_vp] * 10 + [_vp, _sz],
    "smin_linear_rows_fwd": [_vp, _vp, _i] + [_vp] * 4 + [_i] * 4 + [_vp],
    "smin_linear_rows_fwd_xh": [_vp, _vp, _i] + [_vp] * 4 + [_i] * 4 + [_vp],
    "smin_linear_rows_bwd_xh": [_vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp, _vp, _sz],
    "smin_linear_rows_bwd_workspace_bytes": [_i] * 3,
    "smin_linear_rows_bwd": [_vp, _vp, _vp, _i, _vp] + [_i] * 3 + [_vp] * 3 + [_vp, _sz],
    "smin_linear_rows_dx_acc": [_vp, _vp, _i, _vp] + [_i] * 3 + [_vp],
    "smin_group_sum": [_vp, _vp, _i, _i, _i, _vp],
    "smin_video_encoder_fwd": [_vp] * 7 + [_i] * 4 + [_vp] * 2,
    "smin_video_encoder_gate": [_vp] * 3 + [_i] * 3 + [_vp],
    "smin_video_encoder_bwd_workspace_bytes": [_i] * 4,
    "smin_video_encoder_bwd": [_vp] * 6 + [_i] * 4 + [_vp] * 4 + [_vp, _sz],
    "smin_bilstm_layer_fwd": [_vp] * 6 + [_i] * 4 + [_vp] * 3,
    "smin_lstm_pack": [_vp, _vp, _i, _i] + [_vp] * 4,
    "smin_lstm_pack_layers": [_vp, _i, _vp, _vp, _i] + [_vp] * 4,
    "smin_lstm_cluster_error": [],
    "smin_sentence_feature_fwd": [_vp, _vp, _vp, _i, _i, _i, _vp],
    "smin_sentence_feature_bwd": [_vp, _vp, _vp, _i, _i, _i, _vp],
    "smin_bilstm_layer_bwd_workspace_bytes": [_i] * 4,
    "smin_bilstm_layer_bwd": [_vp] * 9 + [_i] * 4 + [_vp] * 4 + [_vp, _sz],
    "smin_step_prologue": [_vp] * 7 + [_i] * 5 + [_vp] * 8,
    "smin_bilstm_layer_bwd_weights": [_vp, _i, _vp, _vp] + [_i] * 4 + [_vp] * 4 + [_vp, _sz],
}
_RESTYPE = {"smin_target_arch": ctypes.c_char_p, "smin_workspace_bytes": _sz,
            "smin_content_attn_bwd_workspace_bytes": _sz, "smin_linear_rows_bwd_workspace_bytes": _sz,
            "smin_bilstm_layer_bwd_workspace_bytes": _sz, "smin_video_encoder_bwd_workspace_bytes": _sz, "smin_word_prep_bwd_workspace_bytes": _sz,
            "smin_col_sum_workspace_bytes": _sz}

_lib = None
_ws = {}


class SminHipError(RuntimeError):
    pass


def build(verbose=False):
    """Compile every HIP source for gfx950 into libsmin_hip.so (hipcc cross-compiles without a GPU)."""
    res = subprocess.run(["make", "-C", CSRC, "-j8"], capture_output=True, text=True)
    if verbose or res.returncode != 0:
        print(res.stdout[-4000:])
        print(res.stderr[-4000:])
    if res.returncode != 0:
        raise SminHipError("building libsmin_hip.so failed (see output above)")
    return LIB_PATH


def load():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise SminHipError(
            f"{LIB_PATH} is missing: the SMIN hot path has no CPU fallback. Build it with "
            "`python -c 'import __graft_entry__ as g; g.build()'` (needs hipcc, --offload-arch=gfx950).")
    lib = ctypes.CDLL(LIB_PATH)
    for name, args in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError here == header/library mismatch
        fn.argtypes = args
        fn.restype = _RESTYPE.get(name, _i)
    if lib.smin_abi_version() != ABI_VERSION:
        raise SminHipError("libsmin_hip.so ABI version mismatch")
    _lib = lib
    if DEFAULT_GEMM_MODE != "f32":                        # deployment switch: SMIN_GEMM_MODE=f32e|bf16x3|bf16 (see set_gemm_mode)
        check(lib.smin_set_gemm_mode(GEMM_MODES[DEFAULT_GEMM_MODE]), "smin_set_gemm_mode")
    return lib


_torch_ops = None


def load_torch():
    """The torch-extension binding (csrc/torch_binding.cpp): registers torch.ops.smin_hip.{smin_forward, smin_loss} -- the
    whole forward as one library call with its autograd graph built in C++.  Raises if the library is missing."""
    global _torch_ops
    if _torch_ops is not None:
        return _torch_ops
    load()                                            # the C ABI library it links against
    if not os.path.exists(TORCH_LIB_PATH):
        raise SminHipError(
            f"{TORCH_LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(or set SMIN.native_host = False to drive the same kernels from the Python host)")
    torch.ops.load_library(TORCH_LIB_PATH)
    ops = torch.ops.smin_hip
    if ops.abi_version() != ABI_VERSION:
        raise SminHipError("libsmin_torch.so / libsmin_hip.so ABI version mismatch")
    _torch_ops = ops
    return ops


def ptr(t):
    """Device pointer of a contiguous HIP tensor of one of the element types the C ABI takes (None -> NULL)."""
    if t is None:
        return None
    if not t.is_cuda:
        raise SminHipError("the SMIN hot path runs on a HIP device only (got a CPU tensor); there is no CPU fallback")
    if not t.is_contiguous():
        raise SminHipError("internal error: non-contiguous tensor handed to the C ABI")
    if t.dtype not in (torch.float32, torch.int32, torch.uint8, torch.float64, torch.bool, torch.int64):
        raise SminHipError(f"unsupported dtype {t.dtype}")
    return ctypes.c_void_p(t.data_ptr())


def stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def workspace(nbytes, device):
    """Persistent scratch buffer per (device, stream), grown on demand: calls on one stream are stream-ordered, and
    units running side by side on two streams never share scratch."""
    key = (device.type, device.index, torch.cuda.current_stream(device).cuda_stream if device.type == "cuda" else 0)
    buf = _ws.get(key)
    if buf is None or buf.numel() < nbytes:
        buf = torch.empty(int(nbytes * 1.25) + 4096, dtype=torch.uint8, device=device)
        _ws[key] = buf
    return buf


def check(rc, name):
    if rc != 0:
        raise SminHipError(f"{name} failed with code {rc}" + (" (argument rejected at csrc line %d)" % (-rc - 1000) if rc < -1000 else ""))


def call(name, *args):
    check(getattr(load(), name)(*args), name)


PROF_TAGS = {1: "moment_fwd", 2: "moment_dx", 3: "moment_dw", 4: "attn_fwd", 5: "attn_bwd"}


def prof_enable(on=True):
    """Bracket the tagged launches (include/smin_hip.h SMIN_PROF_*) with HIP events on their launch stream."""
    check(load().smin_prof_enable(int(on)), "smin_prof_enable")


def prof_read(cap=1 << 16):
    """{tag name: [milliseconds per launch, in launch order]} of everything recorded since prof_enable(True)."""
    tags, ms = (ctypes.c_int32 * cap)(), (ctypes.c_float * cap)()
    n = load().smin_prof_read(tags, ms, cap)
    if n < 0:
        raise SminHipError(f"smin_prof_read failed with code {n}")
    out = {}
    for k in range(n):
        out.setdefault(PROF_TAGS.get(tags[k], str(tags[k])), []).append(ms[k])
    return out


GEMM_MODES = {"f32": 0, "bf16x3": 1, "bf16": 2, "f32e": 3}
DEFAULT_GEMM_MODE = os.environ.get("SMIN_GEMM_MODE", "f32")      # the mode the library starts in (and tests restore)
if DEFAULT_GEMM_MODE not in GEMM_MODES:
    raise ValueError(f"SMIN_GEMM_MODE={DEFAULT_GEMM_MODE!r}: expected one of {sorted(GEMM_MODES)}")


def set_gemm_mode(mode):
    """Arithmetic of the dense contractions (forward, input gradients, weight gradients): "f32" (exact fp32 MFMA, default),
    "f32e" (fp32 emulated on the bf16 matrix cores: exact three-way bf16 split of each operand, six products, fp32 accumulation --
    agrees with "f32" to fp32 rounding), "bf16x3" (two-way split, three products; ~1e-5 relative) or "bf16" (operands rounded to
    bf16 once; ~4e-3 relative per product -- BASELINE.json configs[1])."""
    check(load().smin_set_gemm_mode(GEMM_MODES[mode]), "smin_set_gemm_mode")


def get_gemm_mode():
    m = load().smin_get_gemm_mode()
    return [k for k, v in GEMM_MODES.items() if v == m][0]
