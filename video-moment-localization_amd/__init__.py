"""MI355X-native SMIN hot path (cross-modal fusion + 2D temporal proposal scoring).

Drop-in for the ``models.py`` module surface of ChanukyaVardhan/Video-Moment-Localization:
same classes, constructor/forward signatures and state_dict keys, with the
per-cell work done by hand-written HIP kernels for gfx950 behind the C ABI in
``include/smin_hip.h`` (``libsmin_hip.so``, bound with ctypes in ``_lib.py``).

The directory name contains '-' so it cannot be imported by name; the repo-root ``models.py``
loads it under the module name ``vml_amd``.
"""
import os as _os

# The train step runs on three HIP streams (main, boundary unit / tail, low-priority weight gradients); under data parallel RCCL
# adds its own.  The HIP runtime deals streams onto GPU_MAX_HW_QUEUES hardware queues (default 4): with RCCL's stream present two
# of the step's streams shared a queue and serialised (measured: 18.7 -> 20.6 ms/step at world size 1; 18.9 with 6 or 8 queues).
# Read when the runtime initialises, so it is set here, before anything touches the device; an explicit setting wins.
_os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

from . import _lib, distributed  # noqa: F401,E402
from ._lib import get_gemm_mode, set_gemm_mode  # noqa: F401
from .cells import CellLayout  # noqa: F401
from .modules import (  # noqa: F401
    SMIN, SMI, Attention, Backbone, BoundaryUnit, ContentAttention, ContentUnit, Localization,
    MomentUnit, ProposalGeneration, QueryEncoder, VideoEncoder, compute_content_matrix,
)
from .training import loss_fn, loss_fn_torch, bce_loss, compute_ious, compute_ious_torch, CapturedStep  # noqa: F401
from .labels import build_targets  # noqa: F401
from .feeder import BatchFeeder, build_targets_hip  # noqa: F401
