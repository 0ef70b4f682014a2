"""Drop-in replacement for the reference's ``models.py`` (``from models import SMIN`` in main.py:3,
``from models import *`` in simpletest.py:9).  Everything lives in the package
``video-moment-localization_amd/`` (not importable by name because of the '-'), loaded here as ``vml_amd``."""
import importlib.util
import os
import sys

_PKG_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "video-moment-localization_amd")


def _load_package():
    if "vml_amd" in sys.modules:
        return sys.modules["vml_amd"]
    spec = importlib.util.spec_from_file_location("vml_amd", os.path.join(_PKG_DIR, "__init__.py"),
                                                  submodule_search_locations=[_PKG_DIR])
    mod = importlib.util.module_from_spec(spec)
    sys.modules["vml_amd"] = mod
    spec.loader.exec_module(mod)
    return mod


vml_amd = _load_package()

from vml_amd.modules import (  # noqa: E402,F401
    SMIN, SMI, Attention, Backbone, BoundaryUnit, ContentAttention, ContentUnit, Localization,
    MomentUnit, ProposalGeneration, QueryEncoder, VideoEncoder, compute_content_matrix,
)

__all__ = ["SMIN", "SMI", "Attention", "Backbone", "BoundaryUnit", "ContentAttention", "ContentUnit", "Localization",
           "MomentUnit", "ProposalGeneration", "QueryEncoder", "VideoEncoder", "compute_content_matrix", "vml_amd"]
