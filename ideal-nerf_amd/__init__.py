"""idealnerf_amd: MI355X-native per-ray hot path of IDEAL-NeRF.

Import as ``idealnerf_amd`` (the directory is named ``ideal-nerf_amd``; the alias module
at the repository root makes it importable).  Layout mirrors the reference:

    idealnerf_amd.models.face_nerf.FaceNeRF        <- models/face_nerf.py
    idealnerf_amd.models.audio_net.*               <- models/audio_net.py
    idealnerf_amd.helper                           <- NeRFs/HeadNeRF/helper.py (render math)
    idealnerf_amd.audio_exp_nerf.Network           <- NeRFs/HeadNeRF/train/audio_exp_nerf.py
    idealnerf_amd.train_torso.Network              <- NeRFs/TorsoNeRF/train_torso.py (composite)
    idealnerf_amd.parallel                         <- row-band tiling + RCCL all-gather
"""
__version__ = "0.1.0"

from . import _lib, ops  # noqa: F401
from .models.face_nerf import FaceNeRF, invalidate_packed, set_default_precision, set_render_precision  # noqa: F401
from .models.face_nerf_agg import FaceNeRFAgg  # noqa: F401
