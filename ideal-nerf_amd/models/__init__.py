from .face_nerf import FaceNeRF  # noqa: F401
from .face_nerf_agg import FaceNeRFAgg  # noqa: F401
from .audio_net import AudioNet, AudioAttNet, DeepSpeechAudNet, clip_audio_features  # noqa: F401
