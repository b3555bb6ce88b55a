from .face_nerf import FaceNeRF  # noqa: F401
from .audio_net import AudioNet, AudioAttNet, DeepSpeechAudNet  # noqa: F401
