"""speech_decoding_amd — MI355X (gfx950) implementation of the contrastive training hot path of
SeanNobel/speech-decoding: BrainEncoder, CLIPLoss, Classifier on hand-written HIP kernels."""
from .lib import SdaError, load as load_library          # noqa: F401
from .models import BrainEncoder, Classifier              # noqa: F401
from .loss import CLIPLoss                                # noqa: F401
from .config import Config, load_config                   # noqa: F401

__all__ = ["BrainEncoder", "Classifier", "CLIPLoss", "Config", "load_config", "load_library", "SdaError"]
