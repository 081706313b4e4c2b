from speech_decoding_amd.models import (BrainEncoder, Classifier, ConvBlock, SpatialAttention,  # noqa: F401
                                        SubjectBlock)
