from speech_decoding_amd.loss import CLIPLoss  # noqa: F401

__all__ = ["CLIPLoss"]
