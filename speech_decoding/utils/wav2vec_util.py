"""Drop-in for the reference's `speech_decoding/utils/wav2vec_util.py` (same two names, same arguments, same return
shape/dtype/device): the frozen wav2vec 2.0 embedder runs on MI355X (speech_decoding_amd/wav2vec2.py)."""
import torch

from speech_decoding_amd.wav2vec2 import Wav2Vec2Embedder

COMPUTE_DTYPE = torch.bfloat16          # fp32 for the exact path


def load_wav2vec_model(wav2vec_model, dtype=None, device="cuda:0"):
    """Reference wav2vec_util.py:8-11: `Wav2Vec2Model.from_pretrained(wav2vec_model)`.  The checkpoint is read by
    `transformers` exactly as there (hub name or local directory); its weights are packed for the HIP path."""
    from transformers import Wav2Vec2Model
    model = Wav2Vec2Model.from_pretrained(wav2vec_model)
    return Wav2Vec2Embedder.from_hf(model, dtype or COMPUTE_DTYPE, device)


def getW2VLastFourLayersAvg(wav2vec, waveform):
    """Reference wav2vec_util.py:14-32: (1, L) waveform -> (hidden_size, frames) CPU fp32: ten chunks, mean of the last
    four hidden states per chunk, stacked along time.  `wav2vec` is what load_wav2vec_model returned, or a
    `transformers.Wav2Vec2Model` (packed on first use and cached on the object)."""
    if not isinstance(wav2vec, Wav2Vec2Embedder):
        emb = getattr(wav2vec, "_sda_embedder", None)
        if emb is None:
            emb = Wav2Vec2Embedder.from_hf(wav2vec, COMPUTE_DTYPE)
            wav2vec._sda_embedder = emb
        wav2vec = emb
    return wav2vec.embed(waveform).cpu()


__all__ = ["load_wav2vec_model", "getW2VLastFourLayersAvg"]
