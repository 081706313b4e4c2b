#!/usr/bin/env python3
"""Generate tests/golden/w2v2_small.npz by RUNNING THE REFERENCE's own `getW2VLastFourLayersAvg`
(/root/reference/speech_decoding/utils/wav2vec_util.py:14-32) on a seeded, randomly initialised HuggingFace
`Wav2Vec2Model` of reduced size with xlsr-53's architecture switches.  Build container only (needs /root/reference and
`transformers`); the fixture it writes is committed and travels, this script's dependency on the reference does not.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_w2v2_golden.py

Same import procedure as make_golden.py: `termcolor` (coloured prints, absent here) is registered as an
empty stand-in in THIS process only; every number in the fixture is computed by the reference function and by
`transformers`.  What the fixture pins: the model ARCHITECTURE and the reference's chunking / last-four-mean / stacking.
What it cannot pin: the pretrained facebook/wav2vec2-large-xlsr-53 weights (no network) — "parity unpinned" for those.

Contents: cfg_* (config fields), sd/<key> (state dict, HF key names), waveform (1, L), expected (H, frames) = the
reference function's output, hs/<i> = `output_hidden_states` of the model on waveform[:, :4000].
"""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REF = os.environ.get("SD_REFERENCE", "/root/reference")
sys.dont_write_bytecode = True
sys.path.insert(0, REF)
for name in ("termcolor", "mne", "mne_bids", "omegaconf"):
    if name not in sys.modules:
        m = types.ModuleType(name)
        if name == "termcolor":
            m.cprint = lambda *a, **k: None
        if name == "omegaconf":
            m.DictConfig = dict
            m.open_dict = lambda cfg: cfg
        sys.modules[name] = m

from transformers import Wav2Vec2Config, Wav2Vec2Model                      # noqa: E402  (third party, as the reference uses it)
from speech_decoding.utils.wav2vec_util import getW2VLastFourLayersAvg      # noqa: E402  (the reference)

CFG = dict(conv_dim=[32] * 7, conv_kernel=[10, 3, 3, 3, 3, 2, 2], conv_stride=[5, 2, 2, 2, 2, 2, 2], conv_bias=True,
           hidden_size=64, num_attention_heads=1, intermediate_size=128, num_hidden_layers=5, num_conv_pos_embeddings=16,
           num_conv_pos_embedding_groups=2, layer_norm_eps=1e-5, feat_extract_norm="layer", do_stable_layer_norm=True,
           num_feat_extract_layers=7, hidden_act="gelu", feat_extract_activation="gelu", apply_spec_augment=False)


def main():
    torch.manual_seed(1234)
    model = Wav2Vec2Model(Wav2Vec2Config(**CFG)).eval()
    with torch.no_grad():                       # move biases / LayerNorm parameters off their 0 / 1 initial values
        for k, p in model.named_parameters():
            if k.endswith("bias"):
                p.normal_(0.0, 0.1)
            elif "layer_norm.weight" in k:
                p.normal_(1.0, 0.1)
    waveform = torch.randn(1, 16003)
    expected = getW2VLastFourLayersAvg(model, waveform)
    with torch.no_grad():
        hs = model(input_values=waveform[:, :4000], output_hidden_states=True).hidden_states
    out = {"waveform": waveform.numpy(), "expected": expected.numpy()}
    for k in ("conv_dim", "conv_kernel", "conv_stride", "hidden_size", "num_attention_heads", "intermediate_size",
              "num_hidden_layers", "num_conv_pos_embeddings", "num_conv_pos_embedding_groups"):
        out["cfg_" + k] = np.asarray(CFG[k])
    for k, v in model.state_dict().items():
        if v.is_floating_point() and k != "masked_spec_embed":
            out["sd/" + k] = v.numpy()
    for i, h in enumerate(hs):
        out[f"hs/{i}"] = h[0].numpy()
    path = os.path.join(HERE, "w2v2_small.npz")
    np.savez_compressed(path, **out)
    print(path, os.path.getsize(path), "bytes;", expected.shape, len(hs), "hidden states")


if __name__ == "__main__":
    main()
