"""CPU checks for the wav2vec 2.0 embedder: the oracle restatement against (a) the fixture the REFERENCE's
own `getW2VLastFourLayersAvg` produced (tests/golden/make_w2v2_golden.py) and (b) the installed `transformers` model on
fresh seeded weights; the FFT resampling against scipy; the host-side chunking / frame arithmetic of the product."""
import os

import numpy as np
import pytest
import torch

from oracle import wav2vec2_oracle as W

GOLD = os.path.join(os.path.dirname(__file__), "golden", "w2v2_small.npz")


def load_fixture():
    z = np.load(GOLD)
    cfg = W.W2V2Config(conv_dim=tuple(int(v) for v in z["cfg_conv_dim"]), conv_kernel=tuple(int(v) for v in z["cfg_conv_kernel"]),
                       conv_stride=tuple(int(v) for v in z["cfg_conv_stride"]), hidden_size=int(z["cfg_hidden_size"]),
                       num_attention_heads=int(z["cfg_num_attention_heads"]), intermediate_size=int(z["cfg_intermediate_size"]),
                       num_hidden_layers=int(z["cfg_num_hidden_layers"]), num_conv_pos_embeddings=int(z["cfg_num_conv_pos_embeddings"]),
                       num_conv_pos_embedding_groups=int(z["cfg_num_conv_pos_embedding_groups"]))
    sd = {k[3:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("sd/")}
    hs = [torch.from_numpy(z[f"hs/{i}"]) for i in range(cfg.num_hidden_layers + 1)]
    return cfg, sd, torch.from_numpy(z["waveform"]), torch.from_numpy(z["expected"]), hs


def test_oracle_reproduces_the_reference_function_output():
    cfg, sd, waveform, expected, hs = load_fixture()
    got = W.w2v_last_four_layers_avg(sd, cfg, waveform)
    assert got.shape == expected.shape == (cfg.hidden_size, sum(W.n_frames(b - a, cfg) for a, b in W.chunk_bounds(waveform.shape[-1])))
    np.testing.assert_allclose(got.numpy(), expected.numpy(), rtol=1e-4, atol=2e-5)
    with torch.no_grad():
        mine = W.hidden_states(sd, cfg, waveform[:, :4000])
    assert len(mine) == len(hs)
    for a, b in zip(mine, hs):
        np.testing.assert_allclose(a[0].numpy(), b.numpy(), rtol=1e-4, atol=2e-5)


def test_oracle_matches_installed_transformers_on_fresh_weights():
    tr = pytest.importorskip("transformers")
    cfg = W.W2V2Config(conv_dim=(48,) * 7, hidden_size=96, num_attention_heads=3, intermediate_size=160, num_hidden_layers=4,
                       num_conv_pos_embeddings=32, num_conv_pos_embedding_groups=3)
    torch.manual_seed(7)
    model = tr.Wav2Vec2Model(tr.Wav2Vec2Config(**cfg.hf_kwargs())).eval()
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    for k in sd:
        if k.endswith("bias"):
            sd[k] = 0.1 * torch.randn_like(sd[k])
        elif "layer_norm.weight" in k:
            sd[k] = 1 + 0.1 * torch.randn_like(sd[k])
    model.load_state_dict(sd)
    wave = torch.randn(2, 5000)
    with torch.no_grad():
        ref = model(input_values=wave, output_hidden_states=True).hidden_states
        got = W.hidden_states(sd, cfg, wave)
    assert len(ref) == len(got) == cfg.num_hidden_layers + 1
    for a, b in zip(got, ref):
        np.testing.assert_allclose(a.numpy(), b.numpy(), rtol=1e-4, atol=2e-5)
    # both key spellings of the weight-normed positional conv give the same effective weight
    old = {k.replace("parametrizations.weight.original0", "weight_g").replace("parametrizations.weight.original1", "weight_v"): v
           for k, v in sd.items()}
    assert torch.equal(W.pos_conv_weight(old), W.pos_conv_weight(sd))


@pytest.mark.parametrize("n", [16003, 160000, 99, 10, 1234567])
def test_chunking_matches_numpy_array_split(n):
    from speech_decoding_amd.wav2vec2 import chunk_bounds
    splits = np.array_split(np.arange(n), 10)
    want = [(int(s[0]), int(s[-1]) + 1) if len(s) else None for s in splits]
    for got in (W.chunk_bounds(n), chunk_bounds(n)):
        assert [g for g, w in zip(got, want) if w is not None] == [w for w in want if w is not None]
        assert all(a == b for (a, b), w in zip(got, want) if w is None)


def test_frame_arithmetic_and_config_mapping():
    from speech_decoding_amd.wav2vec2 import Wav2Vec2Config
    cfg = Wav2Vec2Config()
    assert cfg.n_frames(16000) == 49 and cfg.n_frames(400) == 1 and cfg.n_frames(399) == 0      # 20 ms hop, 25 ms window
    assert cfg.n_frames(48000) == W.n_frames(48000, W.W2V2Config())
    tr = pytest.importorskip("transformers")
    hf = tr.Wav2Vec2Config(**W.W2V2Config().hf_kwargs())
    assert Wav2Vec2Config.from_hf(hf) == cfg
    with pytest.raises(ValueError):
        Wav2Vec2Config.from_hf(tr.Wav2Vec2Config())          # base model: group-norm encoder, post-layer-norm transformer


def test_fft_resampling_against_scipy():
    from scipy.signal import resample
    rng = np.random.RandomState(0)
    for n, up in ((500, 120 / 49.9737), (3617, 2.4012), (64, 0.5), (257, 1.0)):
        x = rng.randn(3, n)
        y = W.resample_fft(x, up)
        assert y.shape == (3, int(round(n * up)))
        xp = W.smart_pad(x, 100)
        m_new = int(round(xp.shape[-1] * up))
        off = int(round(100 * up))
        np.testing.assert_allclose(y, resample(xp, m_new, axis=-1)[..., off:off + y.shape[-1]], atol=1e-10)
    # a band-limited signal is resampled exactly (away from the ends)
    t = np.arange(2000) / 50.0
    up = 2.4
    y = W.resample_fft(np.sin(2 * np.pi * 3.0 * t)[None], up)[0]
    t2 = np.arange(y.shape[0]) / (50.0 * up)
    assert np.abs(y - np.sin(2 * np.pi * 3.0 * t2))[200:-200].max() < 1e-3


def test_shim_exposes_the_reference_names():
    import importlib
    pytest.importorskip("torch")
    try:
        mod = importlib.import_module("speech_decoding.utils.wav2vec_util")
    except Exception as e:              # the HIP library is required at import time on purpose (no CPU fallback)
        pytest.skip(f"shim import needs the built extension: {e}")
    assert callable(mod.load_wav2vec_model) and callable(mod.getW2VLastFourLayersAvg)
