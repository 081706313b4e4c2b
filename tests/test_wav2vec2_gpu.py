"""Frozen wav2vec 2.0 embedder on MI355X against the CPU restatement (oracle/wav2vec2_oracle.py, itself
pinned to the installed `transformers` Wav2Vec2Model in tests/test_wav2vec2_cpu.py) on seeded random weights.
Parity status: architecture pinned, pretrained weights unobtainable offline."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

SMALL = dict(conv_dim=(64,) * 7, hidden_size=128, num_attention_heads=2, intermediate_size=256, num_hidden_layers=5,
             num_conv_pos_embeddings=16, num_conv_pos_embedding_groups=4)
MID = dict(conv_dim=(512,) * 7, hidden_size=256, num_attention_heads=4, intermediate_size=512, num_hidden_layers=4,
           num_conv_pos_embeddings=128, num_conv_pos_embedding_groups=16)
# per-tensor relative L2 error of a hidden state against the fp32 oracle on the same weights
REL = {torch.float32: 2e-5, torch.bfloat16: 3e-2, torch.float16: 4e-3}


def rel_l2(a, b):
    return float((a.double() - b.double()).norm() / b.double().norm())


def build(cfg_kwargs, dtype, seed=0):
    from oracle import wav2vec2_oracle as W
    from speech_decoding_amd.wav2vec2 import Wav2Vec2Config, Wav2Vec2Embedder
    ocfg = W.W2V2Config(**cfg_kwargs)
    sd = W.random_state_dict(ocfg, seed)
    emb = Wav2Vec2Embedder(sd, Wav2Vec2Config(**cfg_kwargs), dtype=dtype, device="cuda:0")
    return W, ocfg, sd, emb


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16, torch.float16])
@pytest.mark.parametrize("name,cfg,n", [("small", SMALL, 6000), ("small-ragged", SMALL, 3217), ("mid", MID, 30000),
                                        ("small-long", SMALL, 500123)])       # 1562 frames: 25 key blocks per query block
def test_hidden_states_match_oracle(dtype, name, cfg, n):
    W, ocfg, sd, emb = build(cfg, dtype)
    wave = torch.randn(n, generator=torch.Generator().manual_seed(1))
    with torch.no_grad():
        ref = W.hidden_states(sd, ocfg, wave[None])
    got = emb.hidden_states(wave)
    assert len(got) == len(ref) == ocfg.num_hidden_layers + 1
    assert got[0].shape == ref[0][0].shape == (W.n_frames(n, ocfg), ocfg.hidden_size)
    for i, (g, r) in enumerate(zip(got, ref)):
        assert torch.isfinite(g).all(), f"hidden state {i}"
        assert rel_l2(g.cpu(), r[0]) < REL[dtype], f"hidden state {i}: {rel_l2(g.cpu(), r[0]):.3e}"


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_chunked_last_four_mean_matches_reference_call(dtype):
    """`getW2VLastFourLayersAvg` (wav2vec_util.py:14-32): 10 chunks, mean of the last four hidden states, (H, frames)."""
    W, ocfg, sd, emb = build(SMALL, dtype)
    waveform = torch.randn(1, 20003, generator=torch.Generator().manual_seed(2))
    ref = W.w2v_last_four_layers_avg(sd, ocfg, waveform)
    got = emb.embed(waveform)
    assert got.shape == ref.shape
    assert rel_l2(got.cpu(), ref) < REL[dtype]
    # a second call reuses the workspace and gives the same bits
    assert torch.equal(emb(waveform), got)


def test_resample_matches_restatement():
    from oracle import wav2vec2_oracle as W
    from speech_decoding_amd.wav2vec2 import resample_fft
    x = np.random.RandomState(0).randn(5, 777)
    up = 120.0 / 49.9737
    ref = W.resample_fft(x, up)
    got = resample_fft(torch.from_numpy(x).cuda(), up).cpu().numpy()
    assert got.shape == ref.shape
    np.testing.assert_allclose(got, ref, rtol=0, atol=1e-10)
    short = np.random.RandomState(1).randn(2, 40)            # shorter than the padding
    np.testing.assert_allclose(resample_fft(torch.from_numpy(short).cuda(), 2.4).cpu().numpy(), W.resample_fft(short, 2.4), atol=1e-10)


def test_unsupported_variants_are_refused():
    from speech_decoding_amd.wav2vec2 import Wav2Vec2Config, Wav2Vec2Embedder
    from oracle import wav2vec2_oracle as W
    bad = dict(SMALL, hidden_size=96, num_attention_heads=2)      # head dimension 48
    with pytest.raises(ValueError):
        Wav2Vec2Embedder(W.random_state_dict(W.W2V2Config(**bad)), Wav2Vec2Config(**bad), dtype=torch.float32)


def _fixture():
    import os
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "w2v2_small.npz"))
    kw = dict(conv_dim=tuple(int(v) for v in z["cfg_conv_dim"]), conv_kernel=tuple(int(v) for v in z["cfg_conv_kernel"]),
              conv_stride=tuple(int(v) for v in z["cfg_conv_stride"]), hidden_size=int(z["cfg_hidden_size"]),
              num_attention_heads=int(z["cfg_num_attention_heads"]), intermediate_size=int(z["cfg_intermediate_size"]),
              num_hidden_layers=int(z["cfg_num_hidden_layers"]), num_conv_pos_embeddings=int(z["cfg_num_conv_pos_embeddings"]),
              num_conv_pos_embedding_groups=int(z["cfg_num_conv_pos_embedding_groups"]))
    sd = {k[3:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("sd/")}
    return kw, sd, z


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16, torch.float16])
def test_reference_function_fixture(dtype):
    """tests/golden/w2v2_small.npz: output of the REFERENCE's getW2VLastFourLayersAvg on a seeded HF model (make_w2v2_golden.py)."""
    from speech_decoding_amd.wav2vec2 import Wav2Vec2Config, Wav2Vec2Embedder
    kw, sd, z = _fixture()
    emb = Wav2Vec2Embedder(sd, Wav2Vec2Config(**kw), dtype=dtype)
    got = emb.embed(torch.from_numpy(z["waveform"]))
    assert got.shape == z["expected"].shape and got.dtype == torch.float32
    assert rel_l2(got.cpu(), torch.from_numpy(z["expected"])) < REL[dtype]
    hs = emb.hidden_states(torch.from_numpy(z["waveform"])[0, :4000])
    for i, h in enumerate(hs):
        assert rel_l2(h.cpu(), torch.from_numpy(z[f"hs/{i}"])) < REL[dtype], i
    if dtype == torch.float32:
        np.testing.assert_allclose(got.cpu().numpy(), z["expected"], rtol=1e-3, atol=1e-4)


def test_shim_accepts_a_transformers_model():
    """The drop-in module: same names and call as the reference (wav2vec_util.py:8-32), HF model object in, CPU fp32 out."""
    tr = pytest.importorskip("transformers")
    from speech_decoding.utils import wav2vec_util as shim
    from oracle import wav2vec2_oracle as W
    kw, sd, z = _fixture()
    model = tr.Wav2Vec2Model(tr.Wav2Vec2Config(**W.W2V2Config(**kw).hf_kwargs())).eval()
    missing = model.load_state_dict(sd, strict=False)
    assert set(missing.missing_keys) <= {"masked_spec_embed"}
    shim.COMPUTE_DTYPE = torch.float32
    out = shim.getW2VLastFourLayersAvg(model, torch.from_numpy(z["waveform"]))
    assert out.device.type == "cpu" and out.dtype == torch.float32
    np.testing.assert_allclose(out.numpy(), z["expected"], rtol=1e-3, atol=1e-4)


_XLSR = {}


def _xlsr(dtype):
    """facebook/wav2vec2-large-xlsr-53's dimensions (24 layers, 1024 hidden, 16 heads, 4096 FFN) with seeded random weights."""
    from oracle import wav2vec2_oracle as W
    from speech_decoding_amd.wav2vec2 import Wav2Vec2Config, Wav2Vec2Embedder
    if "sd" not in _XLSR:
        _XLSR["cfg"] = W.W2V2Config()
        _XLSR["sd"] = W.random_state_dict(_XLSR["cfg"], seed=3, scale=0.7)
        wave = torch.randn(40000, generator=torch.Generator().manual_seed(4))
        with torch.no_grad():
            _XLSR["wave"], _XLSR["ref"] = wave, W.hidden_states(_XLSR["sd"], _XLSR["cfg"], wave[None])
    return _XLSR, Wav2Vec2Embedder(_XLSR["sd"], Wav2Vec2Config(), dtype=dtype)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_xlsr53_dimensions_against_oracle(dtype):
    c, emb = _xlsr(dtype)
    got = emb.hidden_states(c["wave"])
    assert len(got) == 25 and got[0].shape == (124, 1024)
    worst = max(rel_l2(g.cpu(), r[0]) for g, r in zip(got, c["ref"]))
    assert worst < (1e-4 if dtype == torch.float32 else 6e-2), worst
    lf = emb.last_four_mean(c["wave"])
    ref = torch.stack([r[0] for r in c["ref"][-4:]]).mean(0)
    assert rel_l2(lf.cpu(), ref) < (1e-4 if dtype == torch.float32 else 6e-2)


def test_graph_replay_equals_eager_launches():
    """HIP-graph replay of a chunk (the default) gives the same bits as launching its kernels one by one."""
    W, ocfg, sd, emb = build(SMALL, torch.bfloat16)
    waveform = torch.randn(1, 30011, generator=torch.Generator().manual_seed(5))
    a = emb.embed(waveform)
    b = emb.embed(waveform * 0.5)           # replays with new input
    emb.use_graphs = False
    assert torch.equal(emb.embed(waveform), a)
    assert torch.equal(emb.embed(waveform * 0.5), b)
    assert not torch.equal(a, b)
