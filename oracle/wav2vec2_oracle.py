"""TEST INFRASTRUCTURE — CPU restatement of the frozen wav2vec 2.0 speech embedder.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.

What the reference does (`/root/reference/speech_decoding/utils/wav2vec_util.py:8-32`, called from
`dataclass/gwilliams2022.py:327-373`): loads HuggingFace `Wav2Vec2Model.from_pretrained("facebook/wav2vec2-large-xlsr-53")`
(`configs/config.yaml:30`), splits the 16 kHz waveform into 10 chunks (`np.array_split`), runs the model on each chunk
with `output_hidden_states=True`, averages the LAST FOUR hidden states, stacks the chunks along time, transposes to
(features, frames), then resamples the frame axis to the brain rate with `mne.filter.resample`.

The arithmetic of the model lives in a THIRD-PARTY dependency that is not vendored in /root/reference:
`transformers==4.24.0` (reference `requirements`/imports; this image carries a newer transformers whose wav2vec2 forward
is the same published architecture).  This file restates that published algorithm (Baevski et al. 2020; the
"layer-norm feature extractor + stable-layer-norm encoder" variant that xlsr-53's config selects:
`feat_extract_norm="layer"`, `do_stable_layer_norm=True`, `conv_bias=True`) in plain PyTorch functional ops:

  feature encoder   7 x [Conv1d(k, stride, no padding) -> LayerNorm(channels) -> GELU]        k = 10,3,3,3,3,2,2  s = 5,2,2,2,2,2,2
  projection        LayerNorm(512) -> Linear(512 -> H)
  encoder           h = h + GELU(SamePad(weight-normed grouped Conv1d(H, H, k=128, pad=64, groups=16)(h)))
                    L x [h = h + Attn(LN(h));  h = h + FFN(LN(h))],  hidden_states = (input of every layer..., LN(h_L))
  attention         softmax((q k^T) / sqrt(d)) v per head, biased q/k/v/out projections

PARITY STATUS: the ARCHITECTURE is pinned — tests/test_wav2vec2_cpu.py checks this restatement against the installed
`transformers` Wav2Vec2Model on seeded random weights (all hidden states), and tests/golden/w2v2_small.npz holds
inputs/outputs generated that way (tests/golden/make_w2v2_golden.py).  The WEIGHTS are not: the pretrained checkpoint
cannot be obtained offline, so "parity unpinned" holds for the reference's actual embeddings.
`mne` is absent from this image as well; `resample_fft` restates the FFT resampling `mne.filter.resample` performs
(documented algorithm: reflect-limited padding, rfft, spectrum zero-padding, irfft, un-padding) and is checked against
`scipy.signal.resample` on the un-padded core — also "parity unpinned" against mne itself.
"""
from dataclasses import dataclass, field
from typing import Dict, List, Tuple

import numpy as np
import torch
import torch.nn.functional as F


@dataclass
class W2V2Config:
    """The subset of HF `Wav2Vec2Config` this path reads; defaults = facebook/wav2vec2-large-xlsr-53."""
    conv_dim: Tuple[int, ...] = (512,) * 7
    conv_kernel: Tuple[int, ...] = (10, 3, 3, 3, 3, 2, 2)
    conv_stride: Tuple[int, ...] = (5, 2, 2, 2, 2, 2, 2)
    conv_bias: bool = True
    hidden_size: int = 1024
    num_attention_heads: int = 16
    intermediate_size: int = 4096
    num_hidden_layers: int = 24
    num_conv_pos_embeddings: int = 128
    num_conv_pos_embedding_groups: int = 16
    layer_norm_eps: float = 1e-5

    def hf_kwargs(self) -> dict:
        return dict(conv_dim=list(self.conv_dim), conv_kernel=list(self.conv_kernel), conv_stride=list(self.conv_stride),
                    conv_bias=self.conv_bias, hidden_size=self.hidden_size, num_attention_heads=self.num_attention_heads,
                    intermediate_size=self.intermediate_size, num_hidden_layers=self.num_hidden_layers,
                    num_conv_pos_embeddings=self.num_conv_pos_embeddings,
                    num_conv_pos_embedding_groups=self.num_conv_pos_embedding_groups, layer_norm_eps=self.layer_norm_eps,
                    feat_extract_norm="layer", do_stable_layer_norm=True, num_feat_extract_layers=len(self.conv_dim),
                    hidden_act="gelu", feat_extract_activation="gelu", apply_spec_augment=False)


def pos_conv_weight(sd: Dict[str, torch.Tensor]) -> torch.Tensor:
    """Effective weight of the weight-normed positional conv (`nn.utils.weight_norm(conv, dim=2)`): w = g * v / ||v||, the
    norm taken over (out, in) per kernel position.  Both the 4.24-era (`weight_g`/`weight_v`) and the parametrize-era
    (`parametrizations.weight.original0/1`) key names are accepted."""
    p = "encoder.pos_conv_embed.conv."
    if p + "weight_g" in sd:
        g, v = sd[p + "weight_g"], sd[p + "weight_v"]
    else:
        g, v = sd[p + "parametrizations.weight.original0"], sd[p + "parametrizations.weight.original1"]
    norm = v.pow(2).sum(dim=(0, 1), keepdim=True).sqrt()
    return g * v / norm


def feature_encoder(sd, cfg: W2V2Config, wave: torch.Tensor) -> torch.Tensor:
    """(B, L) waveform -> (B, frames, 512).  HF Wav2Vec2FeatureEncoder with Wav2Vec2LayerNormConvLayer blocks."""
    h = wave[:, None]
    for i, (k, s) in enumerate(zip(cfg.conv_kernel, cfg.conv_stride)):
        p = f"feature_extractor.conv_layers.{i}."
        h = F.conv1d(h, sd[p + "conv.weight"], sd.get(p + "conv.bias") if cfg.conv_bias else None, stride=s)
        h = F.layer_norm(h.transpose(1, 2), (h.shape[1],), sd[p + "layer_norm.weight"], sd[p + "layer_norm.bias"], 1e-5)
        h = F.gelu(h.transpose(1, 2))
    return h.transpose(1, 2)


def attention(sd, cfg: W2V2Config, p: str, x: torch.Tensor) -> torch.Tensor:
    B, T, H = x.shape
    nh, hd = cfg.num_attention_heads, cfg.hidden_size // cfg.num_attention_heads
    q = F.linear(x, sd[p + "q_proj.weight"], sd[p + "q_proj.bias"]).view(B, T, nh, hd).transpose(1, 2)
    k = F.linear(x, sd[p + "k_proj.weight"], sd[p + "k_proj.bias"]).view(B, T, nh, hd).transpose(1, 2)
    v = F.linear(x, sd[p + "v_proj.weight"], sd[p + "v_proj.bias"]).view(B, T, nh, hd).transpose(1, 2)
    w = torch.softmax(torch.matmul(q, k.transpose(2, 3)) * hd ** -0.5, dim=-1)
    o = torch.matmul(w, v).transpose(1, 2).reshape(B, T, H)
    return F.linear(o, sd[p + "out_proj.weight"], sd[p + "out_proj.bias"])


def hidden_states(sd, cfg: W2V2Config, wave: torch.Tensor) -> List[torch.Tensor]:
    """All `output_hidden_states` of Wav2Vec2Model(eval) for a (B, L) waveform: L+1 tensors (B, frames, H) — the input of
    every encoder layer, then the final LayerNorm of the last layer's output (Wav2Vec2EncoderStableLayerNorm)."""
    eps = cfg.layer_norm_eps
    feats = feature_encoder(sd, cfg, wave)
    h = F.layer_norm(feats, (feats.shape[-1],), sd["feature_projection.layer_norm.weight"],
                     sd["feature_projection.layer_norm.bias"], eps)
    h = F.linear(h, sd["feature_projection.projection.weight"], sd["feature_projection.projection.bias"])
    K = cfg.num_conv_pos_embeddings
    pos = F.conv1d(h.transpose(1, 2), pos_conv_weight(sd), sd["encoder.pos_conv_embed.conv.bias"], padding=K // 2,
                   groups=cfg.num_conv_pos_embedding_groups)
    if K % 2 == 0:
        pos = pos[:, :, :-1]                       # Wav2Vec2SamePadLayer
    h = h + F.gelu(pos).transpose(1, 2)
    out = []
    for i in range(cfg.num_hidden_layers):
        out.append(h)
        p = f"encoder.layers.{i}."
        a = F.layer_norm(h, (h.shape[-1],), sd[p + "layer_norm.weight"], sd[p + "layer_norm.bias"], eps)
        h = h + attention(sd, cfg, p + "attention.", a)
        f = F.layer_norm(h, (h.shape[-1],), sd[p + "final_layer_norm.weight"], sd[p + "final_layer_norm.bias"], eps)
        f = F.gelu(F.linear(f, sd[p + "feed_forward.intermediate_dense.weight"], sd[p + "feed_forward.intermediate_dense.bias"]))
        h = h + F.linear(f, sd[p + "feed_forward.output_dense.weight"], sd[p + "feed_forward.output_dense.bias"])
    out.append(F.layer_norm(h, (h.shape[-1],), sd["encoder.layer_norm.weight"], sd["encoder.layer_norm.bias"], eps))
    return out


def last_four_mean(sd, cfg: W2V2Config, wave: torch.Tensor) -> torch.Tensor:
    """`_process_chunk` of wav2vec_util.py:15-20: mean of hidden_states[-4:] -> (B, frames, H)."""
    return torch.stack(hidden_states(sd, cfg, wave)[-4:]).mean(dim=0)


def chunk_bounds(n_samples: int, n_chunks: int = 10) -> List[Tuple[int, int]]:
    """`np.array_split(range(n), 10)` (wav2vec_util.py:24) as [start, stop) pairs."""
    q, r = divmod(n_samples, n_chunks)
    sizes = [q + 1] * r + [q] * (n_chunks - r)
    edges = np.concatenate([[0], np.cumsum(sizes)])
    return [(int(edges[i]), int(edges[i + 1])) for i in range(n_chunks)]


def w2v_last_four_layers_avg(sd, cfg: W2V2Config, waveform: torch.Tensor, n_chunks: int = 10) -> torch.Tensor:
    """`getW2VLastFourLayersAvg` (wav2vec_util.py:14-32): (1, L) waveform -> (H, total frames)."""
    embs = []
    with torch.no_grad():
        for a, b in chunk_bounds(waveform.shape[-1], n_chunks):
            embs.append(last_four_mean(sd, cfg, waveform[0, a:b].unsqueeze(0)).squeeze(0))
    return torch.vstack(embs).t()


def n_frames(n_samples: int, cfg: W2V2Config) -> int:
    n = n_samples
    for k, s in zip(cfg.conv_kernel, cfg.conv_stride):
        n = (n - k) // s + 1
    return n


def resample_fft(x: np.ndarray, up: float, npad: int = 100) -> np.ndarray:
    """FFT resampling of the last axis by the factor `up` (gwilliams2022.py:369-373 calls `mne.filter.resample(x, up=...)`
    with mne's defaults npad=100, window="boxcar", pad="reflect_limited"): pad each side with the odd (point-reflected)
    extension of `npad` samples, rfft, truncate / zero-pad the spectrum to the new length (Nyquist bin shared), irfft
    scaled by the length ratio, remove round(npad * up) samples of padding.  float64 throughout, as the reference calls
    it.  mne is absent here: parity unpinned (module docstring)."""
    x = np.asarray(x, dtype=np.float64)
    n = x.shape[-1]
    new_len = int(round(n * up))
    xp = smart_pad(x, npad)
    m = xp.shape[-1]
    m_new = max(int(round(m * up)), 1)
    X = np.fft.rfft(xp, axis=-1)
    use = min(m, m_new)
    if use % 2 == 0:                               # the Nyquist bin of the shorter length is shared by +f and -f
        X[..., use // 2] *= 2.0 if m_new < m else 0.5
    Y = np.zeros(x.shape[:-1] + (m_new // 2 + 1,), dtype=np.complex128)
    keep = min(X.shape[-1], Y.shape[-1])
    Y[..., :keep] = X[..., :keep]
    y = np.fft.irfft(Y, n=m_new, axis=-1) * (m_new / m)
    off = int(round(npad * up))
    return y[..., off:off + new_len]


def smart_pad(x: np.ndarray, npad: int) -> np.ndarray:
    """mne's "reflect_limited" padding: the odd extension about each end point, zero-filled where the signal is shorter."""
    n = x.shape[-1]
    z = np.zeros(x.shape[:-1] + (max(npad - n + 1, 0),), dtype=x.dtype)
    left = 2 * x[..., :1] - x[..., min(npad, n - 1):0:-1]
    right = 2 * x[..., -1:] - x[..., -2:-min(npad, n - 1) - 2:-1]
    return np.concatenate([z, left, x, right, z], axis=-1)


def random_state_dict(cfg: W2V2Config, seed: int = 0, scale: float = 1.0) -> Dict[str, torch.Tensor]:
    """Seeded random weights with the HF key names and shapes (tests only; LayerNorm weights near 1)."""
    g = torch.Generator().manual_seed(seed)
    sd = {}

    def lin(name, out_f, in_f):
        sd[name + ".weight"] = torch.randn(out_f, in_f, generator=g) * (scale / np.sqrt(in_f))
        sd[name + ".bias"] = torch.randn(out_f, generator=g) * 0.1

    def ln(name, c):
        sd[name + ".weight"] = 1.0 + 0.1 * torch.randn(c, generator=g)
        sd[name + ".bias"] = 0.1 * torch.randn(c, generator=g)

    cin = 1
    for i, (c, k) in enumerate(zip(cfg.conv_dim, cfg.conv_kernel)):
        p = f"feature_extractor.conv_layers.{i}."
        sd[p + "conv.weight"] = torch.randn(c, cin, k, generator=g) * (scale / np.sqrt(cin * k))
        if cfg.conv_bias:
            sd[p + "conv.bias"] = torch.randn(c, generator=g) * 0.1
        ln(p + "layer_norm", c)
        cin = c
    ln("feature_projection.layer_norm", cfg.conv_dim[-1])
    lin("feature_projection.projection", cfg.hidden_size, cfg.conv_dim[-1])
    H, K, G = cfg.hidden_size, cfg.num_conv_pos_embeddings, cfg.num_conv_pos_embedding_groups
    sd["encoder.pos_conv_embed.conv.weight_g"] = 0.5 + torch.rand(1, 1, K, generator=g)
    sd["encoder.pos_conv_embed.conv.weight_v"] = torch.randn(H, H // G, K, generator=g) * 0.05
    sd["encoder.pos_conv_embed.conv.bias"] = torch.randn(H, generator=g) * 0.1
    ln("encoder.layer_norm", H)
    for i in range(cfg.num_hidden_layers):
        p = f"encoder.layers.{i}."
        for n in ("q_proj", "k_proj", "v_proj", "out_proj"):
            lin(p + "attention." + n, H, H)
        ln(p + "layer_norm", H)
        lin(p + "feed_forward.intermediate_dense", cfg.intermediate_size, H)
        lin(p + "feed_forward.output_dense", H, cfg.intermediate_size)
        ln(p + "final_layer_norm", H)
    return sd
