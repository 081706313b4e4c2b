"""Frozen wav2vec 2.0 speech embedder on MI355X.

Replaces what `/root/reference/speech_decoding/utils/wav2vec_util.py:8-32` does with HuggingFace's `Wav2Vec2Model`
(third party, `transformers==4.24.0` in the reference; weights `facebook/wav2vec2-large-xlsr-53`, config.yaml:30):
chunked inference, mean of the last four hidden states, (features, frames) output, and the FFT resampling to the
brain rate that follows it in `dataclass/gwilliams2022.py:369-373`.

Architecture supported: the one xlsr-53 selects — layer-norm feature encoder (`feat_extract_norm="layer"`), stable
layer-norm transformer (`do_stable_layer_norm=True`), head dimension 64.  Other variants are refused, not approximated.

How it maps to the hardware path (every contraction is `sda_conv_gemm`, the MFMA implicit-GEMM of the training path):
  * feature-encoder layer 0 (1 -> 512 channels, k = 10, stride 5) + LayerNorm + GELU: one kernel (`sda_w2v_conv0`);
  * layers 1..6 (k = 3 / 2, stride 2): a frame-major activation buffer [T][512] read with row pitch stride*512 and row
    length k*512 IS the im2col matrix of the strided conv, so each layer is one GEMM on an overlapping-row view, then
    `sda_layernorm_rows` (+GELU);
  * positional conv (k = 128, 16 groups): channels regrouped per group, then per group the same overlapping-row trick
    (row pitch 64, row length 128*64) -> 16 GEMMs with bias + GELU in the epilogue, merged back with the residual;
  * attention: Q|K in one GEMM, V^T in one GEMM with the operand roles swapped (weights as rows, activations as columns;
    v's bias is folded into the output projection's bias: softmax rows sum to one), `sda_w2v_attention` per layer;
  * out-projection / FFN: GEMMs with bias, GELU and the residual add in their epilogues.
Weights are packed once at construction (the model is frozen).  Parity: DESIGN.md §1 row f4 — architecture pinned against
`transformers` on seeded random weights, pretrained weights unobtainable offline ("parity unpinned" for real embeddings).
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Dict, List, Optional, Tuple

import numpy as np
import torch

from . import lib as L
from . import ops


@dataclass
class Wav2Vec2Config:
    """The fields of HF `Wav2Vec2Config` this path reads; defaults = facebook/wav2vec2-large-xlsr-53."""
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

    @classmethod
    def from_hf(cls, cfg) -> "Wav2Vec2Config":
        if getattr(cfg, "feat_extract_norm", "layer") != "layer" or not getattr(cfg, "do_stable_layer_norm", True):
            raise ValueError("only the layer-norm feature encoder / stable-layer-norm encoder variant (xlsr-53) is built")
        if getattr(cfg, "hidden_act", "gelu") != "gelu" or getattr(cfg, "feat_extract_activation", "gelu") != "gelu":
            raise ValueError("only GELU activations are built")
        return cls(tuple(cfg.conv_dim), tuple(cfg.conv_kernel), tuple(cfg.conv_stride), bool(cfg.conv_bias), cfg.hidden_size,
                   cfg.num_attention_heads, cfg.intermediate_size, cfg.num_hidden_layers, cfg.num_conv_pos_embeddings,
                   cfg.num_conv_pos_embedding_groups, cfg.layer_norm_eps)

    def n_frames(self, n_samples: int) -> int:
        n = n_samples
        for k, s in zip(self.conv_kernel, self.conv_stride):
            n = (n - k) // s + 1
        return n


def chunk_bounds(n_samples: int, n_chunks: int = 10) -> List[Tuple[int, int]]:
    """[start, stop) of `np.array_split(range(n_samples), n_chunks)` (wav2vec_util.py:24)."""
    q, r = divmod(n_samples, n_chunks)
    edges = np.concatenate([[0], np.cumsum([q + 1] * r + [q] * (n_chunks - r))])
    return [(int(edges[i]), int(edges[i + 1])) for i in range(n_chunks)]


class Wav2Vec2Embedder:
    """`Wav2Vec2Model(...).eval()` restricted to what the reference reads from it: hidden states -> last-four mean."""

    def __init__(self, state_dict: Dict[str, torch.Tensor], config: Wav2Vec2Config, dtype=torch.bfloat16, device="cuda:0"):
        L.load()                                    # fail loudly if the HIP extension is missing
        if dtype not in ops.COMPUTE_DTYPES:
            raise ValueError(f"compute dtype {dtype} not supported")
        cfg = self.cfg = config
        self.dtype, self.device = dtype, torch.device(device)
        if cfg.hidden_size % cfg.num_attention_heads or cfg.hidden_size // cfg.num_attention_heads != 64:
            raise ValueError("head dimension must be 64")
        if cfg.hidden_size % cfg.num_conv_pos_embedding_groups:
            raise ValueError("hidden size must divide into the positional-conv groups")
        if len(set(cfg.conv_dim)) != 1 or cfg.num_hidden_layers < 4:
            raise ValueError("feature-encoder layers must share one width; at least four encoder layers are needed")
        self.es = 4 if dtype == torch.float32 else 2
        self._ws: Dict[tuple, dict] = {}
        self.taps: Optional[dict] = None             # diagnostics: set to a dict to receive (T, C) copies of the stages
        # A chunk is ~260 small launches (frames, not batch, are the GEMMs' rows): launch-bound from Python.  The chunk
        # lengths np.array_split produces differ by at most one sample, so one or two HIP graphs replay all ten chunks.
        self.use_graphs = True
        # ... and the chunks are independent: `streams_in_flight` of them run at a time on HIP streams of their own (each with
        # its own buffers and graphs), so one chunk's 24-workgroup GEMMs share the chip with the others'
        self.streams_in_flight = 4
        self._streams: List[torch.cuda.Stream] = []
        self._pack({k: v.detach().to(torch.float64).cpu() for k, v in state_dict.items() if v.is_floating_point()})

    @classmethod
    def from_hf(cls, model, dtype=torch.bfloat16, device="cuda:0") -> "Wav2Vec2Embedder":
        """From a `transformers.Wav2Vec2Model` instance (what `load_wav2vec_model` returns, wav2vec_util.py:8-11)."""
        return cls(model.state_dict(), Wav2Vec2Config.from_hf(model.config), dtype, device)

    # ------------------------------------------------------------------ weights
    def _dev(self, t, dtype=torch.float32):
        return t.to(dtype).to(self.device).contiguous()

    def _linear(self, w: torch.Tensor, b: Optional[torch.Tensor]):
        """[out][in] fp64 -> packed (1, 1, out_p, in_p) compute-dtype operand + padded fp32 bias."""
        out_f, in_f = w.shape
        wp = torch.zeros(L.pad_channels(out_f), L.pad_channels(in_f), dtype=torch.float64)
        wp[:out_f, :in_f] = w
        bp = torch.zeros(wp.shape[0], dtype=torch.float64)
        if b is not None:
            bp[:out_f] = b
        return self._dev(wp, self.dtype).view(1, 1, *wp.shape), self._dev(bp)

    def _pack(self, sd):
        cfg, P = self.cfg, {}
        C0 = cfg.conv_dim[0]
        self.Cp, self.Hp, self.Fp = L.pad_channels(C0), L.pad_channels(cfg.hidden_size), L.pad_channels(cfg.intermediate_size)
        H = cfg.hidden_size
        # feature encoder
        p = "feature_extractor.conv_layers.0."
        P["c0.w"] = self._dev(sd[p + "conv.weight"].reshape(C0, cfg.conv_kernel[0]))
        P["c0.b"] = self._dev(sd[p + "conv.bias"]) if cfg.conv_bias else None
        P["c0.g"], P["c0.be"] = self._dev(sd[p + "layer_norm.weight"]), self._dev(sd[p + "layer_norm.bias"])
        for i in range(1, len(cfg.conv_dim)):
            p = f"feature_extractor.conv_layers.{i}."
            w = sd[p + "conv.weight"]                                   # [co][ci][k]
            k = w.shape[2]
            we = torch.zeros(C0, k, self.Cp, dtype=torch.float64)      # row co = [tap][ci padded]: matches the overlapping-row view
            we[:, :, :C0] = w.permute(0, 2, 1)
            P[f"c{i}.w"], P[f"c{i}.b"] = self._linear(we.reshape(C0, k * self.Cp), sd.get(p + "conv.bias") if cfg.conv_bias else None)
            P[f"c{i}.g"], P[f"c{i}.be"] = self._dev(sd[p + "layer_norm.weight"]), self._dev(sd[p + "layer_norm.bias"])
        P["fp.g"], P["fp.be"] = self._dev(sd["feature_projection.layer_norm.weight"]), self._dev(sd["feature_projection.layer_norm.bias"])
        P["fp.w"], P["fp.b"] = self._linear(sd["feature_projection.projection.weight"], sd["feature_projection.projection.bias"])
        # positional conv: weight_norm(dim=2) -> w = g * v / ||v||_(out,in) per tap; per group [co][tap][ci padded]
        pc = "encoder.pos_conv_embed.conv."
        if pc + "weight_g" in sd:
            g, v = sd[pc + "weight_g"], sd[pc + "weight_v"]
        else:
            g, v = sd[pc + "parametrizations.weight.original0"], sd[pc + "parametrizations.weight.original1"]
        w = g * v / v.pow(2).sum(dim=(0, 1), keepdim=True).sqrt()       # [H][gw][K]
        G, K = cfg.num_conv_pos_embedding_groups, cfg.num_conv_pos_embeddings
        gw = H // G
        self.gw, self.gwp = gw, L.pad_channels(gw)
        if gw % 8:
            raise ValueError("positional-conv group width must be a multiple of 8")
        wg = torch.zeros(G, self.gwp, K, self.gwp, dtype=torch.float64)
        wg[:, :gw, :, :gw] = w.view(G, gw, gw, K).permute(0, 1, 3, 2)
        bg = torch.zeros(G, self.gwp, dtype=torch.float64)
        bg[:, :gw] = sd[pc + "bias"].view(G, gw)
        P["pos.w"] = self._dev(wg.reshape(G, self.gwp, K * self.gwp), self.dtype)
        P["pos.bflat"] = self._dev(sd[pc + "bias"])                     # [H]: added (with the GELU) when the groups are merged
        P["enc.g"], P["enc.be"] = self._dev(sd["encoder.layer_norm.weight"]), self._dev(sd["encoder.layer_norm.bias"])
        for i in range(cfg.num_hidden_layers):
            p, q = f"encoder.layers.{i}.", f"l{i}."
            a = p + "attention."
            P[q + "ln1.g"], P[q + "ln1.be"] = self._dev(sd[p + "layer_norm.weight"]), self._dev(sd[p + "layer_norm.bias"])
            P[q + "ln2.g"], P[q + "ln2.be"] = self._dev(sd[p + "final_layer_norm.weight"]), self._dev(sd[p + "final_layer_norm.bias"])
            P[q + "qk.w"], P[q + "qk.b"] = self._linear(torch.cat([sd[a + "q_proj.weight"], sd[a + "k_proj.weight"]]),
                                                        torch.cat([sd[a + "q_proj.bias"], sd[a + "k_proj.bias"]]))
            wv = torch.zeros(H, self.Hp, dtype=torch.float64)
            wv[:, :H] = sd[a + "v_proj.weight"]
            P[q + "v.w"] = self._dev(wv, self.dtype)                     # rows of the swapped-role GEMM that yields V^T
            # softmax rows sum to one: P (V + 1 b_v^T) = P V + b_v^T, so v's bias moves into the output projection's
            bo = sd[a + "out_proj.bias"] + sd[a + "out_proj.weight"] @ sd[a + "v_proj.bias"]
            P[q + "o.w"], P[q + "o.b"] = self._linear(sd[a + "out_proj.weight"], bo)
            P[q + "f1.w"], P[q + "f1.b"] = self._linear(sd[p + "feed_forward.intermediate_dense.weight"],
                                                        sd[p + "feed_forward.intermediate_dense.bias"])
            P[q + "f2.w"], P[q + "f2.b"] = self._linear(sd[p + "feed_forward.output_dense.weight"],
                                                        sd[p + "feed_forward.output_dense.bias"])
        self.P = P

    # ------------------------------------------------------------------ workspace (per frame count)
    def _workspace(self, T: int, frames: List[int], slot: int = 0) -> dict:
        key = (T, tuple(frames), slot)
        ws = self._ws.get(key)
        if ws is None:
            if len(self._ws) >= 4 * max(1, self.streams_in_flight):
                self._ws.clear()
            cfg, dt, dev = self.cfg, self.dtype, self.device
            rows = lambda t, c: ops.new_rows(1, t, c, dt, dev)
            G, K = cfg.num_conv_pos_embedding_groups, cfg.num_conv_pos_embeddings
            lead = K // 2 + L.ROW_PAD
            ws = dict(feat=[rows(t, self.Cp) for t in frames], h=[rows(T, self.Hp) for _ in range(8)], a=rows(T, self.Hp),
                      o=rows(T, self.Hp), qk=rows(T, 2 * self.Hp), u=rows(T, self.Fp),
                      vt=torch.zeros((cfg.hidden_size, (T + 63) // 64 * 64), dtype=dt, device=dev),
                      xg=torch.zeros((G, lead + T + K + 2 * L.ROW_PAD, self.gwp), dtype=dt, device=dev),
                      lead=lead)
            ws["yg"] = torch.zeros_like(ws["xg"])              # same row geometry: the G groups are ONE batched GEMM's samples
            # split-K scratch of ops.linear_rows: ksplit * T * Cout_p <= (256 / tiles) * T * Cout_p ~ 256 * 128 * 160 floats
            ws["partial"] = torch.empty(2 * 256 * 128 * 160, dtype=torch.float32, device=dev)
            ws["gidx"] = torch.arange(G, dtype=torch.int32, device=dev)
            self._ws[key] = ws
        return ws

    def release_workspace(self):
        self._ws.clear()

    def eval(self):
        """`wav2vec.eval()` of the call site (gwilliams2022.py:330): the embedder is inference-only, nothing to switch."""
        return self

    # ------------------------------------------------------------------ forward
    def _forward(self, wave: torch.Tensor, want_all: bool, slot: int = 0):
        """wave: 1-D fp32 on the device.  Returns (T, [row-layout hidden states kept]) — all of them when want_all (then each
        is copied out), otherwise the last four."""
        cfg, P, dt, es = self.cfg, self.P, self.dtype, self.es
        n = wave.numel()
        frames, t = [], n
        for k, s in zip(cfg.conv_kernel, cfg.conv_stride):
            t = (t - k) // s + 1
            frames.append(t)
        T = frames[-1]
        if T < 1:
            raise ValueError(f"waveform of {n} samples is shorter than the feature encoder's receptive field")
        ws = self._workspace(T, frames, slot)
        Cp, Hp, PADR = self.Cp, self.Hp, L.ROW_PAD
        C0, H = cfg.conv_dim[0], cfg.hidden_size
        # ---- feature encoder (HF Wav2Vec2FeatureEncoder, layer-norm conv layers)
        f = ws["feat"]
        ops.w2v_conv0(wave, P["c0.w"], P["c0.b"], P["c0.g"], P["c0.be"], f[0], frames[0], C0, cfg.conv_kernel[0], cfg.conv_stride[0])
        if self.taps is not None:
            self.taps["feat0"] = ops.rows_view(f[0], 1, C0, frames[0])[0].t().to(torch.float32, copy=True)
        for i in range(1, len(frames)):
            k, s = cfg.conv_kernel[i], cfg.conv_stride[i]
            # view row PAD + t of the previous layer's buffer = its rows PAD + s t ... PAD + s t + k - 1, contiguous
            x_view = f[i - 1].data_ptr() - PADR * (s - 1) * Cp * es
            ops.gemm_view(x_view, P[f"c{i}.w"].data_ptr(), f[i].data_ptr(), rows=frames[i], K=k * Cp, Cout_p=Cp, x_pitch=s * Cp,
                          w_pitch=k * Cp, x_row0=PADR, x_rows_limit=PADR + frames[i], dtype=dt, bias=P[f"c{i}.b"])
            ops.layernorm_rows(f[i], f[i], P[f"c{i}.g"], P[f"c{i}.be"], frames[i], C0, 1e-5, gelu=True)
            if self.taps is not None:
                self.taps[f"feat{i}"] = ops.rows_view(f[i], 1, C0, frames[i])[0].t().to(torch.float32, copy=True)
        # ---- feature projection
        ops.layernorm_rows(f[-1], f[-1], P["fp.g"], P["fp.be"], T, C0, cfg.layer_norm_eps)
        pool, a, o, qk, u, vt = ws["h"], ws["a"], ws["o"], ws["qk"], ws["u"], ws["vt"]
        nxt = [0]

        def fresh():
            b = pool[nxt[0] % len(pool)]
            nxt[0] += 1
            return b
        part = ws["partial"]
        h = ops.linear_rows(f[-1], P["fp.w"], fresh(), T, bias=P["fp.b"], scratch=part)
        if self.taps is not None:
            self.taps["proj"] = self._copy_out(h, T)
        # ---- positional conv embedding: h = h + GELU(conv(h))
        G, K, gw, gwp, lead = cfg.num_conv_pos_embedding_groups, cfg.num_conv_pos_embeddings, self.gw, self.gwp, ws["lead"]
        xg, yg = ws["xg"], ws["yg"]
        ops.w2v_group_split(h, xg, T, gw, G, lead)
        # all G groups in ONE launch: the groups are the GEMM's "samples" (stride = a group's rows), widx selects the weights;
        # bias and GELU are per group, so they move to the merge
        grows = xg.shape[1]
        ops.gemm_view(xg.data_ptr(), P["pos.w"].data_ptr(), yg.data_ptr(), rows=T, K=K * gwp, Cout_p=gwp, x_pitch=gwp,
                      w_pitch=K * gwp, x_row0=PADR, x_rows_limit=G * grows, dtype=dt, batch=G, sample_rows=grows, widx=ws["gidx"])
        h = ops.w2v_group_merge_add(h, yg, fresh(), T, gw, G, bias=P["pos.bflat"], gelu=True)
        # ---- encoder layers (HF Wav2Vec2EncoderLayerStableLayerNorm)
        states = []
        keep_from = 0 if want_all else cfg.num_hidden_layers - 3
        heads = cfg.num_attention_heads
        Tp = vt.shape[1]
        for i in range(cfg.num_hidden_layers):
            q = f"l{i}."
            if i >= keep_from:
                states.append(self._copy_out(h, T) if want_all else h)
            ops.layernorm_rows(h, a, P[q + "ln1.g"], P[q + "ln1.be"], T, H, cfg.layer_norm_eps)
            ops.linear_rows(a, P[q + "qk.w"], qk, T, bias=P[q + "qk.b"], scratch=part)
            # V^T [H][Tp] = W_v [H][Hp] . a^T: the weights are the GEMM's rows, the activations (frames PAD .. PAD + Tp) its columns
            ops.gemm_view(P[q + "v.w"].data_ptr(), a.data_ptr() + PADR * Hp * es, vt.data_ptr(), rows=H, K=Hp, Cout_p=Tp, x_pitch=Hp,
                          w_pitch=Hp, x_row0=0, x_rows_limit=H, dtype=dt)
            ops.w2v_attention(qk.data_ptr(), qk.data_ptr() + Hp * es, vt, o, T, heads, 64, 2 * Hp, 64 ** -0.5)
            h_mid = ops.linear_rows(o, P[q + "o.w"], fresh(), T, bias=P[q + "o.b"], res=h, scratch=part)
            ops.layernorm_rows(h_mid, a, P[q + "ln2.g"], P[q + "ln2.be"], T, H, cfg.layer_norm_eps)
            ops.linear_rows(a, P[q + "f1.w"], u, T, bias=P[q + "f1.b"], gelu=True, scratch=part)
            h = ops.linear_rows(u, P[q + "f2.w"], fresh(), T, bias=P[q + "f2.b"], res=h_mid, scratch=part)
        last = ops.layernorm_rows(h, fresh(), P["enc.g"], P["enc.be"], T, H, cfg.layer_norm_eps)
        states.append(self._copy_out(last, T) if want_all else last)
        return T, states

    def _copy_out(self, buf, T) -> torch.Tensor:
        H = self.cfg.hidden_size
        return ops.rows_view(buf, 1, H, T)[0].t().to(torch.float32, copy=True, memory_format=torch.contiguous_format)   # (T, H)

    @torch.no_grad()
    def hidden_states(self, wave: torch.Tensor) -> List[torch.Tensor]:
        """All `output_hidden_states` of the model for one 1-D waveform: num_hidden_layers + 1 tensors (T, H), fp32."""
        return self._forward(self._wave(wave), True)[1]

    def _last_four_eager(self, wave: torch.Tensor, slot: int = 0) -> torch.Tensor:
        T, st = self._forward(wave, False, slot)
        return ops.w2v_mean4(st[0], st[1], st[2], st[3], T, self.cfg.hidden_size)

    @torch.no_grad()
    def last_four_mean(self, wave: torch.Tensor, slot: int = 0) -> torch.Tensor:
        """`_process_chunk` (wav2vec_util.py:15-20): mean of hidden_states[-4:], (T, H) fp32.  Runs on the current stream with
        buffer set `slot` (concurrent chunks need different slots)."""
        wave = self._wave(wave)
        if not self.use_graphs or self.taps is not None:
            return self._last_four_eager(wave, slot)
        n = wave.numel()
        frames, t = [], n
        for k, s in zip(self.cfg.conv_kernel, self.cfg.conv_stride):
            t = (t - k) // s + 1
            frames.append(t)
        if frames[-1] < 1:
            return self._last_four_eager(wave, slot)     # raises the length error
        graphs = self._workspace(frames[-1], frames, slot).setdefault("graphs", {})      # a graph lives and dies with its buffers
        entry = graphs.get(n)
        if entry is None:
            static_in = wave.clone()
            self._last_four_eager(static_in, slot)        # warm-up outside the capture (first-use allocations, lazy loads)
            torch.cuda.synchronize(self.device)
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                static_out = self._last_four_eager(static_in, slot)
            if len(graphs) >= 4:
                graphs.clear()
            entry = graphs[n] = (graph, static_in, static_out)
        graph, static_in, static_out = entry
        static_in.copy_(wave)
        graph.replay()
        return static_out.clone()

    def _wave(self, wave: torch.Tensor) -> torch.Tensor:
        if wave.dim() != 1:
            raise ValueError("expected a 1-D waveform")
        return wave.to(device=self.device, dtype=torch.float32).contiguous()

    @torch.no_grad()
    def embed(self, waveform: torch.Tensor, n_chunks: int = 10) -> torch.Tensor:
        """`getW2VLastFourLayersAvg(wav2vec, waveform)` (wav2vec_util.py:14-32): (1, L) waveform -> (H, frames) fp32."""
        if waveform.dim() != 2:
            raise ValueError("expected a (1, L) waveform")
        wave = self._wave(waveform[0])
        bounds = chunk_bounds(wave.numel(), n_chunks)
        n = max(1, min(self.streams_in_flight, len(bounds)))
        if n == 1 or self.taps is not None:
            return torch.vstack([self.last_four_mean(wave[a:b]) for a, b in bounds]).t()
        while len(self._streams) < n:
            self._streams.append(torch.cuda.Stream(device=self.device))
        main = torch.cuda.current_stream(self.device)
        ready = torch.cuda.Event()
        ready.record(main)
        out = []
        for i, (a, b) in enumerate(bounds):
            st = self._streams[i % n]
            if i < n:
                st.wait_event(ready)                     # the waveform was produced on the caller's stream
                wave.record_stream(st)
            with torch.cuda.stream(st):
                out.append(self.last_four_mean(wave[a:b], slot=i % n))
        for st in self._streams[:n]:
            done = torch.cuda.Event()
            done.record(st)
            main.wait_event(done)
        for o in out:
            o.record_stream(main)
        return torch.vstack(out).t()

    __call__ = embed


def resample_fft(x: torch.Tensor, up: float, npad: int = 100) -> torch.Tensor:
    """FFT resampling of the last axis by `up`, float64, on the tensor's device (rocFFT through torch.fft) — what
    `mne.filter.resample(embeddings.astype(float64), up=brain_rate / rate_after_wav2vec, axis=-1)` computes
    (gwilliams2022.py:369-373; mne defaults npad=100, boxcar window, "reflect_limited" padding): odd extension of `npad`
    samples at both ends, rfft, truncate / zero-extend the spectrum (shared Nyquist bin), irfft, drop the padding."""
    x = x.to(torch.float64)
    n = x.shape[-1]
    new_len = int(round(n * up))
    p = min(npad, n - 1)
    z = x.new_zeros(x.shape[:-1] + (max(npad - n + 1, 0),))
    left = 2 * x[..., :1] - x[..., 1:p + 1].flip(-1)
    right = 2 * x[..., -1:] - x[..., n - 1 - p:n - 1].flip(-1)
    xp = torch.cat([z, left, x, right, z], dim=-1)
    m = xp.shape[-1]
    m_new = max(int(round(m * up)), 1)
    X = torch.fft.rfft(xp, dim=-1)
    use = min(m, m_new)
    if use % 2 == 0:
        X[..., use // 2] *= 2.0 if m_new < m else 0.5
    Y = X.new_zeros(x.shape[:-1] + (m_new // 2 + 1,))
    keep = min(X.shape[-1], Y.shape[-1])
    Y[..., :keep] = X[..., :keep]
    y = torch.fft.irfft(Y, n=m_new, dim=-1) * (m_new / m)
    off = int(round(npad * up))
    return y[..., off:off + new_len]
