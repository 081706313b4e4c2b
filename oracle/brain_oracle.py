"""CPU oracle for the contrastive training hot path — TEST INFRASTRUCTURE ONLY.

This file is a plain-PyTorch (CPU, fp32 or fp64) *restatement* of the algorithm of the
reference's `speech_decoding/models.py` + `speech_decoding/utils/loss.py`.  It is the checker
for the HIP path, never the product: only `tests/`, `__graft_entry__.smoke()` and
`bench.py`'s `cpu_baseline` leg may import it.  The product (`speech_decoding_amd/`) must never
import from `oracle/`.

Parity pin: `tests/golden/*.npz` were produced by importing and running the reference itself in
the build container (`tests/golden/make_golden.py`, committed); `tests/test_oracle_golden.py`
checks every function below against them.

It is written functionally over a flat parameter dict `P` that uses the reference's
`state_dict()` key names, so a reference checkpoint can be fed in unchanged.
All `file:line` citations are into the reference tree.
"""
from __future__ import annotations

import math
from typing import Dict, Optional, Tuple

import numpy as np
import torch
import torch.nn.functional as TF

Tensor = torch.Tensor
BN_EPS = 1e-5        # nn.BatchNorm1d default, models.py:135,143
BN_MOMENTUM = 0.1    # nn.BatchNorm1d default


# --------------------------------------------------------------------------------------
# geometry (layout.py:38-41) and the Fourier tables (models.py:20-40)
# --------------------------------------------------------------------------------------
def normalise_positions(raw_xy: np.ndarray) -> Tensor:
    """layout.py:38-41 — min-max normalise each axis, then shrink into [0.1, 0.9]."""
    raw_xy = np.asarray(raw_xy, dtype=np.float64)
    lo, hi = raw_xy.min(axis=0), raw_xy.max(axis=0)
    loc = (raw_xy - lo) / (hi - lo)
    loc = loc * 0.8 + 0.1
    return torch.from_numpy(loc.astype(np.float32))


def synthetic_positions(num_channels: int, seed: int = 0) -> Tensor:
    """Seeded stand-in for the MNE sensor lookup (layout.py:9-35 is out of scope)."""
    rng = np.random.RandomState(seed)
    return normalise_positions(rng.rand(num_channels, 2))


def fourier_tables(loc: Tensor, K: int) -> Tuple[Tensor, Tensor]:
    """models.py:20-40 — cos/sin(2π(k·x + l·y)) with m = k*K + l (k-major), shape (K², C)."""
    kk = torch.arange(K).repeat_interleave(K)       # k of pair m
    ll = torch.arange(K).repeat(K)                  # l of pair m
    x, y = loc[:, 0], loc[:, 1]
    # the reference multiplies int64 k,l by float32 x,y in an einsum (type promotion -> fp32)
    phi = 2 * torch.pi * (kk[:, None] * x[None, :] + ll[:, None] * y[None, :])
    return torch.cos(phi), torch.sin(phi)


def dropout_mask(loc: Tensor, centre_idx: Optional[int], d_drop: float) -> Optional[Tensor]:
    """models.py:81-83 — 0 where ‖loc − loc[centre]‖ < d_drop, else 1.  None ⇒ eval mode."""
    if centre_idx is None:
        return None
    dist = (loc - loc[centre_idx]).norm(dim=-1)
    return torch.where(dist < d_drop, 0.0, 1.0).to(loc.dtype)


# --------------------------------------------------------------------------------------
# encoder stages
# --------------------------------------------------------------------------------------
def sa_weights(P: Dict[str, Tensor]) -> Tensor:
    """models.py:49-58 — a = Re(z)·cos + Im(z)·sin, softmax over sensors. Returns (D1, C)."""
    z = P["subject_block.spatial_attention.z"]
    cos = P["subject_block.spatial_attention.cos"]
    sin = P["subject_block.spatial_attention.sin"]
    a = z.real @ cos + z.imag @ sin
    return torch.softmax(a, dim=-1)


def spatial_attention(P, X: Tensor, mask: Optional[Tensor]) -> Tensor:
    """models.py:45-65 (+ SpatialDropout 77-86, no 1/(1-p) rescale)."""
    W = sa_weights(P)
    if mask is not None:
        X = X * mask.to(X.dtype)[None, :, None]
    return torch.einsum("oc,bct->bot", W.to(X.dtype), X)


def subject_block(P, X: Tensor, subject_idxs, mask: Optional[Tensor], taps: Optional[dict] = None) -> Tensor:
    """models.py:111-117 — SA → shared 1×1 (bias) → per-subject 1×1 (no bias).

    The per-sample Python loop (114-116) is restated as a gathered batched matmul.
    `taps` (tests only): receives the shared conv's output with its gradient retained — the terms whose sum over
    (batch, time) is that conv's bias gradient."""
    H = spatial_attention(P, X, mask)
    H = TF.conv1d(H, P["subject_block.conv.weight"], P["subject_block.conv.bias"])
    if taps is not None:
        if H.requires_grad:
            H.retain_grad()
        taps["subject_block.conv.out"] = H
    idx = torch.as_tensor(subject_idxs).long().tolist()
    Ws = torch.stack([P[f"subject_block.subject_layer.{s}.weight"][:, :, 0] for s in idx])
    return torch.bmm(Ws, H)


def batchnorm_train(x: Tensor, w: Tensor, b: Tensor, stats: Optional[dict], prefix: str) -> Tensor:
    """nn.BatchNorm1d training mode (models.py:158,161): biased batch variance over (B, T) for the
    normalisation; running stats move by momentum 0.1 with the UNBIASED variance."""
    n = x.shape[0] * x.shape[2]
    mean = x.mean(dim=(0, 2))
    var = ((x - mean[None, :, None]) ** 2).mean(dim=(0, 2))
    if stats is not None:
        with torch.no_grad():
            rm, rv = stats[prefix + ".running_mean"], stats[prefix + ".running_var"]
            rm.mul_(1 - BN_MOMENTUM).add_(BN_MOMENTUM * mean.detach().to(rm.dtype))
            rv.mul_(1 - BN_MOMENTUM).add_(BN_MOMENTUM * (var.detach() * n / (n - 1)).to(rv.dtype))
            stats[prefix + ".num_batches_tracked"] += 1
    xhat = (x - mean[None, :, None]) / torch.sqrt(var[None, :, None] + BN_EPS)
    return xhat * w[None, :, None] + b[None, :, None]


def batchnorm_eval(x: Tensor, w, b, rm, rv) -> Tensor:
    scale = w / torch.sqrt(rv.to(x.dtype) + BN_EPS)
    return (x - rm.to(x.dtype)[None, :, None]) * scale[None, :, None] + b[None, :, None]


def block_dilations(k: int) -> Tuple[int, int, int]:
    """models.py:133,141,149 — conv0: 2^((2k)%5), conv1: 2^((2k+1)%5), conv2: always 2."""
    return 2 ** ((2 * k) % 5), 2 ** ((2 * k + 1) % 5), 2


def conv_block(P, x: Tensor, k: int, training: bool, stats: Optional[dict]) -> Tensor:
    """models.py:152-166."""
    pre = f"conv_blocks.conv{k}."
    d0, d1, d2 = block_dilations(k)

    def bn(h, j):
        name = pre + f"batchnorm{j}"
        if training:
            return batchnorm_train(h, P[name + ".weight"], P[name + ".bias"], stats, name)
        return batchnorm_eval(h, P[name + ".weight"], P[name + ".bias"],
                              P[name + ".running_mean"], P[name + ".running_var"])

    h = TF.conv1d(x, P[pre + "conv0.weight"], P[pre + "conv0.bias"], padding=d0, dilation=d0)
    if k > 0:
        h = h + x
    h = TF.gelu(bn(h, 0))
    h = TF.conv1d(h, P[pre + "conv1.weight"], P[pre + "conv1.bias"], padding=d1, dilation=d1) + h
    h = TF.gelu(bn(h, 1))
    h = TF.conv1d(h, P[pre + "conv2.weight"], P[pre + "conv2.bias"], padding=d2, dilation=d2)
    a, g = h.chunk(2, dim=1)                       # F.glu(dim=-2), models.py:164
    return a * torch.sigmoid(g)


def brain_encoder_forward(P, X: Tensor, subject_idxs, *, training: bool, loc: Optional[Tensor] = None,
                          drop_centre: Optional[int] = None, d_drop: float = 0.1,
                          stats: Optional[dict] = None, taps: Optional[dict] = None) -> Tensor:
    """models.py:191-196.  `drop_centre` replaces the NumPy global-RNG draw at models.py:81 so tests
    can inject it; in eval mode (training=False) dropout is the identity and BN uses running stats.
    `stats` (dict of running_mean/var/num_batches_tracked clones) receives the BN updates."""
    mask = dropout_mask(loc, drop_centre, d_drop) if (training and drop_centre is not None) else None
    h = subject_block(P, X, subject_idxs, mask, taps)
    for k in range(5):
        h = conv_block(P, h, k, training, stats)
    h = TF.gelu(TF.conv1d(h, P["conv_final1.weight"], P["conv_final1.bias"]))
    h = TF.gelu(TF.conv1d(h, P["conv_final2.weight"], P["conv_final2.bias"]))
    return h


# --------------------------------------------------------------------------------------
# loss (loss.py:38-84, fast path) and retrieval accuracy (models.py:208-248)
# --------------------------------------------------------------------------------------
def clip_logits(Y: Tensor, Z: Tensor, temp: Tensor) -> Tensor:
    """loss.py:60-71 — rows = first argument (speech), cols = second (brain); no eps in the norms."""
    B = Y.shape[0]
    y = Y.reshape(B, -1)
    z = Z.reshape(B, -1)
    y = y / y.norm(dim=-1, keepdim=True)
    z = z / z.norm(dim=-1, keepdim=True)
    return (y @ z.T) * torch.exp(temp)


def clip_loss(Y: Tensor, Z: Tensor, temp: Tensor, reduction: str = "mean") -> Tuple[Tensor, Tensor]:
    """loss.py:38-84 with fast=True.  Returns (loss, logits)."""
    assert Y.shape[0] > 1, "Batch size must be greater than 1."     # loss.py:40
    logits = clip_logits(Y, Z, temp)
    tgt = torch.arange(Y.shape[0])
    loss = (TF.cross_entropy(logits, tgt, reduction=reduction)
            + TF.cross_entropy(logits.t(), tgt, reduction=reduction)) / 2
    return loss, logits


def clip_loss_blockwise(Y: Tensor, Zsrc: Tensor, col_src, temp: Tensor, *, grad_blocks=(), chunk: int = 32768,
                        reduction: str = "mean") -> dict:
    """loss.py:58-79 (fast path) and its gradient for batches too large for autograd's saved copies — the loss block of the
    data-parallel configurations at full size (2048 / 4096 speech rows, contraction length up to 1 024 000).

    Y: (B, N) speech rows.  The B brain rows are given indirectly: brain row j is Zsrc[col_src[j]] (Zsrc: (M, N) with
    M <= B distinct rows; col_src: length-B index list), so a test can repeat a block of columns without the CPU paying for
    the repeated similarity products.  Same function of (Y, Z, temp) as clip_loss: only the order of the arithmetic differs —
    the two big contractions (Y Z^T and dS^T Y) run in column chunks of fp32 matmuls accumulated in fp64, and the chain
    rule through the small (B x B) tail is autograd's.  Returns loss, logits (B, B), dtemp and, for every j0:j1 in
    `grad_blocks`, dZ[(j0, j1)] = d loss / d Z[j0:j1] (B_blk, N) TREATING THE B BRAIN ROWS AS INDEPENDENT variables (what each
    data-parallel rank computes for its own columns).  Pinned to clip_loss + autograd by tests/test_oracle_golden.py."""
    B, N = Y.shape
    col_src = torch.as_tensor(col_src, dtype=torch.long)
    M = Zsrc.shape[0]
    S = torch.zeros((B, M), dtype=torch.float64)
    ysq = torch.zeros(B, dtype=torch.float64)
    zsq = torch.zeros(M, dtype=torch.float64)
    for c0 in range(0, N, chunk):
        yc, zc = Y[:, c0: c0 + chunk].float(), Zsrc[:, c0: c0 + chunk].float()
        S += (yc @ zc.T).double()
        ysq += yc.double().pow(2).sum(dim=1)
        zsq += zc.double().pow(2).sum(dim=1)
    yn = ysq.sqrt()                                                   # loss.py:64-65, no eps
    Sl = S[:, col_src].clone().requires_grad_(True)                    # (B, B) raw dot products <Y_i, Z_j>
    zn = zsq.sqrt()[col_src].clone().requires_grad_(True)
    t = temp.detach().double().clone().requires_grad_(True)
    logits = Sl / (yn[:, None] * zn[None, :]) * torch.exp(t)          # loss.py:68-71
    tgt = torch.arange(B)
    loss = (TF.cross_entropy(logits, tgt, reduction=reduction) + TF.cross_entropy(logits.t(), tgt, reduction=reduction)) / 2
    loss.backward()
    out = {"loss": loss.detach(), "logits": logits.detach(), "dtemp": t.grad.detach(), "dZ": {}}
    dS, dzn = Sl.grad, zn.grad                                         # d loss / d <Y_i, Z_j>,  d loss / d |Z_j|
    for (j0, j1) in grad_blocks:
        Zb = Zsrc[col_src[j0:j1]]
        coef = (dzn[j0:j1] / zn.detach()[j0:j1]).float()               # d|Z_j| / dZ_j = Z_j / |Z_j|
        g = torch.empty((j1 - j0, N), dtype=torch.float32)
        W = dS[:, j0:j1].t().float().contiguous()
        for c0 in range(0, N, chunk):
            g[:, c0: c0 + chunk] = W @ Y[:, c0: c0 + chunk].float() + coef[:, None] * Zb[:, c0: c0 + chunk].float()
        out["dZ"][(j0, j1)] = g
    return out


def topk_accuracy(Z: Tensor, Y: Tensor, k: int = 10) -> Tuple[float, float]:
    """models.py:208-248 restated as one matmul: sim[i, j] = cos(Y_i, Z_j) (rows = speech after the
    transpose at models.py:233); top-1 = argmax on the diagonal, top-k = diagonal within top-k."""
    B = Z.shape[0]
    z = Z.reshape(B, -1).double()
    y = Y.reshape(B, -1).double()
    denom = torch.clamp(y.norm(dim=-1)[:, None] * z.norm(dim=-1)[None, :], min=1e-8)
    sim = (y @ z.T) / denom
    diag = torch.arange(B)
    top1 = (sim.argmax(dim=1) == diag).double().mean().item()
    idx = torch.topk(sim, k, dim=1).indices
    topk = (idx == diag[:, None]).any(dim=1).double().mean().item()
    return top1, topk


def classifier_loop(Z: Tensor, Y: Tensor, k: int = 10) -> Tuple[float, float]:
    """models.py:226-243 as written (B² Python iterations) — small cases only."""
    B = Z.shape[0]
    x = Z.reshape(B, -1)
    y = Y.reshape(B, -1)
    sim = torch.empty(B, B)
    for i in range(B):
        for j in range(B):
            sim[i, j] = (x[i] @ y[j]) / max(float(x[i].norm() * y[j].norm()), 1e-8)
    sim = sim.T
    diag = torch.arange(B)
    top1 = (sim.argmax(dim=1) == diag).float().mean().item()
    rows = torch.topk(sim, k, dim=1).indices
    topk = float(np.mean([int(l) in r.tolist() for r, l in zip(rows, diag)]))
    return top1, topk


# --------------------------------------------------------------------------------------
# parameter construction helpers (shapes of SURVEY.md §8b)
# --------------------------------------------------------------------------------------
def param_shapes(C: int, S: int, D1: int, D2: int, F: int, K: int) -> Dict[str, Tuple[tuple, str]]:
    """Ordered {key: (shape, kind)} in the reference's state_dict order. kind ∈ param/buffer/complex."""
    sh: Dict[str, Tuple[tuple, str]] = {}
    sa = "subject_block.spatial_attention."
    sh[sa + "z"] = ((D1, K * K), "complex")
    sh[sa + "cos"] = ((K * K, C), "buffer")
    sh[sa + "sin"] = ((K * K, C), "buffer")
    sh["subject_block.conv.weight"] = ((D1, D1, 1), "param")
    sh["subject_block.conv.bias"] = ((D1,), "param")
    for s in range(S):
        sh[f"subject_block.subject_layer.{s}.weight"] = ((D1, D1, 1), "param")
    for k in range(5):
        cin = D1 if k == 0 else D2
        pre = f"conv_blocks.conv{k}."
        sh[pre + "conv0.weight"] = ((D2, cin, 3), "param")
        sh[pre + "conv0.bias"] = ((D2,), "param")
        for nm in ("weight", "bias", "running_mean", "running_var"):
            sh[pre + "batchnorm0." + nm] = ((D2,), "param" if nm in ("weight", "bias") else "buffer")
        sh[pre + "batchnorm0.num_batches_tracked"] = ((), "buffer")
        sh[pre + "conv1.weight"] = ((D2, D2, 3), "param")
        sh[pre + "conv1.bias"] = ((D2,), "param")
        for nm in ("weight", "bias", "running_mean", "running_var"):
            sh[pre + "batchnorm1." + nm] = ((D2,), "param" if nm in ("weight", "bias") else "buffer")
        sh[pre + "batchnorm1.num_batches_tracked"] = ((), "buffer")
        sh[pre + "conv2.weight"] = ((2 * D2, D2, 3), "param")
        sh[pre + "conv2.bias"] = ((2 * D2,), "param")
    sh["conv_final1.weight"] = ((2 * D2, D2, 1), "param")
    sh["conv_final1.bias"] = ((2 * D2,), "param")
    sh["conv_final2.weight"] = ((F, 2 * D2, 1), "param")
    sh["conv_final2.bias"] = ((F,), "param")
    return sh


def seeded_params(C: int, S: int, D1: int, D2: int, F: int, K: int, *, seed: int = 0,
                  loc: Optional[Tensor] = None) -> Dict[str, Tensor]:
    """Documented, reproducible fill used for the full-dimension golden spot checks: every entry is
    drawn in state_dict key order from ONE torch.Generator(seed):
      conv weights ~ U(-b, b), b = 1/sqrt(fan_in) (fan_in = C_in·k);  conv biases ~ U(-b, b);
      BN weight ~ U(0.5, 1.5), BN bias ~ U(-0.5, 0.5), running_mean ~ U(-0.1, 0.1),
      running_var ~ U(0.5, 1.5), num_batches_tracked = 0;  z.re, z.im ~ U[0, 1) (models.py:33);
      cos/sin from `fourier_tables(loc, K)`.
    """
    g = torch.Generator().manual_seed(seed)
    if loc is None:
        loc = synthetic_positions(C, seed)
    cos, sin = fourier_tables(loc, K)
    P: Dict[str, Tensor] = {}
    shapes = param_shapes(C, S, D1, D2, F, K)

    def U(shape, lo, hi):
        return torch.rand(shape, generator=g) * (hi - lo) + lo

    for key, (shape, kind) in shapes.items():
        if kind == "complex":
            re = U(shape, 0.0, 1.0)
            im = U(shape, 0.0, 1.0)
            P[key] = torch.complex(re, im)
        elif key.endswith(".cos"):
            P[key] = cos.clone()
        elif key.endswith(".sin"):
            P[key] = sin.clone()
        elif key.endswith("num_batches_tracked"):
            P[key] = torch.zeros((), dtype=torch.long)
        elif ".batchnorm" in key:
            if key.endswith(".weight"):
                P[key] = U(shape, 0.5, 1.5)
            elif key.endswith(".bias"):
                P[key] = U(shape, -0.5, 0.5)
            elif key.endswith("running_mean"):
                P[key] = U(shape, -0.1, 0.1)
            else:
                P[key] = U(shape, 0.5, 1.5)
        elif key.endswith(".weight"):
            bound = 1.0 / math.sqrt(shape[1] * shape[2])
            P[key] = U(shape, -bound, bound)
        else:  # conv bias: the bound uses the fan-in of the matching weight
            w = P[key[: -len("bias")] + "weight"]
            bound = 1.0 / math.sqrt(w.shape[1] * w.shape[2])
            P[key] = U(shape, -bound, bound)
    return P


def synthetic_batch(B: int, C: int, T: int, F: int, S: int, *, seed: int = 1234):
    """SURVEY.md §8(d) inputs: X ~ N(0,1) clamped ±20, Y ~ N(0,1), subject ~ U{0..S-1} (int32)."""
    g = torch.Generator().manual_seed(seed)
    X = torch.randn(B, C, T, generator=g).clamp_(-20, 20)
    Y = torch.randn(B, F, T, generator=g)
    subj = torch.randint(0, S, (B,), generator=g, dtype=torch.int32)
    return X, Y, subj


def train_step(P: Dict[str, Tensor], temp: Tensor, X, Y, subj, *, loc, drop_centre, d_drop=0.1,
               stats=None, reduction="mean", training=True, taps=None):
    """One forward+loss+backward (train.py:187-201) on leaf copies; returns (loss, Z, logits, grads).
    training=False: the same through an encoder in .eval() mode (BatchNorm on running statistics, no dropout).
    taps (tests only): dict that receives intermediate tensors with retained gradients (see subject_block)."""
    leaves = {}
    for k, v in P.items():
        if v.is_floating_point() or v.is_complex():
            if not (k.endswith("running_mean") or k.endswith("running_var")
                    or k.endswith(".cos") or k.endswith(".sin")):
                leaves[k] = v.detach().clone().requires_grad_(True)
    Q = dict(P)
    Q.update(leaves)
    t = temp.detach().clone().requires_grad_(True)
    Z = brain_encoder_forward(Q, X, subj, training=training, loc=loc, drop_centre=drop_centre,
                              d_drop=d_drop, stats=stats, taps=taps)
    loss, logits = clip_loss(Y, Z, t, reduction)
    loss.backward()
    grads = {k: v.grad for k, v in leaves.items()}
    grads["temp"] = t.grad
    return loss.detach(), Z.detach(), logits.detach(), grads


# --------------------------------------------------------------------------------------
# batch collate: gwilliams2022.py:651-661 = preproc_utils.py:128-142 + 69-90
# --------------------------------------------------------------------------------------
def collate_batch(X: Tensor, baseline_len_samp: int, clamp_lim: float, clamp: bool = True) -> Tensor:
    """Baseline-correct every (sample, channel) row by the mean of its first `baseline_len_samp` samples, then
    RobustScaler over time per row (median, inter-quartile range with linear interpolation, zero range -> 1;
    sklearn.preprocessing.RobustScaler defaults as called at preproc_utils.py:82), then clamp."""
    X = X.to(torch.float32)
    X = X - X[:, :, :baseline_len_samp].mean(dim=-1, keepdim=True)
    x = X.numpy()
    q25, med, q75 = np.percentile(x, [25.0, 50.0, 75.0], axis=-1)
    scale = q75 - q25
    scale[scale == 0.0] = 1.0
    out = torch.from_numpy(((x - med[..., None]) / scale[..., None]).astype(np.float32))
    if clamp:
        out = out.clamp(-clamp_lim, clamp_lim)
    return out
