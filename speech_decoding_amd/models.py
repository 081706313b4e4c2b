"""MI355X-native `BrainEncoder` / `Classifier` with the reference's constructor and forward signatures
(speech_decoding/models.py:169-248) and state_dict keys.

The modules below only HOLD parameters (so `.parameters()`, `.to()`, `.state_dict()`, Adam all work as
with the reference); the forward and backward computations run in libsdamd.so through
`engine.EncoderEngine` as ONE autograd node.  There is no PyTorch fallback for the math.
"""
from __future__ import annotations

import os
from typing import Dict, List, Optional

import numpy as np
import torch
import torch.nn as nn

from . import lib as L
from . import ops
from .engine import EncoderDims, EncoderEngine, block_dilations
from .layout import ch_locations_2d
from . import loss as _loss


def _opt(args, key, default=None):
    if isinstance(args, dict):
        return args.get(key, default)
    return getattr(args, key, default)


def resolve_dtype(args) -> torch.dtype:
    name = os.environ.get("SDA_COMPUTE_DTYPE") or _opt(args, "compute_dtype", "fp32") or "fp32"
    name = str(name).lower()
    if name in ("fp32", "float32", "f32"):
        return torch.float32
    if name in ("bf16", "bfloat16"):
        return torch.bfloat16
    if name in ("fp16", "float16", "half", "f16"):
        return torch.float16
    raise ValueError(f"compute_dtype must be fp32, bf16 or fp16, got {name!r}")


def _dp_group():
    from .distributed import active_group
    return active_group()


class SpatialAttention(nn.Module):
    """Parameter holder for models.py:14-65: complex `z` (D1, K²) ~ U[0,1)+iU[0,1), buffers cos/sin (K², C)."""

    def __init__(self, args):
        super().__init__()
        K = int(args.K)
        loc = ch_locations_2d(args)                                    # (C, 2) in [0.1, 0.9]
        self.z = nn.Parameter(torch.rand(size=(int(args.D1), K * K), dtype=torch.cfloat))
        kk = torch.arange(K).repeat_interleave(K)                       # m = k*K + l
        ll = torch.arange(K).repeat(K)
        phi = 2 * torch.pi * (kk[:, None] * loc[None, :, 0] + ll[:, None] * loc[None, :, 1])
        self.register_buffer("cos", torch.cos(phi))
        self.register_buffer("sin", torch.sin(phi))
        self.loc = loc                                                  # plain attribute, as in the reference
        self.d_drop = float(args.d_drop)
        self._tables_T = None

    def _tables_key(self):
        # rebuilt when the buffers move or are written in place (load_state_dict copies into them)
        return (str(self.cos.device), self.cos.data_ptr(), self.cos._version, self.sin._version)

    def transposed_tables(self):
        key = self._tables_key()
        if self._tables_T is None or self._tables_T[0] != key:
            self._tables_T = (key, self.cos.t().contiguous(), self.sin.t().contiguous())
        return self._tables_T[1], self._tables_T[2]

    def gemm_tables(self):
        """Operand tables of the two contractions of the weight build for the matrix-core path (ops._sa_gemm_tables)."""
        from . import ops
        key = self._tables_key()
        if getattr(self, "_tables_G", None) is None or self._tables_G[0] != key:
            self._tables_G = (key,) + tuple(ops.sa_gemm_tables(self.cos, self.sin))
        return self._tables_G[1], self._tables_G[2]

    def draw_mask(self) -> torch.Tensor:
        """models.py:81-83: one centre per forward from NumPy's global RNG; 0 within d_drop of it."""
        centre = int(np.random.randint(self.loc.shape[0]))
        return self.mask_for(centre)

    def device_masks(self, device) -> torch.Tensor:
        """All C possible dropout masks (row = centre sensor), built once and kept on the device."""
        m = getattr(self, "_masks", None)
        if m is None or m.device != torch.device(device):
            dist = (self.loc[:, None, :] - self.loc[None, :, :]).norm(dim=-1)
            m = self._masks = torch.where(dist < self.d_drop, 0.0, 1.0).to(torch.float32).to(device)
        return m

    def mask_for(self, centre: int) -> torch.Tensor:
        dist = (self.loc - self.loc[centre]).norm(dim=-1)
        return torch.where(dist < self.d_drop, 0.0, 1.0).to(torch.float32)


class SubjectLayers(nn.Module):
    """The reference's ModuleList of S bias-free 1x1 convs (models.py:98-109) stored as ONE (S, D1, D1, 1)
    parameter so a kernel can select the matrix by subject index; state_dict keys stay `{s}.weight`."""

    def __init__(self, num_subjects: int, D1: int):
        super().__init__()
        ws = [nn.Conv1d(D1, D1, kernel_size=1, bias=False).weight.detach() for _ in range(num_subjects)]
        self.weight = nn.Parameter(torch.stack(ws))                     # same RNG draws as S Conv1d inits
        self.num_subjects = num_subjects

    def __len__(self):
        return self.num_subjects

    def _save_to_state_dict(self, destination, prefix, keep_vars):
        for s in range(self.num_subjects):
            w = self.weight[s]
            destination[f"{prefix}{s}.weight"] = w if keep_vars else w.detach()

    def _load_from_state_dict(self, state_dict, prefix, local_metadata, strict, missing_keys, unexpected_keys, error_msgs):
        for s in range(self.num_subjects):
            key = f"{prefix}{s}.weight"
            if key in state_dict:
                with torch.no_grad():
                    self.weight[s].copy_(state_dict[key])
            elif strict:
                missing_keys.append(key)
        if strict:
            for key in state_dict:
                if key.startswith(prefix) and key[len(prefix):].split(".")[0].isdigit():
                    if int(key[len(prefix):].split(".")[0]) >= self.num_subjects:
                        unexpected_keys.append(key)


class SubjectBlock(nn.Module):
    def __init__(self, args):
        super().__init__()
        D1 = int(args.D1)
        self.spatial_attention = SpatialAttention(args)
        self.conv = nn.Conv1d(D1, D1, kernel_size=1, stride=1)
        self.subject_layer = SubjectLayers(int(args.num_subjects), D1)


class ConvBlock(nn.Module):
    """Parameter holder for models.py:120-150."""

    def __init__(self, k: int, D1: int, D2: int):
        super().__init__()
        cin = D1 if k == 0 else D2
        d0, d1, d2 = block_dilations(k)
        self.conv0 = nn.Conv1d(cin, D2, kernel_size=3, padding="same", dilation=d0)
        self.batchnorm0 = nn.BatchNorm1d(D2)
        self.conv1 = nn.Conv1d(D2, D2, kernel_size=3, padding="same", dilation=d1)
        self.batchnorm1 = nn.BatchNorm1d(D2)
        self.conv2 = nn.Conv1d(D2, 2 * D2, kernel_size=3, padding="same", dilation=d2)


class _EncoderFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, module: "BrainEncoder", X, subject_idxs, mask, need_grad, *params):
        P = module._param_dict(params)
        ectx = module.engine.forward(P, X, subject_idxs, training=module.training, mask=mask, need_grad=need_grad)
        ctx.module, ctx.ectx, ctx.P = module, ectx, P
        ctx.nparams = len(params)
        B, _, T = X.shape
        return ops.rows_view(ectx.bufs["Z"], B, module.F, T)

    @staticmethod
    def backward(ctx, dZ):
        module, ectx = ctx.module, ctx.ectx
        P = ctx.P
        dZt = _loss.as_rows(dZ, ectx.B, module.F, ectx.T, module.engine.dtype, "dZ")
        g = module.engine.backward(P, ectx, dZt)
        out = [g.get(name) for name in module._param_names]
        return (None, None, None, None, None, *out)


class BrainEncoder(nn.Module):
    """Drop-in for models.py:169-196: `BrainEncoder(args)(X, subject_idxs) -> (B, F, T)`.

    The returned tensor is a zero-copy (B, F, T) *view* of the channels-last buffer the kernels wrote
    (strides (Tp*Fp, 1, Fp)); `CLIPLoss` / `Classifier` from this package consume it without a copy."""

    def __init__(self, args):
        super().__init__()
        self.num_subjects = int(args.num_subjects)
        self.D1, self.D2, self.K = int(args.D1), int(args.D2), int(args.K)
        self.F = int(args.F) if not args.preprocs["last4layers"] else 1024          # models.py:176
        self.dataset_name = args.dataset
        self.compute_dtype = resolve_dtype(args)

        self.subject_block = SubjectBlock(args)
        self.conv_blocks = nn.Sequential()
        for k in range(5):
            self.conv_blocks.add_module(f"conv{k}", ConvBlock(k, self.D1, self.D2))
        self.conv_final1 = nn.Conv1d(self.D2, 2 * self.D2, kernel_size=1)
        self.conv_final2 = nn.Conv1d(2 * self.D2, self.F, kernel_size=1)

        self.sync_batchnorm = True          # under torch.distributed: BN statistics over the global batch
        self._engine: Optional[EncoderEngine] = None
        self._fixed_centre: Optional[int] = None
        self.drop_centre_sync = "seed"        # under data parallelism: "seed" (identical np.random on all ranks, checked) | "broadcast"
        self._centre_draws = 0
        self._param_names = self._build_names()

    # ---- parameter plumbing -------------------------------------------------------------
    def _build_names(self) -> List[str]:
        names = ["z", "sb_w", "sb_b", "subj_w"]
        for k in range(5):
            names += [f"b{k}.c0w", f"b{k}.c0b", f"b{k}.bn0w", f"b{k}.bn0b", f"b{k}.c1w", f"b{k}.c1b",
                      f"b{k}.bn1w", f"b{k}.bn1b", f"b{k}.c2w", f"b{k}.c2b"]
        return names + ["f1w", "f1b", "f2w", "f2b"]

    def _ordered_params(self) -> List[torch.Tensor]:
        sb = self.subject_block
        ps = [sb.spatial_attention.z, sb.conv.weight, sb.conv.bias, sb.subject_layer.weight]
        for k in range(5):
            blk = getattr(self.conv_blocks, f"conv{k}")
            ps += [blk.conv0.weight, blk.conv0.bias, blk.batchnorm0.weight, blk.batchnorm0.bias,
                   blk.conv1.weight, blk.conv1.bias, blk.batchnorm1.weight, blk.batchnorm1.bias,
                   blk.conv2.weight, blk.conv2.bias]
        return ps + [self.conv_final1.weight, self.conv_final1.bias, self.conv_final2.weight, self.conv_final2.bias]

    def _param_dict(self, params) -> Dict[str, torch.Tensor]:
        P = {n: p.detach() for n, p in zip(self._param_names, params)}
        sa = self.subject_block.spatial_attention
        P["cos"], P["sin"] = sa.cos, sa.sin
        P["cosT"], P["sinT"] = sa.transposed_tables()
        P["sa_tab_f"], P["sa_tab_b"] = sa.gemm_tables() if sa.cos.is_cuda else (None, None)
        for k in range(5):
            blk = getattr(self.conv_blocks, f"conv{k}")
            for j, bn in ((0, blk.batchnorm0), (1, blk.batchnorm1)):
                P[f"b{k}.bn{j}rm"], P[f"b{k}.bn{j}rv"] = bn.running_mean, bn.running_var
                P[f"b{k}.bn{j}nbt"] = bn.num_batches_tracked        # counted by the statistics kernel itself (training mode)
        return P

    @property
    def engine(self) -> EncoderEngine:
        if self._engine is None or self._engine.dtype != self.compute_dtype:
            C = self.subject_block.spatial_attention.cos.shape[1]
            self._engine = EncoderEngine(EncoderDims(C, self.num_subjects, self.D1, self.D2, self.F, self.K),
                                         self.compute_dtype, group=_dp_group() if self.sync_batchnorm else None)
        return self._engine

    @property
    def grads_are_reduced(self) -> bool:
        """True when backward already SUM-all-reduced this module's gradients across ranks (data parallel,
        overlapped with backward); callers then all-reduce only what lives outside the encoder (CLIPLoss.temp)."""
        e = self.engine
        return e.group is not None and e.overlap_grad_allreduce

    def set_compute_dtype(self, dtype: torch.dtype):
        self.compute_dtype = dtype
        return self

    def set_drop_centre(self, centre: Optional[int]):
        """Testing hook: pin SpatialDropout's centre instead of drawing it from np.random (models.py:81)."""
        self._fixed_centre = centre

    # ---- forward ------------------------------------------------------------------------
    def forward(self, X: torch.Tensor, subject_idxs) -> torch.Tensor:
        sa = self.subject_block.spatial_attention
        assert X.shape[1] == sa.loc.shape[0]                                      # models.py:78
        if not X.is_cuda:
            raise L.SdaError("BrainEncoder needs X on the MI355X device (there is no CPU path)")
        mask = None
        if self.training:
            if self._fixed_centre is not None:
                centre = self._fixed_centre
            else:
                centre = int(np.random.randint(sa.loc.shape[0]))            # models.py:81, NumPy global RNG
            mask = sa.device_masks(X.device)[centre]            # (C, C) table resident on the device: no upload
            group = _dp_group()
            if group is not None and self._fixed_centre is None:
                # "same drop centre for all samples in batch" (models.py:69) — the batch is global, so every rank must use
                # the same centre.  NumPy's global generator is seeded identically on all ranks
                # (distributed.seed_numpy_all_ranks; train.py / bench.py do it) and nothing else draws from it, so every
                # rank draws the same centre by itself: no collective in front of the forward.  That lockstep is CHECKED —
                # on the first data-parallel forward and every 256th after it (one small all-gather + a host read-back) —
                # and a divergence is an error, not a silent semantic change.  drop_centre_sync = "broadcast" restores the
                # per-step broadcast of rank 0's mask (a latency-bound collective on the critical path) for callers that
                # draw from np.random elsewhere.
                import torch.distributed as dist
                if self.drop_centre_sync == "broadcast":
                    mask = mask.clone()         # the cached table must not be overwritten by the broadcast
                    dist.broadcast(mask, src=0, group=group)
                else:
                    if self._centre_draws % 256 == 0:
                        mine = torch.tensor([centre], dtype=torch.int32, device=X.device)
                        allc = torch.empty(dist.get_world_size(group), dtype=torch.int32, device=X.device)
                        dist.all_gather_into_tensor(allc, mine, group=group)
                        if len(set(allc.tolist())) != 1:
                            raise L.SdaError(f"SpatialDropout centres differ across ranks ({allc.tolist()}): NumPy's global generator "
                                             "is out of step — call speech_decoding_amd.distributed.seed_numpy_all_ranks() after "
                                             "init_process_group and draw from np.random on all ranks alike, or set "
                                             "brain_encoder.drop_centre_sync = 'broadcast'")
                    self._centre_draws += 1
        params = self._ordered_params()
        need_grad = torch.is_grad_enabled() and any(p.requires_grad for p in params)   # (grad mode is off inside Function.forward)
        return _EncoderFn.apply(self, X, subject_idxs, mask, need_grad, *params)


class Classifier(nn.Module):
    """Drop-in for models.py:199-248: top-1 / top-10 retrieval accuracy of speech rows against brain
    columns.  The B² Python loop is one similarity GEMM + a rank kernel; results equal the reference's
    ranking because logits are a positive rescaling of the cosine similarities."""

    def __init__(self, args=None):
        super().__init__()
        self.factor = 1
        self.global_candidates = True       # under torch.distributed: rank against the global batch

    @torch.no_grad()
    def ranks(self, Z: torch.Tensor, Y: torch.Tensor) -> torch.Tensor:
        """Rank of every speech row's own brain column (0 = top-1) as a DEVICE tensor: what forward() reduces to the two
        accuracies, without the host read-back (a training loop can collect these and read them once per epoch)."""
        if Z.size(0) < 10:
            raise RuntimeError("selected index k out of range")      # torch.topk(…, 10) on fewer than 10 columns
        return _loss.retrieval_ranks(Y, Z, self.global_candidates)

    @torch.no_grad()
    def forward(self, Z: torch.Tensor, Y: torch.Tensor, test: bool = False):
        cnt = self.ranks(Z, Y).cpu().numpy()
        return float((cnt == 0).mean()), np.mean(cnt < 10)
