"""Data-parallel helpers: one process per GPU, torch.distributed over RCCL (backend "nccl" on ROCm).

What shards and what is exchanged:
  * samples are sharded across ranks, weights replicated;
  * speech embeddings Y are all-gathered inside CLIPLoss so the negatives span the global batch;
  * BatchNorm partial statistics are all-reduced inside the encoder (sync_batchnorm);
  * parameter gradients are SUM-all-reduced in flat buckets (the loss shares already carry the
    1/(2*B_global) normalisation, so the sum over ranks is the global-batch gradient): the encoder does it
    per layer group INSIDE backward, asynchronously on RCCL's stream (engine.overlap_grad_allreduce);
    `allreduce_gradients` below serves whatever is left (CLIPLoss.temp) or everything when that is off.
"""
from __future__ import annotations

from typing import Iterable, List

import torch
import torch.distributed as dist


def _flat_views(params: Iterable[torch.nn.Parameter]) -> List[torch.Tensor]:
    out = []
    for p in params:
        if p.grad is None:
            p.grad = torch.zeros_like(p)          # e.g. subjects absent from this rank's shard
        g = p.grad
        out.append(torch.view_as_real(g).reshape(-1) if g.is_complex() else g.reshape(-1))
    return out


def allreduce_gradients(params: Iterable[torch.nn.Parameter], group=None, bucket_bytes: int = 64 << 20):
    """SUM all-reduce of every parameter gradient in flat fp32 buckets (few large collectives: xGMI rings
    are per-link bound, so fewer, larger messages win)."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return
    views = _flat_views(list(params))
    bucket, size = [], 0
    def flush():
        nonlocal bucket, size
        if not bucket:
            return
        flat = torch.cat(bucket)
        dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
        off = 0
        for v in bucket:
            v.copy_(flat[off: off + v.numel()])
            off += v.numel()
        bucket, size = [], 0
    for v in views:
        bucket.append(v)
        size += v.numel() * 4
        if size >= bucket_bytes:
            flush()
    flush()


def broadcast_parameters(module: torch.nn.Module, src: int = 0, group=None):
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return
    for t in list(module.parameters()) + list(module.buffers()):
        data = torch.view_as_real(t.data) if t.is_complex() else t.data
        dist.broadcast(data, src=src, group=group)


def shard_range(n: int, rank: int, world: int):
    """Contiguous shard [lo, hi) of n samples for `rank` (global sample order = rank-major)."""
    per = n // world
    return rank * per, (rank + 1) * per


def merge_row_softmax_stats(row_max: torch.Tensor, row_sum: torch.Tensor, group=None):
    """Each rank holds, for every GLOBAL speech row, (max, sum exp(l - max)) over ITS OWN block of brain
    columns.  Returns the row-wise log-sum-exp over all columns of all ranks (2 small all-reduces):
        M = max_r m_r ;  S = sum_r s_r * exp(m_r - M) ;  lse = M + log S."""
    if group is not None and dist.is_initialized() and dist.get_world_size(group) > 1:     # None = stay local
        gmax = row_max.clone()
        dist.all_reduce(gmax, op=dist.ReduceOp.MAX, group=group)
        row_sum = row_sum * torch.exp(row_max - gmax)
        dist.all_reduce(row_sum, op=dist.ReduceOp.SUM, group=group)
        row_max = gmax
    return row_max + torch.log(row_sum)
