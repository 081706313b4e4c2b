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

import os
from typing import Iterable, List

import torch
import torch.distributed as dist


_side_groups = {}


def active_group():
    """The process group the data-parallel paths use: WORLD when torch.distributed is initialised with more than one
    rank, else None (single-process code path, no collectives).  SDA_DP_SINGLE_RANK=1 keeps the data-parallel path
    on even at world size 1 — every collective then runs through the backend with one rank (the RCCL call sequence
    can be exercised on a one-GPU box; results equal the single-process path)."""
    if not (dist.is_available() and dist.is_initialized()):
        return None
    if dist.get_world_size() > 1 or os.environ.get("SDA_DP_SINGLE_RANK") == "1":
        return dist.group.WORLD
    return None


def init_process_group(backend: str = "nccl", **kw):
    """dist.init_process_group with RCCL's stream of the DEFAULT group at high priority: that group carries the step's
    latency-bound collectives (the 2 * Cp-float BatchNorm all-reduces, the row-statistics gather — each one a wait on the
    high-priority training stream's chain), which would otherwise queue behind the normal-priority weight-gradient GEMMs for
    CUs.  The bulk communicators (side_group: "gather", "grads") keep the default priority.  SDA_RCCL_HIGH_PRIORITY=0 turns
    it off.  By design, not by measurement: at world size 1 RCCL launches no kernel, and no multi-GPU box was available."""
    if backend == "nccl" and os.environ.get("SDA_RCCL_HIGH_PRIORITY", "1") != "0" and "pg_options" not in kw:
        try:
            opts = dist.ProcessGroupNCCL.Options()
            opts.is_high_priority_stream = True
            kw["pg_options"] = opts
        except (AttributeError, RuntimeError):
            pass
    return dist.init_process_group(backend, **kw)


_EMULATED_WORLD = 1


def emulate_world(n: int):
    """Measurement aid (bench.py --emulate-world N): with ONE rank (SDA_DP_SINGLE_RANK=1) the loss behaves as rank 0 of an
    N-rank job — its speech-row buffer holds N * B_local samples of which (N - 1) * B_local are resident stand-ins for the rows
    the all-gather would have delivered, the logits block is (N * B_local) x B_local, the loss is normalised by the global
    batch — while every collective of the step is issued for real at world size 1.  What it measures is the COMPUTE side of a
    rank's step at the N-rank shape; the wire is not in it."""
    global _EMULATED_WORLD
    if n < 1:
        raise ValueError("emulate_world: n >= 1")
    _EMULATED_WORLD = int(n)


def emulated_world() -> int:
    return _EMULATED_WORLD


def side_group(name: str, group=None):
    """A second communicator over the same ranks as `group` (default: WORLD), created once per name.

    RCCL runs the collectives of ONE communicator in issue order on one stream, so a 1.3 GB speech-row all-gather
    or a gradient-bucket all-reduce issued on the default group would sit IN FRONT of the next latency-critical
    BatchNorm-statistics all-reduce.  The bulk collectives therefore get communicators of their own ("gather",
    "grads").  Every rank must reach the first call for a name at the same point of the program (it is a
    collective): the callers create it on their first collective of that kind, which all ranks issue together."""
    if not (dist.is_available() and dist.is_initialized()) or (dist.get_world_size(group) == 1
                                                               and os.environ.get("SDA_DP_SINGLE_RANK") != "1"):
        return group
    key = (name, id(group) if group is not None else 0)
    g = _side_groups.get(key)
    if g is None:
        ranks = dist.get_process_group_ranks(group if group is not None else dist.group.WORLD)
        g = _side_groups[key] = dist.new_group(ranks=ranks)
    return g


def shutdown():
    """Orderly end of a data-parallel run: everything queued on this rank's device is finished, every rank has arrived, the
    side communicators ("gather", "grads") go first, the default group last.  (Call CLIPLoss.drain() before it when a
    prefetch may be pending.)  Without the order a rank that leaves early takes its transport threads down under a peer's
    collective — an abort at exit that looks like a failed run."""
    if not (dist.is_available() and dist.is_initialized()):
        return
    if torch.cuda.is_available():
        torch.cuda.synchronize()
    dist.barrier()
    for g in list(_side_groups.values()):
        try:
            dist.destroy_process_group(g)
        except (ValueError, RuntimeError, AssertionError):
            pass                     # already gone (or the default group itself at world size 1)
    _side_groups.clear()
    dist.barrier()
    dist.destroy_process_group()


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
    are per-link bound, so fewer, larger messages win).  The gradients are MOVED into the bucket: after the call each
    `.grad` is a view of its bucket (one pack, no copy back); a single gradient is reduced in place."""
    if active_group() is None and group is None:
        return
    plist = [p for p in params]
    _flat_views(plist)                                    # materialises missing gradients (zeros)
    bucket: List[torch.nn.Parameter] = []
    size = 0

    def flush():
        nonlocal bucket, size
        if not bucket:
            return
        if len(bucket) == 1 and bucket[0].grad.is_contiguous():
            g = bucket[0].grad
            dist.all_reduce(torch.view_as_real(g) if g.is_complex() else g, op=dist.ReduceOp.SUM, group=group)
        else:
            # (every entry starts at an even float offset: a complex gradient is viewed as complex inside the bucket)
            parts, offs, off = [], [], 0
            for p in bucket:
                v = (torch.view_as_real(p.grad) if p.grad.is_complex() else p.grad).reshape(-1)
                if off % 2:
                    parts.append(v.new_zeros(1))
                    off += 1
                offs.append(off)
                parts.append(v)
                off += v.numel()
            flat = torch.cat(parts)
            dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
            for p, o in zip(bucket, offs):
                g = p.grad
                v = flat[o: o + g.numel() * (2 if g.is_complex() else 1)]
                p.grad = torch.view_as_complex(v.view(*g.shape, 2)) if g.is_complex() else v.view(g.shape)
        bucket, size = [], 0
    for p in plist:
        bucket.append(p)
        size += p.grad.numel() * (8 if p.grad.is_complex() else 4)
        if size >= bucket_bytes:
            flush()
    flush()


def seed_numpy_all_ranks(seed: int = None, group=None) -> int:
    """Put NumPy's GLOBAL generator in the same state on every rank: SpatialDropout draws its centre from it once per
    training forward (models.py:81) and the batch — hence the centre — is global.  seed=None: rank 0 draws one and
    broadcasts it.  Returns the seed.  (A no-op without torch.distributed; callers that draw from np.random elsewhere must
    do so on all ranks alike, or use BrainEncoder.drop_centre_sync = "broadcast".)"""
    import numpy as np
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        if seed is not None:
            np.random.seed(int(seed))
        return -1 if seed is None else int(seed)
    box = [int(seed) if seed is not None else int(np.random.randint(0, 2 ** 31 - 1))]
    dist.broadcast_object_list(box, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
    np.random.seed(box[0])
    return box[0]


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


def merge_row_softmax_stats(row_max: torch.Tensor, row_sum: torch.Tensor, group=None, diag: torch.Tensor = None):
    """Each rank holds, for every GLOBAL speech row, (max, sum exp(l - max)) over ITS OWN block of brain
    columns.  Returns the row-wise log-sum-exp over all columns of all ranks:
        M = max_r m_r ;  S = sum_r s_r * exp(m_r - M) ;  lse = M + log S.
    ONE collective: the (max, sum[, diag]) rows of every rank are all-gathered (3 * B_global floats per rank)
    and merged locally in rank order — the small collectives of a step are latency-bound, so they are packed.
    With `diag` (each rank's positives' logits, zero where the positive lives elsewhere) returns
    (lse, diag summed over ranks)."""
    if group is not None and dist.is_initialized():     # None = stay local
        world = dist.get_world_size(group)
        parts = [row_max, row_sum] + ([diag] if diag is not None else [])
        n = row_max.numel()
        if (len(parts) == 3 and all(p.dtype == torch.float32 and p.is_contiguous() for p in parts)
                and row_sum.data_ptr() == row_max.data_ptr() + 4 * n and diag.data_ptr() == row_max.data_ptr() + 8 * n):
            mine = row_max.as_strided((3, n), (n, 1))                 # the three are rows of one buffer (ops.clip_logits_stats)
        else:
            mine = torch.stack([p.to(torch.float32) for p in parts]).contiguous()         # (k, Bg)
        flat = torch.empty((world * mine.shape[0], mine.shape[1]), dtype=torch.float32, device=mine.device)
        dist.all_gather_into_tensor(flat, mine, group=group)          # concatenation along dim 0, rank-major
        table = flat.view(world, mine.shape[0], mine.shape[1])
        if table.is_cuda and table.shape[1] == 3:                     # one kernel instead of ~ten framework launches
            from . import ops
            return ops.clip_merge_rows(table)
        return combine_row_stats(table)
    lse = row_max + torch.log(row_sum)
    return lse if diag is None else (lse, diag)


def combine_row_stats(allp: torch.Tensor):
    """The merge itself, on the gathered (world, 2 or 3, B_global) table of per-rank (max, sum exp(l - max)[, diag]) rows, in
    rank order: lse, or (lse, diag) when the table carries the positives' logits."""
    gmax = allp[:, 0].max(dim=0).values
    row_sum = (allp[:, 1] * torch.exp(allp[:, 0] - gmax)).sum(dim=0)
    lse = gmax + torch.log(row_sum)
    return lse if allp.shape[1] < 3 else (lse, allp[:, 2].sum(dim=0))


class CollectiveTimer:
    """Diagnostics for the scaling runs: HIP-event brackets around every torch.distributed collective of a few steps, on the
    stream that issues it (and, for async ones, around the wait) — what each collective costs the stream that has to wait
    for it, grouped by (operation, payload size).  install() wraps dist.all_reduce / all_gather_into_tensor / broadcast;
    uninstall() restores them; summary() -> {label: (calls, total_us)}.  Never installed inside a timed region."""

    OPS = ("all_reduce", "all_gather_into_tensor", "broadcast")

    def __init__(self):
        self.records = []
        self._saved = {}

    def _wrap(self, name, fn):
        timer = self

        def wrapped(*args, **kw):
            t = args[1] if name == "all_gather_into_tensor" else args[0]
            label = f"{name} {t.numel() * t.element_size()} B"
            if not t.is_cuda:
                return fn(*args, **kw)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            work = fn(*args, **kw)
            if work is not None and kw.get("async_op"):
                wait = work.wait

                def timed_wait(*a, **k):
                    w0, w1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    w0.record()
                    r = wait(*a, **k)
                    w1.record()
                    timer.records.append((label + " (wait of async)", w0, w1))
                    return r
                try:
                    work.wait = timed_wait
                except AttributeError:
                    pass
                e1.record()
                timer.records.append((label + " (async issue)", e0, e1))
                return work
            e1.record()
            timer.records.append((label, e0, e1))
            return work
        return wrapped

    def install(self):
        for name in self.OPS:
            self._saved[name] = getattr(dist, name)
            setattr(dist, name, self._wrap(name, self._saved[name]))
        return self

    def uninstall(self):
        for name, fn in self._saved.items():
            setattr(dist, name, fn)
        self._saved = {}

    def summary(self):
        torch.cuda.synchronize()
        out = {}
        for label, e0, e1 in self.records:
            n, us = out.get(label, (0, 0.0))
            out[label] = (n + 1, us + e0.elapsed_time(e1) * 1e3)
        return out
