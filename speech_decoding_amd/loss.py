"""MI355X-native `CLIPLoss` with the reference's signature (speech_decoding/utils/loss.py:28-84).

forward(x, y, fast=True, return_logits=False): x = speech embeddings Y (B, F, T), y = brain embeddings
Z (B, F, T); loss = (CE(logits) + CE(logitsᵀ)) / 2 with logits = x̂ ŷᵀ · exp(temp).  The norms, the
similarity GEMM, the softmax statistics, the gradient coefficient matrix and the embedding gradient
all run in libsdamd.so.  Under torch.distributed (one process per GPU) the speech rows are all-gathered
so the negatives span the GLOBAL batch; each rank owns the columns of its local brain embeddings.
"""
from __future__ import annotations

from typing import Optional

import torch
import torch.nn as nn

from . import lib as L
from . import ops
from . import engine as E

import weakref

_rank_cache = []          # [(weakref(Y), weakref(Z), Y._version, Z._version, ranks)] from the last CLIPLoss forward


class LossState:
    """Buffers that outlive one call, owned by ONE CLIPLoss instance (nothing is shared between instances by shape):
    the rotating packed-speech buffers and the pending speech-side prefetch.  Every hand-out of a ring slot bumps
    that slot's generation; a backward checks the generation it was given, so a buffer recycled by later forwards
    raises instead of silently feeding another batch's rows into the gradient."""

    def __init__(self):
        self.rings = {}           # (key, B, C, T, dtype, device) -> [buffers, next index, generations]
        self.prefetched = []      # at most one pending prefetch
        self.ring_depth = 2
        self.emulated = {}        # distributed.emulate_world: the resident stand-ins for the other ranks' speech rows

    def drain(self):
        """Wait for a pending prefetch's collectives (the speech-row all-gather issued one batch ahead and never consumed:
        the last one of a run) and drop it.  A process group torn down while such a collective is still in flight aborts
        the process ("terminate called without an active exception": a transport thread destroyed while joinable)."""
        for item in self.prefetched:
            works, done = item[7], item[8]
            if done is not None:
                done.synchronize()
            for work in works:
                work.wait()
        if self.prefetched and torch.cuda.is_available():
            torch.cuda.synchronize()
        self.prefetched.clear()

    def clear(self):
        self.drain()
        self.rings.clear()
        self.emulated.clear()


_DEFAULT_STATE = LossState()      # for the module-level helpers called without a CLIPLoss instance


class RingSlot:
    """A ring buffer handed out at generation `gen`; valid() is False once the slot has been handed out again."""
    __slots__ = ("ring", "idx", "gen", "buf")

    def __init__(self, ring, idx, gen, buf):
        self.ring, self.idx, self.gen, self.buf = ring, idx, gen, buf

    def valid(self) -> bool:
        return self.ring[2][self.idx] == self.gen


def _cache_ranks(Y, Z, cnt):
    _rank_cache.clear()
    _rank_cache.append((weakref.ref(Y), weakref.ref(Z), Y._version, Z._version, cnt))


def _cached_ranks(Y, Z):
    """Ranks of the last CLIPLoss forward — valid for the SAME two tensor objects while neither has been written in place
    since (torch's version counters, as ops.ROW_NORMS checks them): an edit of Z between loss_func(Y, Z) and
    classifier(Z, Y) sends the classifier back to computing its own ranks."""
    for wy, wz, vy, vz, cnt in _rank_cache:
        if wy() is Y and wz() is Z and Y._version == vy and Z._version == vz:
            return cnt
    return None


def _rows_base(t: torch.Tensor, B: int, Cc: int, T: int, dtype):
    """If `t` is a (B, C, T) view laid out exactly like ops.rows_view(...) of an RL buffer, return that
    buffer as a (rows_alloc, Cp) tensor sharing its storage (zero copy); else None."""
    if t.dtype != dtype or not t.is_cuda or t.dim() != 3:
        return None
    Cp, rows = L.pad_channels(Cc), L.rows_alloc(B, T)
    if tuple(t.stride()) != (L.rows_tp(T) * Cp, 1, Cp) or t.storage_offset() != L.ROW_PAD * Cp:
        return None
    if t.untyped_storage().nbytes() < rows * Cp * t.element_size():
        return None
    return t.detach().as_strided((rows, Cp), (Cp, 1), 0)


def _ring_rows(state: LossState, key, B, Cc, T, dtype, device) -> RingSlot:
    """Persistent RL buffers for packed inputs (no per-call 200 MB allocation + memset).  `ring_depth` buffers
    rotate so that the buffer a pending backward still needs survives one further forward (e.g. an eval
    pass between forward and backward); a backward that comes later than that finds its slot's generation changed."""
    k = (key, B, Cc, T, dtype, str(device))
    ring = state.rings.get(k)
    if ring is None:
        if len(state.rings) >= 8:                  # ragged last batches etc.: keep the most recent shapes only
            state.rings.pop(next(iter(state.rings)))
        depth = state.ring_depth
        ring = state.rings[k] = [[ops.new_rows(B, T, L.pad_channels(Cc), dtype, device) for _ in range(depth)], 0, [0] * depth]
    idx = ring[1] % len(ring[0])
    ring[1] += 1
    ring[2][idx] += 1
    return RingSlot(ring, idx, ring[2][idx], ring[0][idx])


def as_rows(t: torch.Tensor, B: int, Cc: int, T: int, dtype, name: str, ring_key=None, state: Optional[LossState] = None,
            want_slot: bool = False):
    """Return the RL buffer behind `t` (zero copy when `t` is a rows_view made by this package), else
    pack a plain (B, C, T) tensor into an RL buffer (a rotating persistent one of `state` when `ring_key` is given).
    With want_slot: (buffer, RingSlot or None)."""
    if tuple(t.shape) != (B, Cc, T):
        raise ValueError(f"{name}: expected shape {(B, Cc, T)}, got {tuple(t.shape)}")
    base = _rows_base(t, B, Cc, T, dtype)
    if base is not None:
        return (base, None) if want_slot else base
    if not t.is_cuda:
        raise L.SdaError(f"{name} must live on the MI355X device (there is no CPU path)")
    slot = None
    if ring_key is not None:
        slot = _ring_rows(state or _DEFAULT_STATE, ring_key, B, Cc, T, dtype, t.device)   # pack_rows rewrites every valid row
        buf = slot.buf
    else:
        buf = ops.new_rows(B, T, L.pad_channels(Cc), dtype, t.device)
    ops.pack_rows(t.detach().float().contiguous(), buf)
    return (buf, slot) if want_slot else buf


def _dist_group():
    from .distributed import active_group
    return active_group()


def gather_speech_rows(Yt_local: torch.Tensor, B: int, T: int, group, async_op: bool = False,
                       state: Optional[LossState] = None, slot: Optional[RingSlot] = None):
    """All-gather the packed speech rows of every rank (RCCL all_gather over xGMI; samples are contiguous
    blocks of Tp rows, so the gather lands directly in RL order) together with their squared norms (computed
    once, on the rank that owns the rows).  Returns (Yt, ysq, Bm, col0, B_global, works, slot): `slot` is the ring
    slot that owns the rows the backward will read (the local pack without a group, the gathered buffer with one)."""
    row_elems = L.rows_tp(T) * Yt_local.shape[1]
    ysq_local = ops.rows_sumsq(Yt_local, B, row_elems, row_elems)
    if group is None:
        return Yt_local, ysq_local, B, 0, B, [], slot
    import torch.distributed as dist
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    Tp = L.rows_tp(T)
    from .distributed import emulated_world
    if world == 1 and emulated_world() > 1:
        return _gather_emulated(Yt_local, ysq_local, B, T, group, emulated_world(), async_op, state or _DEFAULT_STATE)
    slot = _ring_rows(state or _DEFAULT_STATE, "loss.Yall", B * world, Yt_local.shape[1], T, Yt_local.dtype, Yt_local.device)
    Yt = slot.buf
    ysq = torch.empty(B * world, dtype=torch.float32, device=Yt_local.device)
    from .distributed import side_group
    bulk = side_group("gather", group)          # own communicator: must not queue in front of the BatchNorm all-reduces
    w1 = dist.all_gather_into_tensor(Yt[: B * world * Tp], Yt_local[: B * Tp], group=bulk, async_op=async_op)
    w2 = dist.all_gather_into_tensor(ysq, ysq_local, group=bulk, async_op=async_op)
    return Yt, ysq, B * world, rank * B, B * world, ([w1, w2] if async_op else []), slot


EMULATE_COPY_REMOTE = True      # emulated world: re-deliver the stand-in rows every step (a device copy in place of the wire)


def _gather_emulated(Yt_local, ysq_local, B, T, group, W, async_op, state: LossState):
    """distributed.emulate_world(W) at world size 1: this rank is rank 0 of W.  Its own rows travel through the real
    all-gather (one rank); the (W - 1) * B rows of the other ranks are resident stand-ins (seeded N(0, 1) rows packed by
    sda_pack_rows, norms by sda_rows_sumsq), copied into the gathered buffer every step on the stream the gather is issued
    from — HBM sees the bytes a real gather would have landed (plus the copy's reads)."""
    import torch.distributed as dist
    from .distributed import side_group
    Tp, Cp, dev, dt = L.rows_tp(T), Yt_local.shape[1], Yt_local.device, Yt_local.dtype
    key = ("remote", B, W, Cp, T, dt, str(dev))
    rem = state.emulated.get(key)
    if rem is None:
        rows = torch.zeros(((W - 1) * B * Tp, Cp), dtype=dt, device=dev)
        nsq = torch.empty((W - 1) * B, dtype=torch.float32, device=dev)
        g = torch.Generator(device=dev).manual_seed(4242)
        tmp = ops.new_rows(B, T, Cp, dt, dev)
        for k in range(W - 1):
            ops.pack_rows(torch.randn((B, Cp, T), generator=g, device=dev), tmp)
            rows[k * B * Tp: (k + 1) * B * Tp].copy_(tmp[: B * Tp])
            nsq[k * B: (k + 1) * B] = ops.rows_sumsq(tmp, B, Tp * Cp, Tp * Cp)
        rem = state.emulated[key] = (rows, nsq)
    slot = _ring_rows(state, "loss.Yall", B * W, Cp, T, dt, dev)
    Yt = slot.buf
    ysq = torch.empty(B * W, dtype=torch.float32, device=dev)
    bulk = side_group("gather", group)
    w1 = dist.all_gather_into_tensor(Yt[: B * Tp], Yt_local[: B * Tp], group=bulk, async_op=async_op)
    w2 = dist.all_gather_into_tensor(ysq[:B], ysq_local, group=bulk, async_op=async_op)
    if EMULATE_COPY_REMOTE or slot.gen == 1:
        Yt[B * Tp: W * B * Tp].copy_(rem[0])
    ysq[B:].copy_(rem[1])
    return Yt, ysq, B * W, 0, B * W, ([w1, w2] if async_op else []), slot


_prefetch_streams = {}    # device -> side stream the speech-side work runs on
PREFETCH_ON_SIDE_STREAM = True


def prefetch_speech(Y: torch.Tensor, dtype=None, global_negatives: bool = True, state: Optional[LossState] = None):
    """Start the speech-side work of the loss EARLY: pack Y into row layout, take its row norms and, under data
    parallelism, launch the all-gather of the packed rows asynchronously on RCCL's stream.  Y does not depend on
    the encoder, so calling this before `brain_encoder(X, ...)` hides the 1.4 GB (8 GPUs, config 3) gather behind
    the forward.  The pack + norms (HBM-bound, ~0.2 ms at config 2) run on a side stream of their own, beside the
    encoder's first layers instead of in front of them."""
    B, F, T = Y.shape
    dtype = dtype or torch.float32
    state = state or _DEFAULT_STATE
    group = _dist_group() if global_negatives else None
    done = None
    if PREFETCH_ON_SIDE_STREAM and Y.is_cuda:
        main = torch.cuda.current_stream(Y.device)
        side = _prefetch_streams.get(str(Y.device))
        if side is None:
            side = _prefetch_streams[str(Y.device)] = torch.cuda.Stream(device=Y.device)
        ev = torch.cuda.Event()
        ev.record(main)
        side.wait_event(ev)                       # Y itself, and the previous user of the ring buffer, are on `main`
        with torch.cuda.stream(side):
            Yt_local, slot = as_rows(Y, B, F, T, dtype, "x (speech embeddings)", ring_key="loss.Y", state=state, want_slot=True)
            Yt, ysq, Bm, col0, Bg, works, slot = gather_speech_rows(Yt_local, B, T, group, async_op=True, state=state, slot=slot)
            done = torch.cuda.Event()
            done.record(side)
        ysq.record_stream(main)
    else:
        Yt_local, slot = as_rows(Y, B, F, T, dtype, "x (speech embeddings)", ring_key="loss.Y", state=state, want_slot=True)
        Yt, ysq, Bm, col0, Bg, works, slot = gather_speech_rows(Yt_local, B, T, group, async_op=True, state=state, slot=slot)
    state.prefetched.clear()
    state.prefetched.append((weakref.ref(Y), dtype, Yt, ysq, Bm, col0, Bg, works, done, slot))


def _take_prefetched(state: LossState, Y, dtype):
    for wy, dt, Yt, ysq, Bm, col0, Bg, works, done, slot in state.prefetched:
        if wy() is Y and dt == dtype:
            state.prefetched.clear()
            if done is not None:
                torch.cuda.current_stream(Yt.device).wait_event(done)
            for work in works:
                work.wait()                     # current stream waits for RCCL's stream; the host does not block
            return Yt, ysq, Bm, col0, Bg, slot
    return None


class _ClipFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, module: "CLIPLoss", Y, Z, temp):
        import torch.distributed as dist
        B, F, T = Z.shape
        if Y.requires_grad:
            raise L.SdaError("CLIPLoss: the speech embeddings (first argument) are data on this path and get no gradient "
                             "(train.py:191 passes precomputed wav2vec2 features); detach them")
        dtype = Z.dtype if Z.dtype in ops.COMPUTE_DTYPES else torch.float32
        Zt = as_rows(Z, B, F, T, dtype, "y (brain embeddings)")
        group = _dist_group() if module.global_negatives else None
        state = module._state
        pre = _take_prefetched(state, Y, dtype)
        if pre is not None:
            Yt, ysq, Bm, col0, Bg, slot = pre
        else:
            Yt_local, slot = as_rows(Y, B, F, T, dtype, "x (speech embeddings)", ring_key="loss.Y", state=state, want_slot=True)
            Yt, ysq, Bm, col0, Bg, _, slot = gather_speech_rows(Yt_local, B, T, group, state=state, slot=slot)
        loss, logits, cnt, cctx = E.clip_forward(Yt, Zt, temp.detach(), Bm=Bm, Bn=B, T=T, col0=col0,
                                                 reduction=module.reduction, B_global=Bg, dist_group=group, ysq=ysq)
        if group is not None:
            # one collective for both: the rank counts (exact in fp32) and the loss share -> global loss for
            # reporting (backward uses the local share)
            both = torch.cat([cnt.to(torch.float32), loss.reshape(1)])
            dist.all_reduce(both, group=group)
            cnt, loss = both[:-1].round().to(torch.int32), both[-1:]
        ctx.cctx, ctx.shape, ctx.dtype, ctx.group = cctx, (B, F, T), dtype, group
        ctx.set_materialize_grads(False)            # no zero-filled gradient for the (non-differentiable) logits output
        ctx.y_slot = slot
        ctx.z_requires_grad = Z.requires_grad
        _cache_ranks(Y, Z, cnt[col0: col0 + B])
        ctx.mark_non_differentiable(logits)
        return loss.reshape(()), logits

    @staticmethod
    def backward(ctx, dloss, _dlogits):
        c = ctx.cctx
        B, F, T = ctx.shape
        dZ = None
        if dloss is None:                           # (the loss itself took no part in what was differentiated)
            return None, None, None, None
        scale = dloss.to(torch.float32)
        if ctx.z_requires_grad:
            if ctx.y_slot is not None and not ctx.y_slot.valid():
                raise L.SdaError("CLIPLoss.backward: the packed speech rows of this forward were recycled by later forwards "
                                 "of the same CLIPLoss (more than `ring_depth` forwards before this backward); raise "
                                 "loss_func._state.ring_depth or call backward earlier")
            # a buffer of its own per backward (two pending backwards must not share one): the GEMM rewrites every sample's
            # rows, pad rows included (exact zeros), so only the slack behind the last sample needs a fill
            dZt = ops.new_rows_uninit(B, T, c.Zt.shape[1], ctx.dtype, c.Zt.device)
            E.clip_backward(c, dZt, scale.reshape(1).contiguous())     # dloss folded into the GEMM epilogue
            dZ = ops.rows_view(dZt, B, F, T)
        dtemp = ops.scalar_mul(c.dtemp, scale.reshape(1))
        return None, None, dZ, dtemp


class CLIPLoss(nn.Module):
    def __init__(self, args):
        super().__init__()
        L.load()
        self.reduction = str(args.reduction)
        if self.reduction not in ("mean", "sum"):
            raise ValueError("reduction must be 'mean' or 'sum'")
        self.temp = nn.Parameter(torch.tensor([float(args.init_temperature)]))
        self.global_negatives = True
        self.last_logits: Optional[torch.Tensor] = None
        self._state = LossState()           # packed-speech ring + pending prefetch of THIS instance

    def prefetch(self, x: torch.Tensor, compute_dtype=torch.float32):
        """Optional: call with the speech embeddings BEFORE running the encoder (see prefetch_speech)."""
        prefetch_speech(x, compute_dtype, self.global_negatives, state=self._state)

    def drain(self):
        """Wait for (and drop) a prefetch that no forward will consume — call before tearing the process group down."""
        self._state.drain()

    def release_buffers(self):
        """Drop the persistent packed-speech buffers (e.g. between a training and an evaluation phase)."""
        self._state.clear()

    def forward(self, x, y, fast=True, return_logits=False):
        batch_size = x.size(0)
        assert batch_size > 1, "Batch size must be greater than 1."               # loss.py:40
        if not fast:
            # loss.py:46-50: logits[i][j] = cos(y_i, x_j) — the fast way's matrix TRANSPOSED and without the learned
            # temperature (which then receives no gradient).  The symmetric loss is the same function of it, so the same
            # kernels run with a zero log-temperature; only the returned logits are turned.
            zero = torch.zeros(1, dtype=torch.float32, device=self.temp.device)
            loss, logits = _ClipFn.apply(self, x, y, zero)
            self.last_logits = logits
            return (logits.t(), loss) if return_logits else loss
        loss, logits = _ClipFn.apply(self, x, y, self.temp)
        self.last_logits = logits
        if return_logits:
            return logits, loss
        return loss


@torch.no_grad()
def retrieval_ranks(Y: torch.Tensor, Z: torch.Tensor, global_candidates: bool = True) -> torch.Tensor:
    """Rank of each speech row's own brain column (0 = top-1).  Reuses the ranks computed by the last
    CLIPLoss forward on the same tensors; otherwise runs the similarity GEMM + rank kernel."""
    hit = _cached_ranks(Y, Z)
    if hit is not None:
        return hit
    B, F, T = Z.shape
    dtype = Z.dtype if Z.dtype in ops.COMPUTE_DTYPES else torch.float32
    Zt = as_rows(Z, B, F, T, dtype, "Z")
    Yt = as_rows(Y, B, F, T, dtype, "Y")
    temp = torch.zeros(1, dtype=torch.float32, device=Zt.device)
    group = _dist_group() if global_candidates else None     # under data parallelism: the GLOBAL batch
    Yt, ysq, Bm, col0, Bg, _, _ = gather_speech_rows(Yt, B, T, group)
    _, _, cnt, _ = E.clip_forward(Yt, Zt, temp, Bm=Bm, Bn=B, T=T, col0=col0, B_global=Bg, dist_group=group, ysq=ysq)
    if group is not None:
        import torch.distributed as dist
        dist.all_reduce(cnt, group=group)
    return cnt[col0: col0 + B]
