"""Host-side orchestration of the HIP kernels for one encoder forward / backward.

Mirrors the data flow of the reference's `BrainEncoder.forward` (models.py:191-196) and of the
backward pass autograd derives from it, but every stage is a libsdamd.so kernel launched on the
current HIP stream over channels-last "row layout" buffers that stay resident in HBM between
forward and backward.  torch supplies memory and streams only.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Dict, List, Optional

import ast
import os

import numpy as np
import torch

from . import lib as L
from . import ops


def block_dilations(k: int):
    """models.py:133,141,149"""
    return 2 ** ((2 * k) % 5), 2 ** ((2 * k + 1) % 5), 2


@dataclass
class EncoderDims:
    C: int
    S: int
    D1: int
    D2: int
    F: int
    K: int

    @property
    def Cp(self): return L.pad_channels(self.C)
    @property
    def D1p(self): return L.pad_channels(self.D1)
    @property
    def D2p(self): return L.pad_channels(self.D2)
    @property
    def F1(self): return 2 * self.D2
    @property
    def F1p(self): return L.pad_channels(2 * self.D2)
    @property
    def Fp(self): return L.pad_channels(self.F)


@dataclass
class EncoderCtx:
    """Everything the backward needs; buffers are owned by the engine's workspace."""
    B: int
    T: int
    gen: int
    training: bool
    bufs: Dict[str, torch.Tensor] = field(default_factory=dict)
    packed: Dict[str, torch.Tensor] = field(default_factory=dict)
    bn: Dict[str, tuple] = field(default_factory=dict)
    mask: Optional[torch.Tensor] = None
    widx: Optional[torch.Tensor] = None
    subj_perm: Optional[torch.Tensor] = None
    subj_seg: Optional[torch.Tensor] = None
    subj_slices: int = 1                                  # K-segments per subject in the per-subject weight gradient
    W_sa: Optional[torch.Tensor] = None
    packed_T: Dict[str, torch.Tensor] = field(default_factory=dict)
    packed_T_ready: Optional[torch.cuda.Event] = None     # side-stream packing of packed_T has finished
    glu_fused: bool = False                               # F.glu ran in conv2's epilogue: bufs hold the gate, not [value | gate]
    composed: Optional[tuple] = None                      # composed SubjectBlock: (Wd, T1aug, Ws) fp32 factors kept for backward


class EncoderEngine:
    def __init__(self, dims: EncoderDims, dtype: torch.dtype = torch.float32, group=None):
        L.load()                          # fail loudly if the HIP extension is missing
        self.d = dims
        self.dtype = dtype
        self.group = group                # torch.distributed group for synchronised BatchNorm statistics
        self.overlap_grad_allreduce = True   # under DP: SUM all-reduce each layer group's gradients as soon as
                                             # they exist, on RCCL's stream, overlapped with the rest of backward
        self._ws: Dict[tuple, torch.Tensor] = {}
        self._ws_shapes: List[tuple] = []    # (space, B, T) groups in least-recently-used order
        self.max_workspace_shapes = 2        # per space: a ragged last batch gets a second set, a third shape evicts the oldest
        self._seg_cache: Dict[tuple, tuple] = {}
        self._gen = 0                        # generation of the TRAIN workspace (bumped by grad-mode forwards only)
        self.reuse_workspace = True
        self.fuse_bn_backward_stats = True   # BatchNorm-backward sums in the data-gradient conv's epilogue
        self.wgrad_target_wgs = 256          # workgroups per weight-gradient launch (split over sample segments)
        # weight-gradient chains (wgrad_gemm -> reduce_slabs -> unpack) depend only on dy and a saved
        # activation, never on each other or on the data-gradient chain: run them on a second HIP stream
        self.wgrad_flat_rows = True          # shared-weight gradients contract a segment's rows as one run (pad rows of dy are zero)
        self.wgrad_side_stream = True
        self.wgrad_reduce_stream = False     # the K-split slab sums behind every weight-gradient GEMM (HBM-bound, 17 launches, 0.36 ms
                                             # of the weight-gradient stream's 3.5 ms in the step) on a THIRD stream, so that the
                                             # next GEMM does not queue behind them: measured 7.06-7.10 vs 6.99-7.07 ms (and 0.4 ms
                                             # more host time per step for the extra events) — off
        self.side_stream_priority = 0        # HIP stream priority of the weight-gradient / packing stream
        self.probe = None                    # diagnostics (tools/stream_waits.py): a list collects (label, event, event) around
                                             # every point where the main stream waits for another stream
        self.pack_on_side_stream = True      # per-step operand packing runs beside the first layers, not in front
        self.forward_pair_tiles = True       # forward k = 3 convs (nothing competes for the CU's LDS there): two
                                             # tiles per workgroup share each weight slab — fewer LDS-DMA bytes per FLOP
        self.flat_tiles_forward = True       # k = 3 convs on the 256-row flat-tile kernel (conv3_flat.hip) where it applies
        self.flat_tiles_forward_fp32 = True  # ... for fp32 storage as well (round 5: the fp32 instantiation runs 128-row tiles only —
                                             # 80 accumulator registers per wave, no scratch; round 4's 256-row form spilled ~300
                                             # registers into its K loop and the exact path went back to the tile kernel)
        self.flat_tile_options = 1024        # extra conv3_flat flags: 1024 = the second workgroup of a CU takes its 128-row tile FIRST
                                             # (the pair's epilogues — HBM bursts with the matrix pipe idle — fall at different
                                             # times: 68.6 -> 67.3 us per 320 -> 320 conv with the priority hand-over, round 4),
                                             # 64 = no priority hand-over between the two (diagnostic)
        self.fuse_glu_forward = True         # F.glu in conv2's epilogue (flat-tile kernel, D2p % 80 == 0): no [value | gate] buffer
        self.bias_sums_at_end = False        # ... all of them at the END of backward, on the weight-gradient stream (round 5: seven
                                             # ~10 us launches leave the main chain — and lengthen the serial tail the optimiser
                                             # waits for: 6.72-6.75 against 6.64-6.68 ms, three alternations; off)
        self.bias_sums_on_side = False       # final reduction of the bias-gradient column sums on the weight-gradient stream
                                             # (round 3: on; re-measured in round 4 against the same library, three alternations on
                                             # one box: 7.15-7.20 ms on, 7.09-7.14 ms off — the short reductions delay the
                                             # weight-gradient GEMMs queued behind them more than they cost the main stream)
        self.fuse_glu_backward = False       # the GLU backward in the epilogue of the conv that produces its incoming gradient
                                             # (needs the fused forward: bufs hold (out, gate)); not with flat-tile data gradients.
                                             # Off: measured 7.87 vs 7.77 ms — the separate pass is HBM-bound and runs beside the
                                             # weight-gradient stream's MFMA work for free, the heavier conv epilogue does not
        self.tail_products_on_side = True    # composed SubjectBlock backward: the parameter-space products only the optimiser reads
                                             # (subj_w, sb_w, sb_b) on the weight-gradient stream, off the chain to dz
        self.subj_wgrad_target_wgs = 0       # workgroups of the PER-SUBJECT weight-gradient launches (0 = wgrad_target_wgs): more
                                             # slices per subject even out the subjects' unequal sample counts
        self.skip_x0_gradient = True         # composed SubjectBlock: its weight gradient straight from block 0's dh0 and X (kernel-3
                                             # per-subject weight gradient + chain rule) instead of conv0's data gradient + dx0 (x) X
        self.compose_subject_block = True    # SpatialAttention, the shared 1x1 conv and the per-subject 1x1 conv as ONE per-subject
                                             # matrix (needs a spare padding channel for the folded bias: C < Cp)
        # backward keeps the 128-row tile kernel (40 KB of LDS per workgroup): the flat kernel's two 76 KB workgroups fill a
        # CU's LDS, the weight-gradient GEMMs of the side stream then wait for the conv instead of running beside it
        # (measured in the step: +2 %; with one flat workgroup per CU: +7 %)
        self.flat_tiles_backward = False
        self.flat_backward_one_per_cu = False
        # the 1x1 projections (conv_final1/2) on conv1_flat.hip's 256-row flat tiles: forward (conv_final2 then leaves per-row
        # partial sums of squares instead of per-tile statistics for ||Z_b||^2), and their data gradients — conv_final2's with
        # the GELU backward of conv_final1 and its bias-gradient column sums in the epilogue (SDA_EPI_GELU_BWD: no
        # gelu_backward_colsum pass over the 640-wide gradient)
        self.bn_backward_store_dg = False    # the data-gradient convs that feed a BatchNorm+GELU backward store dg = dy * GELU'(u) (which
                                             # their statistics epilogue computes anyway) instead of dy: the pass that applies the
                                             # BatchNorm backward does not evaluate GELU' again (SDA_EPI_BN_STORE_DG).  Off: worth
                                             # 0.02 ms of 7.16 (that pass is HBM-bound either way), and the extra rounding of dg to
                                             # bf16 moves the conv2 bias gradients — almost cancelling sums over 92 160 rows — from
                                             # under to over their 6e-2 parity bound at the full batch (7.7e-2)
        self.fuse_gelu_backward_1x1 = False  # conv_final2's data gradient applies conv_final1's GELU backward in its epilogue and keeps
                                             # the bias-gradient column sums (SDA_EPI_GELU_BWD): no pass over the 640-wide gradient.
                                             # Off: measured 7.10 vs 7.04 ms — like the GLU backward below, the separate pass is
                                             # HBM-bound and runs beside the weight-gradient stream for free; vector work added to an
                                             # MFMA kernel's epilogue takes the matrix pipe's issue slots
        # (measured twice: with the step on torch's default stream neither paid — 7.15-7.21 vs 7.16 ms; with the step's chain on
        #  a high-priority stream (streams.py), where a flat kernel gets the CUs it wants when it wants them, the backward pair
        #  is worth 0.10 ms — 6.77-6.87 vs 6.91-6.92, three alternations — and the forward pair nothing: 6.80-6.87 with both)
        self.flat_1x1_forward = False
        self.flat_1x1_backward = True
        # round 5: conv1_wide.hip's 256-row x 256 / 320-channel tiles (eight waves, one persistent workgroup per CU, the next
        # tile's K-steps requested before the epilogue of the current one; 16-bit storage, widths that are multiples of 256 or
        # 320).  Alone at config 2 (tools/probes/bench_1x1_wide.py, us, tile / flat / wide): conv_final2 forward 246 / 253 / 216,
        # conv_final1 forward 105 / 106 / 106, conv_final2's data gradient 211 / 180 / 180, conv_final1's 60 / 46 / 58.  In the
        # step: conv_final2 forward on it (nothing co-runs in forward; -0.03 ms, inside the noise of an alternation); the data
        # gradients on it +0.2 ms (one 160 KB workgroup per CU: the weight-gradient GEMMs wait instead of running beside them)
        self.wide_1x1_forward = True         # conv_final2's forward (widths that are multiples of 256)
        self.wide_1x1_forward_all = False    # ... conv_final1's as well (320-channel tiles: equal to the tile kernel)
        self.wide_1x1_backward = False
        self.dgrad_wave_priority = True      # the backward's tile-kernel data gradients at s_setprio 3 (SDA_CONV_WAVE_PRIO): with the
                                             # HBM-bound passes of the chain raised as well (elementwise.hip, SDA_EW_BWD_PRIO) the
                                             # step's chain wins each SIMD's issue arbitration against the weight-gradient GEMMs it
                                             # runs beside: 6.725 against 6.771 ms, six alternations (either one alone: nothing)
        self.flat_1x1_options = 0            # extra conv1_flat flags (1024 = staggered tile order, 32768 = one workgroup per CU)
        # CU partition for backward (experiment, default off): k > 0 gives the data-gradient chain (the stream backward() is
        # called on hands over to a CU-masked stream) k of the 8 XCDs and the weight-gradient stream the other 8 - k, instead
        # of letting the two streams' workgroups compete for every CU (hipExtStreamCreateWithCUMask)
        self.cu_partition_xcds = 0
        self._part = {}
        self._side = {}
        self._const = {}                     # persistent operand buffers (composed SubjectBlock matrices)

        # diagnostics: SDA_ENGINE_<switch>=<python literal> overrides one of the SWITCHES above (plain bool / int / float
        # attributes set in this constructor: never a property, a buffer table or the dims)
        switches = {k for k, v in vars(self).items() if not k.startswith("_") and isinstance(v, (bool, int, float))}
        for key, val in os.environ.items():
            if not key.startswith("SDA_ENGINE_"):
                continue
            name = key[11:]
            if name not in switches:
                raise L.SdaError(f"{key}: no such engine switch (known: {', '.join(sorted(switches))})")
            try:
                lit = ast.literal_eval(val)
            except (ValueError, SyntaxError) as e:
                raise L.SdaError(f"{key}={val!r} is not a Python literal: {e}") from None
            if not isinstance(lit, (bool, int, float)):
                raise L.SdaError(f"{key}={val!r}: a switch takes a bool, int or float")
            setattr(self, name, lit)

    def _new_side_stream(self, dev) -> torch.cuda.Stream:
        """The weight-gradient / packing stream: torch's pool for priorities it knows (0 normal, -1 high), the C ABI's
        stream for HIP's LOW priority (1), which torch.cuda.Stream cannot express."""
        if self.side_stream_priority <= 0:
            return torch.cuda.Stream(device=dev, priority=self.side_stream_priority)
        with torch.cuda.device(dev):
            return torch.cuda.ExternalStream(ops.stream_create_priority(self.side_stream_priority), device=dev)

    @property
    def world(self) -> int:
        if self.group is None:
            return 1
        import torch.distributed as dist
        return dist.get_world_size(self.group)

    def _allreduce(self, t: torch.Tensor):
        import torch.distributed as dist
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)

    # ------------------------------------------------------------------ workspace
    def _rows(self, name: str, B: int, T: int, Cp: int, device, space: str) -> torch.Tensor:
        key = (space, name, B, T, Cp, self.dtype, str(device))
        if self.reuse_workspace and key in self._ws:
            return self._ws[key]
        buf = ops.new_rows(B, T, Cp, self.dtype, device)
        if self.reuse_workspace:
            self._ws[key] = buf
        return buf

    def _touch_shape(self, space: str, B: int, T: int):
        """LRU over (space, B, T) buffer sets: a training run sees one batch shape plus, with drop_last=False, one
        ragged last batch; anything beyond `max_workspace_shapes` per space releases the least recently used set
        (a pending backward keeps its own buffers alive through its ctx)."""
        g = (space, B, T)
        if g in self._ws_shapes:
            self._ws_shapes.remove(g)
        self._ws_shapes.append(g)
        mine = [x for x in self._ws_shapes if x[0] == space]
        while len(mine) > self.max_workspace_shapes:
            old = mine.pop(0)
            self._ws_shapes.remove(old)
            for k in [k for k in self._ws if (k[0], k[2], k[3]) == old]:
                del self._ws[k]

    def release_workspace(self):
        self._ws.clear()
        self._ws_shapes.clear()
        self._seg_cache.clear()

    def _uniform_segments(self, B: int, ntiles: int, device):
        # segments are dealt round-robin to the 8 XCDs (wgrad_gemm's block order), so use a multiple of 8
        nseg = 8 * max(1, round(self.wgrad_target_wgs / (8 * max(1, ntiles))))
        nseg = int(min(B, nseg)) if B >= 8 else int(max(1, min(B, nseg)))
        key = (B, nseg, str(device))
        if key not in self._seg_cache:
            edges = np.floor(np.linspace(0, B, nseg + 1)).astype(np.int32)
            # (no permutation: consecutive samples; the kernel then needs no index load in its chunk loop)
            self._seg_cache[key] = (None, torch.from_numpy(edges).to(device))
        perm, seg = self._seg_cache[key]
        return perm, seg, nseg

    @property
    def composed(self) -> bool:
        return bool(self.compose_subject_block and self.d.C < self.d.Cp)

    @property
    def glu_fused(self) -> bool:
        return bool(self.fuse_glu_forward and self.flat_forward and self.d.D2p % 80 == 0)

    @property
    def flat_forward(self) -> bool:
        """Forward k = 3 convs on conv3_flat.hip: the 16-bit storage types; fp32 only on request (flat_tiles_forward_fp32)."""
        return bool(self.flat_tiles_forward and (self.dtype != torch.float32 or self.flat_tiles_forward_fp32))

    def _wait(self, label: str, stream, event):
        """stream.wait_event(event); with a probe attached, bracketed by timing events (how long the stream sat idle)."""
        if self.probe is None:
            stream.wait_event(event)
            return
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        stream.wait_event(event)
        e1.record(stream)
        self.probe.append((label, e0, e1))

    # ------------------------------------------------------------------ operand packing plans
    def _plans(self, P, dev):
        key = (str(dev), self.dtype, self.glu_fused, self.composed)
        if getattr(self, "_plan_key", None) == key:
            return self._fwd_plan, self._bwd_plan
        d = self.d
        f, b = ops.PackPlan(self.dtype, dev), ops.PackPlan(self.dtype, dev)
        glu = dict(glu_half=d.D2, glu_half_p=d.D2p)
        glu_fwd = dict(glu, glu_tile=80) if self.glu_fused else glu      # SDA_EPI_GLU's 80 value + 80 gate channels per tile
        if not self.composed:                # (the composed SubjectBlock builds its per-subject operand itself)
            f.add_weight("sb_w", P["sb_w"], d.D1p, d.D1p)
            f.add_vector("sb_b", P["sb_b"], d.D1p)
            f.add_weight("subj_w", P["subj_w"], d.D1p, d.D1p)
            b.add_weight("sb_w", P["sb_w"], d.D1p, d.D1p, mode=1)
            b.add_weight("subj_w", P["subj_w"], d.D1p, d.D1p, mode=1)
        for k in range(5):
            cin_p = d.D1p if k == 0 else d.D2p
            for j in (0, 1):
                ci_p = cin_p if j == 0 else d.D2p
                f.add_weight(f"b{k}.c{j}w", P[f"b{k}.c{j}w"], d.D2p, ci_p)
                f.add_vector(f"b{k}.c{j}b", P[f"b{k}.c{j}b"], d.D2p)
                b.add_weight(f"b{k}.c{j}w", P[f"b{k}.c{j}w"], d.D2p, ci_p, mode=1)
            f.add_weight(f"b{k}.c2w", P[f"b{k}.c2w"], 2 * d.D2p, d.D2p, **glu_fwd)
            f.add_vector(f"b{k}.c2b", P[f"b{k}.c2b"], 2 * d.D2p, **glu_fwd)
            b.add_weight(f"b{k}.c2w", P[f"b{k}.c2w"], 2 * d.D2p, d.D2p, mode=1, **glu)
        f.add_weight("f1w", P["f1w"], d.F1p, d.D2p)
        f.add_vector("f1b", P["f1b"], d.F1p)
        f.add_weight("f2w", P["f2w"], d.Fp, d.F1p)
        f.add_vector("f2b", P["f2b"], d.Fp)
        b.add_weight("f1w", P["f1w"], d.F1p, d.D2p, mode=1)
        b.add_weight("f2w", P["f2w"], d.Fp, d.F1p, mode=1)
        self._fwd_plan, self._bwd_plan, self._plan_key = f, b, key
        return f, b

    # ------------------------------------------------------------------ forward
    def forward(self, P: Dict[str, torch.Tensor], X: torch.Tensor, subject_idxs, *, training: bool,
                mask: Optional[torch.Tensor], need_grad: bool, momentum: float = 0.1, eps: float = 1e-5) -> EncoderCtx:
        d, dt = self.d, self.dtype
        B, Cc, T = X.shape
        assert Cc == d.C, f"expected {d.C} channels, got {Cc}"          # models.py:78
        dev = X.device
        space = "train" if need_grad else "eval"
        if need_grad:                    # no-grad forwards live in the "eval" buffers: they leave a pending backward intact
            self._gen += 1
        self._touch_shape(space, B, T)
        ctx = EncoderCtx(B=B, T=T, gen=self._gen, training=training)
        ctx.glu_fused = self.glu_fused
        bufs, pk = ctx.bufs, ctx.packed

        def rows(name, Cp):
            if not need_grad:            # inference: two ping-pong buffers per width are enough
                name = f"pp{rows.flip.setdefault(Cp, 0) % 3}"
                rows.flip[Cp] += 1
            t = self._rows(name, B, T, Cp, dev, space)
            return t
        rows.flip = {}

        # subject indices: CPU int tensor in the reference (train.py:189); validate like ModuleList indexing
        sidx = torch.as_tensor(subject_idxs).detach().to("cpu").to(torch.int64).numpy()
        if sidx.shape != (B,):
            raise ValueError("subject_idxs must have shape (B,)")
        if (sidx < 0).any() or (sidx >= d.S).any():
            raise IndexError("subject index out of range")                # ModuleList semantics, models.py:115
        up = ops.UPLOADER.upload                # pinned staging: no implicit host<->stream synchronisation
        ctx.widx = up(("widx", space), sidx.astype(np.int32), dev)
        if need_grad:
            # per-subject weight gradient: samples sorted by subject, one K-segment per (slice j, subject s) in
            # j-major order.  With many subjects one slice each is enough (the S segments already fill the GPU);
            # with few (S = 1 in configs 1/4) every subject's samples are cut into r slices so that the launch still
            # has ~wgrad_target_wgs workgroups, and the r slabs of a subject are summed afterwards in fixed order.
            tile_m = 160 if d.D1p % 160 == 0 else (128 if d.D1p % 128 == 0 else 64)
            ntiles = (d.D1p // tile_m) * (d.D1p // (128 if d.D1p % 128 == 0 else 64))
            if self.composed:       # the per-subject gradient is the k = 3 one of (block 0's dh0, X): D2p x 64-channel tiles
                tm = 160 if d.D2p % 160 == 0 else (128 if d.D2p % 128 == 0 else 64)
                ntiles = (d.D2p // tm) * (d.Cp // 64)
            # (sized by the subjects PRESENT in the batch: a batch drawn from a few recordings — 8 of 27 subjects — otherwise
            # runs the launch on a third of its workgroups: +0.3 ms per step, measured with the resident feed in round 5)
            present = max(1, int(np.unique(sidx).size))
            r = int(max(1, min(max(1, B // present), round((self.subj_wgrad_target_wgs or self.wgrad_target_wgs) / max(1, ntiles * present)))))
            perm, seg = subject_segments(sidx, d.S, r)
            ctx.subj_perm = up("subj_perm", perm, dev)
            ctx.subj_seg = up("subj_seg", seg, dev)
            ctx.subj_slices = r
        ctx.mask = mask

        # ---- operand packing (fp32 master weights -> compute dtype, K-contiguous, zero padded): ONE launch
        fwd_plan, bwd_plan = self._plans(P, dev)
        main = torch.cuda.current_stream(dev)
        packed_ready = None
        if self.pack_on_side_stream:
            side = self._side.get(str(dev))
            if side is None:
                side = self._side[str(dev)] = self._new_side_stream(dev)
            composed = self.composed
            Xt = rows("Xt", d.Cp)                # (before the event: a first-use buffer is zero-filled on the MAIN stream)
            ev = torch.cuda.Event()
            ev.record(main)                      # the optimiser's update of P is on the main stream
            side.wait_event(ev)
            with torch.cuda.stream(side):
                pk.update(fwd_plan.run(P))
                # the input's layout change rides on the same stream: the main stream meanwhile computes the SpatialAttention
                # weights and composes the SubjectBlock matrices (parameter-sized work that needs neither)
                ops.pack_rows(X, Xt, ones_channel=d.C if composed else None)
                X.record_stream(side)
                packed_ready = torch.cuda.Event()
                packed_ready.record(side)
                if need_grad:
                    ctx.packed_T = bwd_plan.run(P)      # [tap][ci][co] operands of the data-gradient convs
                    ctx.packed_T_ready = torch.cuda.Event()
                    ctx.packed_T_ready.record(side)
        else:
            pk.update(fwd_plan.run(P))
            if need_grad:
                ctx.packed_T = bwd_plan.run(P)
            composed = self.composed
            Xt = rows("Xt", d.Cp)
            ops.pack_rows(X, Xt, ones_channel=d.C if composed else None)
        k3_flags = L.CONV_PAIR_TILES if self.forward_pair_tiles else 0
        if self.flat_forward:
            k3_flags |= L.CONV_FLAT_TILES | self.flat_tile_options

        # ---- SubjectBlock (models.py:111-117)
        bufs["Xt"] = Xt
        # (composed SubjectBlock: the SpatialAttention weights are wanted in fp32 — the "packed operand" W * mask then IS the
        # fp32 matrix the composition below multiplies)
        W_sa, Wp = ops.sa_weights_forward(P["z"], P["cos"], P["sin"], mask, d.D1p, d.Cp, torch.float32 if composed else dt,
                                          fwd_table=P.get("sa_tab_f"))
        ctx.W_sa = W_sa
        if composed:
            # models.py:111-117 is three linear maps in a row with nothing between them: x0 = W_subj[s] (W_sb (W_sa X) + b_sb).
            # Composed per subject in fp32 parameter space — (S, D1, C + 1) with the bias riding on Xt's constant channel —
            # one per-sample-weight GEMM replaces three, and the backward needs no data gradient at all here (X is an input):
            # one per-subject weight gradient, then the chain rule on (S, D1, C)-sized matrices.  All of these small products
            # run on ops.param_gemm (exact-fp32 MFMA on strided views: no padding, packing, cat or copy around them).
            Wd = Wp[0, 0, : d.D1, : d.C]                                                         # W_sa * mask, (D1, C) view
            Ws = P["subj_w"][..., 0]                                                             # (S, D1, D1) view
            T1aug = torch.empty((d.D1, d.C + 1), dtype=torch.float32, device=dev)               # [W_sb W_d | b_sb]
            ops.param_gemm(P["sb_w"][..., 0], Wd, out=T1aug[:, : d.C])
            ops.copy3d(T1aug[:, d.C:], P["sb_b"][:, None])
            key = ("wtot", str(dev), dt)
            Wtot = self._const.get(key)
            if Wtot is None:
                Wtot = self._const[key] = torch.zeros((d.S, 1, d.D1p, d.Cp), dtype=dt, device=dev)
            ops.param_gemm(Ws, T1aug, out=Wtot[:, 0, : d.D1, : d.C + 1])                      # rounded to the compute dtype on the way out
            if need_grad:
                ctx.composed = (Wd, T1aug, Ws)
            if packed_ready is not None:
                self._wait("packed operands (forward)", main, packed_ready)
            x = ops.conv_gemm(Xt, Wtot, rows("x0", d.D1p), B=B, T=T, KS=1, dil=0, widx=ctx.widx,
                              alg_dims=(d.C, d.D1))
            bufs["x0"] = x
        else:
            if packed_ready is not None:
                self._wait("packed operands (forward)", main, packed_ready)
            h_sa = ops.conv_gemm(Xt, Wp, rows("h_sa", d.D1p), B=B, T=T, KS=1, dil=0, alg_dims=(d.C, d.D1))
            bufs["h_sa"] = h_sa
            h_c = ops.conv_gemm(h_sa, pk["sb_w"], rows("h_c", d.D1p), B=B, T=T, KS=1, dil=0, bias=pk["sb_b"],
                                alg_dims=(d.D1, d.D1))
            bufs["h_c"] = h_c
            x = ops.conv_gemm(h_c, pk["subj_w"], rows("x0", d.D1p), B=B, T=T, KS=1, dil=0, widx=ctx.widx, alg_dims=(d.D1, d.D1))
            bufs["x0"] = x

        # ---- 5 ConvBlocks (models.py:152-166)
        ntile = B * ops.n_t_tiles(T)
        world = self.world
        count = float(B) * T * world            # BatchNorm statistics span the GLOBAL batch under data parallelism
        for k in range(5):
            cin_p = d.D1p if k == 0 else d.D2p
            dil = block_dilations(k)
            for j in (0, 1):
                alg = (d.D1 if (k == 0 and j == 0) else d.D2, d.D2)
                pre = f"b{k}.c{j}"
                w, bias = pk[pre + "w"], pk[pre + "b"]
                res = x if (j == 1 or k > 0) else None
                h = rows(f"b{k}.h{j}", d.D2p)
                bnp = f"b{k}.bn{j}"
                if training:
                    nt = ops.conv_stats_rows(B, T, 3, d.D2p, k3_flags)
                    stats = torch.empty((nt, 2, d.D2p), dtype=torch.float32, device=dev)
                    ops.conv_gemm(x, w, h, B=B, T=T, KS=3, dil=dil[j], bias=bias, res=res, stats=stats, alg_dims=alg,
                                  flags=k3_flags)
                    if self.group is not None:   # one 2*Cp-float all-reduce per BatchNorm
                        # (the column-sum kernel, ~10 us; the generic slab sum took 34 us per BatchNorm on this chain)
                        stats = ops.reduce_stats(stats).view(1, 2, d.D2p)
                        self._allreduce(stats)
                        nt = 1
                    mean, rstd, scale, shift, bcoef = ops.bn_finalize(stats, nt, count, P[bnp + "w"], P[bnp + "b"],
                                                                      P[bnp + "rm"], P[bnp + "rv"], d.D2p, True, eps, momentum,
                                                                      want_bwd_coef=True, batches_tracked=P.get(bnp + "nbt"))
                else:
                    ops.conv_gemm(x, w, h, B=B, T=T, KS=3, dil=dil[j], bias=bias, res=res, alg_dims=alg, flags=k3_flags)
                    # (the backward coefficient table only when a backward may follow: eval-mode BatchNorm is then a fixed
                    # per-channel affine map on the running statistics)
                    mean, rstd, scale, shift, *rest = ops.bn_finalize(None, 0, count, P[bnp + "w"], P[bnp + "b"], P[bnp + "rm"],
                                                                      P[bnp + "rv"], d.D2p, False, eps, momentum,
                                                                      want_bwd_coef=need_grad)
                    bcoef = rest[0] if rest else None
                ctx.bn[bnp] = (mean, rstd, bcoef)
                a = ops.bn_gelu_forward(h, rows(f"b{k}.a{j}", d.D2p), scale, shift, B, T)
                bufs[f"b{k}.h{j}"], bufs[f"b{k}.a{j}"] = h, a
                x = a
            w, bias = pk[f"b{k}.c2w"], pk[f"b{k}.c2b"]
            if ctx.glu_fused:        # F.glu in the conv's epilogue: only the product and (for backward) the gate are stored
                gate = rows(f"b{k}.g", d.D2p) if need_grad else None
                x = ops.conv_gemm(x, w, rows(f"x{k + 1}", d.D2p), B=B, T=T, KS=3, dil=dil[2], bias=bias, y_pre=gate,
                                  alg_dims=(d.D2, 2 * d.D2), flags=k3_flags | L.EPI_GLU)
                bufs[f"b{k}.g"], bufs[f"x{k + 1}"] = gate, x
            else:
                c2 = ops.conv_gemm(x, w, rows(f"b{k}.c2", 2 * d.D2p), B=B, T=T, KS=3, dil=dil[2], bias=bias,
                                   alg_dims=(d.D2, 2 * d.D2), flags=k3_flags)
                x = ops.glu_forward(c2, rows(f"x{k + 1}", d.D2p), B, T)
                bufs[f"b{k}.c2"], bufs[f"x{k + 1}"] = c2, x

        # ---- two 1x1 projections with GELU (models.py:194-195)
        u1, g1 = rows("u1", d.F1p), rows("g1", d.F1p)
        f_flags = (L.CONV_FLAT_TILES | self.flat_1x1_options) if self.flat_1x1_forward else 0
        wide_ok = lambda cp: dt != torch.float32 and (cp % 256 == 0 or cp % 320 == 0)      # noqa: E731
        f1_flags = L.CONV_WIDE_TILES if (self.wide_1x1_forward_all and wide_ok(d.F1p)) else f_flags
        f2_wide = bool(self.wide_1x1_forward and dt != torch.float32 and d.Fp % 256 == 0)
        ops.conv_gemm(x, pk["f1w"], g1, B=B, T=T, KS=1, dil=0, bias=pk["f1b"], y_pre=u1 if need_grad else None,
                      gelu=True, alg_dims=(d.D2, d.F1), flags=f1_flags)
        # Z is handed to the caller: a FRESH buffer per forward (the reference returns a new tensor each call), so
        # embeddings kept across forwards stay valid; everything else lives in the reused workspace
        u2, Zt = rows("u2", d.Fp), ops.new_rows_uninit(B, T, d.Fp, dt, dev)
        # ||Z_b||^2 for the loss comes out of the epilogue's sums: no separate pass over Z (loss.py:65)
        if f2_wide or (f_flags and d.Fp % 128 == 0):
            zparts = torch.empty((B * L.rows_tp(T), d.Fp // 128), dtype=torch.float32, device=dev)
            ops.conv_gemm(g1, pk["f2w"], Zt, B=B, T=T, KS=1, dil=0, bias=pk["f2b"], y_pre=u2 if need_grad else None,
                          gelu=True, row_sumsq=zparts, alg_dims=(d.F1, d.F), flags=L.CONV_WIDE_TILES if f2_wide else f_flags)
            ops.ROW_NORMS.put(Zt, ops.rows_sumsq_from_row_parts(zparts, B, T))
        else:
            zstats = torch.empty((B * ops.n_t_tiles(T), 2, d.Fp), dtype=torch.float32, device=dev)
            ops.conv_gemm(g1, pk["f2w"], Zt, B=B, T=T, KS=1, dil=0, bias=pk["f2b"], y_pre=u2 if need_grad else None,
                          gelu=True, stats=zstats, alg_dims=(d.F1, d.F))
            ops.ROW_NORMS.put(Zt, ops.rows_sumsq_from_stats(zstats, B))
        bufs.update(u1=u1, g1=g1, u2=u2, Z=Zt)
        if not need_grad:
            ctx.bufs = {"Z": Zt}
            ctx.packed = {}
        return ctx

    # ------------------------------------------------------------------ backward
    def _partition_streams(self, dev, k: int):
        """(main, side, main_cus): two CU-masked streams.  Bit i of a mask is CU (i // 8) of XCD (i % 8) — the runtime deals
        consecutive indices round-robin to the XCDs, and an XCD whose share of the mask is EMPTY runs unrestricted
        (tools/probes/cumask_probe.py: masks `i % 8 < 1` and `i % 8 < 4` change nothing, `i < 128` halves every XCD) — so a
        partition is a subset of every XCD's CUs: the first k / 8 of each XCD's 32 (k = 5: 20 CUs x 8 XCDs) against the rest.
        k = 8 + m (diagnostic): both streams unrestricted."""
        key = (str(dev), k)
        if key not in self._part:
            cus = torch.cuda.get_device_properties(dev).multi_processor_count
            per_xcd = cus // 8
            bits_a = [1 if (i // 8) < (per_xcd * k) // 8 else 0 for i in range(cus)]
            if k >= 8:
                bits_a = [1] * cus
            if k == 9:        # (diagnostic) the same hand-over on two ordinary streams
                self._part[key] = (torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev), cus)
                return self._part[key]
            streams = []
            for bits in (bits_a, [1 - b for b in bits_a] if k < 8 else bits_a):
                words = [sum(bits[32 * w + j] << j for j in range(32) if 32 * w + j < cus) for w in range((cus + 31) // 32)]
                streams.append(torch.cuda.ExternalStream(ops.stream_create_cumask(words), device=dev))
            self._part[key] = (streams[0], streams[1], sum(bits_a))
        return self._part[key]

    def backward(self, P: Dict[str, torch.Tensor], ctx: EncoderCtx, dZt: torch.Tensor) -> Dict[str, torch.Tensor]:
        k = int(self.cu_partition_xcds)
        if not (0 < k <= 9 and self.wgrad_side_stream):
            return self._backward(P, ctx, dZt)
        dev = dZt.device
        outer = torch.cuda.current_stream(dev)
        main_p, side_p, main_cus = self._partition_streams(dev, k)
        ev = torch.cuda.Event()
        ev.record(outer)
        main_p.wait_event(ev)
        saved_side = self._side.get(str(dev))
        self._side[str(dev)] = side_p
        old_limit = L.load().sda_set_cu_limit(main_cus)          # persistent grids launched below size themselves for the partition
        try:
            with torch.cuda.stream(main_p):
                grads = self._backward(P, ctx, dZt)
                done = torch.cuda.Event()
                done.record(main_p)
        finally:
            L.load().sda_set_cu_limit(old_limit)
            if saved_side is not None:
                self._side[str(dev)] = saved_side
            else:
                self._side.pop(str(dev), None)
        # (no record_stream: blocks allocated on main_p are consumed on `outer` strictly behind `done`, and the next backward's
        # main_p work starts behind an event of `outer` — record_stream would only make the allocator hold them back)
        outer.wait_event(done)
        return grads

    def _backward(self, P: Dict[str, torch.Tensor], ctx: EncoderCtx, dZt: torch.Tensor) -> Dict[str, torch.Tensor]:
        if ctx.gen != self._gen and self.reuse_workspace:
            raise L.SdaError("the activation workspace of this forward was overwritten by a later forward of the same "
                             "encoder; call backward before the next training-mode forward, or set "
                             "engine.reuse_workspace = False")
        d, dt = self.d, self.dtype
        B, T, bufs = ctx.B, ctx.T, ctx.bufs
        dev = dZt.device
        if getattr(ctx, "packed_T_ready", None) is not None:
            self._wait("packed operands (backward)", torch.cuda.current_stream(dev), ctx.packed_T_ready)
        grads: Dict[str, torch.Tensor] = {}
        scratch = ops.reduce_scratch(max(d.Fp, 2 * d.D2p, d.F1p), dev)
        pending = []                          # (work, names) of in-flight gradient all-reduces
        overlap = self.group is not None and self.overlap_grad_allreduce

        def flush(names):
            """Pack the named gradients into one flat bucket, start its SUM all-reduce asynchronously and
            re-point the gradients at views of the bucket (no copy back)."""
            if not overlap:
                return
            import torch.distributed as dist
            from .distributed import side_group
            flats = [torch.view_as_real(grads[n]).reshape(-1) if grads[n].is_complex() else grads[n].reshape(-1) for n in names]

            def pack_and_reduce():
                # on the SIDE stream, where the weight gradients of this group are produced: the main stream (the
                # critical path) never waits for them; on_side() makes the side stream wait for the few gradients
                # that come from the main stream (biases, BatchNorm affine)
                b = torch.cat(flats)
                if side is not None:
                    for f in flats:               # sources made on the main stream are read here, on the side stream
                        f.record_stream(side)
                pending.append(dist.all_reduce(b, op=dist.ReduceOp.SUM, group=side_group("grads", self.group), async_op=True))
                return b
            if red is not None:                   # the weight gradients of this group come from the slab-sum stream
                ev = torch.cuda.Event()
                ev.record(red)
                side.wait_event(ev)
            bucket = on_side(pack_and_reduce)
            off = 0
            for n, f in zip(names, flats):
                v = bucket[off: off + f.numel()]
                grads[n] = torch.view_as_complex(v.view(*grads[n].shape, 2)) if grads[n].is_complex() else v.view(grads[n].shape)
                off += f.numel()

        def tmp(name, Cp):
            return self._rows("bw." + name, B, T, Cp, dev, "train")

        main = torch.cuda.current_stream(dev)
        side = None
        if self.wgrad_side_stream:
            side = self._side.get(str(dev))
            if side is None:
                side = self._side[str(dev)] = self._new_side_stream(dev)

        def on_side(fn):
            """Run `fn` (launches + allocations) on the side stream once everything queued on the main stream
            so far is done; outputs are handed back to the main stream by join_side()."""
            if side is None:
                return fn()
            ev = torch.cuda.Event()
            ev.record(main)
            side.wait_event(ev)
            # (set_stream both ways instead of the torch.cuda.stream() context: ~40 calls per backward, and the context
            # manager costs the host 15-20 us a time; backward() is entered on `main` and nothing in `fn` leaves a third stream current)
            torch.cuda.set_stream(side)
            try:
                out = fn()
            finally:
                torch.cuda.set_stream(main)
            for t in (out if isinstance(out, (tuple, list)) else (out,)):
                t.record_stream(main)
            return out

        red = None
        if side is not None and self.wgrad_reduce_stream:
            red = self._side.get(str(dev) + "/reduce")
            if red is None:
                red = self._side[str(dev) + "/reduce"] = torch.cuda.Stream(device=dev)

        def join_side():
            if side is not None:
                ev = torch.cuda.Event()
                ev.record(side)
                self._wait("weight-gradient stream joined", main, ev)
            if red is not None:
                ev = torch.cuda.Event()
                ev.record(red)
                self._wait("slab-sum stream joined", main, ev)

        def wgrad(dy, x, KS, dil, Cout, Cin, **glu):
            Cout_p, Cin_p = dy.shape[1], x.shape[1]
            tile_m = 160 if Cout_p % 160 == 0 else (128 if Cout_p % 128 == 0 else 64)
            perm, seg, nseg = self._uniform_segments(B, (Cout_p // tile_m) * (Cin_p // 64), dev)

            def chain():
                slabs = ops.wgrad_gemm(dy, x, B=B, T=T, KS=KS, dil=dil, perm=perm, seg_start=seg, nseg=nseg,
                                       alg_dims=(Cin, Cout), flat_rows=self.wgrad_flat_rows)
                if red is None:
                    return ops.reduce_unpack_wgrad(slabs, Cout, Cin, KS, **glu)
                ev = torch.cuda.Event()
                ev.record(side)
                red.wait_event(ev)
                with torch.cuda.stream(red):
                    out = ops.reduce_unpack_wgrad(slabs, Cout, Cin, KS, **glu)
                slabs.record_stream(red)
                return out
            return on_side(chain)

        ntile = B * ops.n_t_tiles(T)

        W0cat = None
        if ctx.composed is not None and self.skip_x0_gradient:
            # block 0's conv0 weights as [d][tap][o] (o zero-padded to D2p; the padding is written once), the left operand of the
            # composed SubjectBlock's gradient at the END of backward: a parameter-only copy, so it is made here, on the
            # weight-gradient stream while that has nothing to do, not in the serial tail of the step
            key = ("w0cat", str(dev))
            W0cat = self._const.get(key)
            if W0cat is None:
                W0cat = self._const[key] = torch.zeros((d.D1, 3, d.D2p), dtype=torch.float32, device=dev)
            _w0 = W0cat
            on_side(lambda: ops.copy3d(_w0[:, :, : d.D2], P["b0.c0w"].permute(1, 2, 0)))
            # its reader (the G product at the end of backward) runs on the MAIN stream: an event of the side stream right
            # behind the copy, waited for just before that product (long complete by then: the wait costs nothing, and the
            # result no longer depends on how far the side stream has got)
            w0cat_ready = None
            if side is not None:
                w0cat_ready = torch.cuda.Event()
                w0cat_ready.record(side)

        def dgrad(dy, key, w_fp32, Cout_p, Cin_p, out, KS, dil, res=None, widx=None, bn=None, glu_bwd=None, **glu):
            """Data-gradient conv.  bn = (h, coef): `out` is the gradient entering GELU(BN(h)); the conv's epilogue
            then also emits the per-tile BatchNorm-backward sums (returned as second value) — the separate
            reduction pass over (out, h) is not needed.  glu_bwd = (x_out, gate) of the block BELOW: the conv's output is
            the gradient entering that block's F.glu, and the epilogue writes the GLU backward [d value | d gate] into `out`
            (twice as wide) plus the per-tile column sums of both halves (second value) instead of the gradient itself."""
            bflags = (L.CONV_FLAT_TILES | (L.CONV_ONE_PER_CU if self.flat_backward_one_per_cu else 0)) if (self.flat_tiles_backward and KS == 3) else 0
            if self.dgrad_wave_priority and not bflags:
                bflags |= L.CONV_WAVE_PRIO
            if glu_bwd is not None:
                st = torch.empty((ops.conv_stats_rows(B, T, KS, Cin_p, 0), 2, Cin_p), dtype=torch.float32, device=dev)
                ops.conv_gemm(dy, ctx.packed_T[key], out, B=B, T=T, KS=KS, dil=dil, res=res, widx=widx, stats=st, glu_bwd=glu_bwd,
                              alg_dims=(w_fp32.shape[-3], w_fp32.shape[-2]))
                return out, st
            if bn is None or not self.fuse_bn_backward_stats or bn[1] is None:
                return ops.conv_gemm(dy, ctx.packed_T[key], out, B=B, T=T, KS=KS, dil=dil, res=res, widx=widx,
                                     alg_dims=(w_fp32.shape[-3], w_fp32.shape[-2]), flags=bflags), None
            st = torch.empty((ops.conv_stats_rows(B, T, KS, out.shape[1], bflags), 2, out.shape[1]), dtype=torch.float32, device=dev)
            ops.conv_gemm(dy, ctx.packed_T[key], out, B=B, T=T, KS=KS, dil=dil, res=res, widx=widx, stats=st,
                          bn_x=bn[0], bn_coef=bn[1], alg_dims=(w_fp32.shape[-3], w_fp32.shape[-2]),
                          flags=bflags | (L.EPI_BN_STORE_DG if self.bn_backward_store_dg else 0))
            return out, st

        def bias_grad(cs, C, glu_half=0, glu_half_p=0):
            """Padded column sums -> bias gradient: a view when the unpadded vector is a prefix of the padded one (no GLU
            split, or GLU halves without padding between them); the un-packing kernel otherwise."""
            if glu_half == 0 or glu_half_p == glu_half:
                return cs[:C]
            return ops.unpack_vector(cs, C, glu_half, glu_half_p)

        # ---- final projections
        # the last stage of a bias gradient (partial rows -> column sums) feeds nothing on the main stream: it goes to the
        # weight-gradient stream, each launch with partial rows of its own
        deferred_sums = []                    # (gradient name, finish closure): final reductions postponed to the end of backward

        def colsum_on_side(fn, *args, width, then=None, name=None):
            """Column sums of fn's output; `then` (the bias-gradient un-packing, where it needs a kernel) runs on the SAME
            stream as the sums' final reduction, right behind it: on the main stream it would read them before they exist.
            bias_sums_at_end: the partial rows are kept and the final reduction of every bias gradient runs once, on the
            weight-gradient stream behind its last GEMM — seven ~10 us launches leave the main chain (nothing on it reads a bias
            gradient); not under overlapped gradient all-reduces, which want each layer group's gradients as they appear."""
            if name is not None and side is not None and self.bias_sums_at_end and not overlap:
                finish = fn(*args, B, T, ops.reduce_scratch(width, dev), defer=True)
                deferred_sums.append((name, (lambda: then(finish())) if then is not None else finish))
                return None
            if side is None or not self.bias_sums_on_side:
                cs = fn(*args, B, T, scratch)
                return then(cs) if then is not None else cs
            finish = fn(*args, B, T, ops.reduce_scratch(width, dev), defer=True)
            return on_side((lambda: then(finish())) if then is not None else finish)

        def finish_deferred_sums():
            if not deferred_sums:
                return
            def run():
                return [fin() for _, fin in deferred_sums]
            outs = on_side(run)
            for (nm, _), g_ in zip(deferred_sums, outs):
                grads[nm] = g_

        du2 = tmp("du2", d.Fp)
        grads["f2b"] = colsum_on_side(ops.gelu_backward_colsum, bufs["u2"], dZt, du2, width=d.Fp, then=lambda cs: bias_grad(cs, d.F), name="f2b")
        b1_flags = (L.CONV_FLAT_TILES | self.flat_1x1_options) if self.flat_1x1_backward else 0
        wide_ok = lambda cp: dt != torch.float32 and (cp % 256 == 0 or cp % 320 == 0)      # noqa: E731
        b2_flags = b1_flags                   # conv_final2's data gradient (width F1p) / conv_final1's (width D2p)
        if self.wide_1x1_backward and wide_ok(d.F1p):
            b2_flags = L.CONV_WIDE_TILES
        b1w_flags = L.CONV_WIDE_TILES if (self.wide_1x1_backward and wide_ok(d.D2p)) else b1_flags
        du1 = tmp("du1", d.F1p)
        if self.fuse_gelu_backward_1x1 or (b2_flags & L.CONV_WIDE_TILES) or (b1_flags and (d.F1p % 160 == 0 or d.F1p % 128 == 0)):
            # conv_final2's data gradient with conv_final1's GELU backward in its epilogue: du1 directly, plus per-unit column sums
            gst = torch.empty((ops.conv_stats_rows(B, T, 1, d.F1p, b2_flags | L.EPI_GELU_BWD), 2, d.F1p), dtype=torch.float32, device=dev)
            ops.conv_gemm(du2, ctx.packed_T["f2w"], du1, B=B, T=T, KS=1, dil=0, gelu_bwd_u=bufs["u1"], stats=gst,
                          alg_dims=(d.F, d.F1), flags=b2_flags)
            grads["f2w"] = wgrad(du2, bufs["g1"], 1, 0, d.F, d.F1)
            grads["f1b"] = on_side(lambda: bias_grad(ops.reduce_stats(gst)[:d.F1p], d.F1))
        else:
            dg1, _ = dgrad(du2, "f2w", P["f2w"], d.Fp, d.F1p, tmp("dg1", d.F1p), 1, 0)
            grads["f2w"] = wgrad(du2, bufs["g1"], 1, 0, d.F, d.F1)
            grads["f1b"] = colsum_on_side(ops.gelu_backward_colsum, bufs["u1"], dg1, du1, width=d.F1p, then=lambda cs: bias_grad(cs, d.F1), name="f1b")
        # Where the forward kept (out, gate) of every F.glu, the conv that produces the gradient entering a block's GLU (this
        # 1x1 data gradient for block 4, conv0's data gradient of block k + 1 for block k) applies the GLU backward in its
        # epilogue: `glu_pending` = (dc2, per-tile column sums) for the block about to be processed, and dx is never stored
        glu_in_epilogue = bool(ctx.glu_fused and self.fuse_glu_backward and not self.flat_tiles_backward)
        glu_pending = None
        if glu_in_epilogue:
            glu_pending = dgrad(du1, "f1w", P["f1w"], d.F1p, d.D2p, tmp("dc2.4", 2 * d.D2p), 1, 0, glu_bwd=(bufs["x5"], bufs["b4.g"]))
            dx = None
        else:
            if b1w_flags:
                dx = ops.conv_gemm(du1, ctx.packed_T["f1w"], tmp("dxA", d.D2p), B=B, T=T, KS=1, dil=0, alg_dims=(d.F1, d.D2), flags=b1w_flags)
            else:
                dx, _ = dgrad(du1, "f1w", P["f1w"], d.F1p, d.D2p, tmp("dxA", d.D2p), 1, 0)
        grads["f1w"] = wgrad(du1, bufs["x5"], 1, 0, d.F1, d.D2)
        flush(["f2w", "f2b", "f1w", "f1b"])

        # ---- ConvBlocks, last to first
        # the gradient of the ten biases that feed a training-mode BatchNorm (identically zero): rows of ONE fresh zero buffer
        # per backward — autograd hands these views to the parameters' .grad, so a buffer kept across steps would alias
        # engine-owned memory into .grad (an in-place clip with a non-finite factor would poison every later step)
        null_bias = ops.zeros((10, d.D2), torch.float32, dev) if ctx.training else None
        flip = 0
        for k in range(4, -1, -1):
            cin, cin_p = (d.D1, d.D1p) if k == 0 else (d.D2, d.D2p)
            dil = block_dilations(k)
            glu = dict(glu_half=d.D2, glu_half_p=d.D2p)
            c2b = lambda cs: bias_grad(cs, 2 * d.D2, **glu)          # noqa: E731  (a kernel when D2 is not a multiple of 64)
            if glu_pending is not None:
                dc2, gst = glu_pending
                grads[f"b{k}.c2b"] = c2b(ops.reduce_stats(gst))   # [sum d value | sum d gate] over all rows
            else:
                dc2 = tmp(f"dc2.{k}", 2 * d.D2p)      # per-layer buffers: a side-stream wgrad may still read them
                if ctx.glu_fused:
                    grads[f"b{k}.c2b"] = colsum_on_side(ops.glu_backward_colsum_og, bufs[f"x{k + 1}"], bufs[f"b{k}.g"], dx, dc2,
                                                        width=2 * d.D2p, then=c2b, name=f"b{k}.c2b")
                else:
                    grads[f"b{k}.c2b"] = c2b(ops.glu_backward_colsum(bufs[f"b{k}.c2"], dx, dc2, B, T, scratch))
            da1, tstats = dgrad(dc2, f"b{k}.c2w", P[f"b{k}.c2w"], 2 * d.D2p, d.D2p, tmp("da", d.D2p), 3, dil[2],
                                bn=(bufs[f"b{k}.h1"], ctx.bn[f"b{k}.bn1"][2]), **glu)
            # the weight-gradient chain is queued AFTER the data-gradient conv: on the side stream it then runs
            # beside the HBM-bound BatchNorm backward kernels that follow, not beside the MFMA-bound conv
            grads[f"b{k}.c2w"] = wgrad(dc2, bufs[f"b{k}.a1"], 3, dil[2], 2 * d.D2, d.D2, **glu)
            x_in = bufs[f"x{k}"]
            for j in (1, 0):
                bnp = f"b{k}.bn{j}"
                mean, rstd, _ = ctx.bn[bnp]
                dh = tmp(f"dh.{k}.{j}", d.D2p)
                world = self.world
                # eval-mode BatchNorm (running statistics) has no batch-statistics terms in its input gradient: an infinite
                # count zeroes them (dbeta / N, dgamma / N) while dgamma / dbeta themselves stay the plain sums, rank-local
                sync = self.group is not None and ctx.training
                dgam, dbet = ops.bn_gelu_backward(da1, bufs[f"b{k}.h{j}"], mean, rstd, P[bnp + "w"], P[bnp + "b"], dh, B, T,
                                                  scratch, count=float(B) * T * world if ctx.training else float("inf"),
                                                  allreduce=self._allreduce if sync else None, tile_stats=tstats,
                                                  dy_is_dg=bool(self.bn_backward_store_dg and tstats is not None))
                # under DP the sums are already global on every rank; the gradient all-reduce (SUM) follows
                if sync:
                    # (one launch for both rows, on the weight-gradient stream: only the optimiser reads these)
                    both = dgam._base if dgam._base is not None else torch.stack([dgam, dbet])
                    if side is not None:
                        both.record_stream(side)
                    scaled = on_side(lambda both=both: both / world)
                    dgam, dbet = scaled[0], scaled[1]
                grads[bnp + "w"], grads[bnp + "b"] = dgam[: d.D2], dbet[: d.D2]
                src = bufs[f"b{k}.a0"] if j == 1 else x_in
                ci, ci_p = (d.D2, d.D2p) if j == 1 else (cin, cin_p)
                # conv0/conv1 feed a training-mode BatchNorm, which removes any per-channel constant: the bias
                # gradient is identically zero (the reference's autograd reports rounding noise there)
                grads[f"b{k}.c{j}b"] = null_bias[2 * k + j] if ctx.training else bias_grad(ops.colsum(dh, B, T, scratch), d.D2)
                res = dh if (j == 1 or k > 0) else None
                out = tmp("da", d.D2p) if j == 1 else tmp("dxB" if flip == 0 else "dxA", ci_p)
                if k == 0 and j == 0 and ctx.composed is not None and self.skip_x0_gradient:
                    # the composed SubjectBlock takes its weight gradient straight from dh0 and X (below): the gradient with
                    # respect to x0 is never needed, this data-gradient conv is not run
                    da1, tstats, dh0 = None, None, dh
                elif j == 0 and k > 0 and glu_in_epilogue:     # its output is the gradient entering block k - 1's GLU
                    glu_pending = dgrad(dh, f"b{k}.c{j}w", P[f"b{k}.c{j}w"], d.D2p, ci_p, tmp(f"dc2.{k - 1}", 2 * d.D2p), 3, dil[j],
                                        res=res, glu_bwd=(bufs[f"x{k}"], bufs[f"b{k - 1}.g"]))
                    da1, tstats = None, None
                else:
                    da1, tstats = dgrad(dh, f"b{k}.c{j}w", P[f"b{k}.c{j}w"], d.D2p, ci_p, out, 3, dil[j], res=res,
                                        bn=(bufs[f"b{k}.h0"], ctx.bn[f"b{k}.bn0"][2]) if j == 1 else None)
                grads[f"b{k}.c{j}w"] = wgrad(dh, src, 3, dil[j], d.D2, ci)
            dx = da1
            flip ^= 1
            flush([f"b{k}.c2w", f"b{k}.c2b", f"b{k}.c1w", f"b{k}.c1b", f"b{k}.bn1w", f"b{k}.bn1b",
                   f"b{k}.c0w", f"b{k}.c0b", f"b{k}.bn0w", f"b{k}.bn0b"])

        # ---- SubjectBlock
        dhs = dx                                            # (rows, D1p)
        def subj_wgrad():
            r = ctx.subj_slices
            slabs = ops.wgrad_gemm(dhs, bufs["h_c"], B=B, T=T, KS=1, dil=0, perm=ctx.subj_perm, seg_start=ctx.subj_seg,
                                   nseg=r * d.S, flat_rows=self.wgrad_flat_rows)   # (r*S, 1, D1p, D1p), slice-major
            if r > 1:
                slabs = ops.reduce_slabs(slabs.view(r, -1)).view(d.S, 1, d.D1p, d.D1p)
            return ops.unpack_conv_wgrad(slabs, d.S, d.D1, d.D1, 1, d.D1p, d.D1p)
        if ctx.composed is not None:
            Wd, T1aug, Ws = ctx.composed
            r = ctx.subj_slices
            # x0 = W_tot[s] X feeds block 0's conv0 and nothing else, and h0 = sum_tap W0[tap] x0[t + (tap - 1) dil], so
            #   dL/dW_tot[s] = sum_tap W0[tap]^T M[s][tap],   M[s][tap] = sum_{b in s, t} dh0[b, t] (x) X[b, t + (tap - 1) dil]
            # = the per-subject kernel-3 weight gradient of (dh0, X) followed by one batched (D1 x 3 D2) . (3 D2 x C+1) product:
            # neither conv0's data gradient (a 320 -> 320 kernel-3 conv) nor a weight gradient over dx0 is computed.
            if self.skip_x0_gradient:
                M = ops.wgrad_gemm(dh0, bufs["Xt"], B=B, T=T, KS=3, dil=block_dilations(0)[0], perm=ctx.subj_perm,
                                   seg_start=ctx.subj_seg, nseg=r * d.S, flat_rows=self.wgrad_flat_rows)     # (r*S, 3, D2p, Cp); column C: the folded bias
                if r > 1:
                    M = ops.reduce_slabs(M.view(r, -1))
                if w0cat_ready is not None:
                    self._wait("W0cat copy (side stream)", main, w0cat_ready)
                G = ops.param_gemm(W0cat.view(d.D1, 3 * d.D2p), M.view(d.S, 3 * d.D2p, d.Cp)[:, :, : d.C + 1])   # (S, D1, C + 1)
            else:
                slabs = ops.wgrad_gemm(dhs, bufs["Xt"], B=B, T=T, KS=1, dil=0, perm=ctx.subj_perm, seg_start=ctx.subj_seg,
                                       nseg=r * d.S, flat_rows=self.wgrad_flat_rows)   # (r*S, 1, D1p, Cp): dL/dW_tot[s], column C = dL/db_tot[s]
                if r > 1:
                    slabs = ops.reduce_slabs(slabs.view(r, -1))
                G = slabs.view(d.S, d.D1p, d.Cp)[:, : d.D1, : d.C + 1]                            # (S, D1, C + 1) view
            # The chain to the last gradient of the step is M -> G -> W_subj^T G -> dT1 -> dWd -> dz; the products that only
            # the optimiser reads (subj_w, sb_w, sb_b) leave it for the weight-gradient stream (tail_products_on_side)
            off_chain = on_side if self.tail_products_on_side else (lambda fn: fn())
            grads["subj_w"] = off_chain(lambda: ops.param_gemm(G, T1aug.t()).view(d.S, d.D1, d.D1, 1))   # W_tot[s] = W_subj[s] T1aug
            part = ops.param_gemm(Ws.transpose(1, 2), G)                                         # W_subj[s]^T G[s] per subject ...
            dT1f = ops.reduce_slabs(part.view(d.S, -1)).view(d.D1, d.C + 1)                      # ... summed in subject order
            dT1 = dT1f[:, : d.C]
            grads["sb_b"], grads["sb_w"] = off_chain(lambda: (
                ops.copy3d(torch.empty(d.D1, dtype=torch.float32, device=dev), dT1f[:, d.C]),
                ops.param_gemm(dT1, Wd.t()).unsqueeze(-1)))
            dWd = ops.param_gemm(P["sb_w"][..., 0].t(), dT1)
            grads["z"] = ops.sa_weights_backward(dWd, ctx.W_sa, ctx.mask, P["cosT"], P["sinT"], P["z"].shape[1],
                                                 bwd_table=P.get("sa_tab_b"))
            flush(["subj_w", "sb_w", "sb_b", "z"])
            finish_deferred_sums()
            join_side()
            for work in pending:
                work.wait()
            return grads
        grads["subj_w"] = on_side(subj_wgrad)
        dh_c, _ = dgrad(dhs, "subj_w", P["subj_w"], d.D1p, d.D1p, tmp("dh_c", d.D1p), 1, 0, widx=ctx.widx)
        grads["sb_w"] = wgrad(dh_c, bufs["h_sa"], 1, 0, d.D1, d.D1)
        grads["sb_b"] = bias_grad(ops.colsum(dh_c, B, T, scratch), d.D1)
        dh_sa, _ = dgrad(dh_c, "sb_w", P["sb_w"], d.D1p, d.D1p, tmp("dh_sa", d.D1p), 1, 0)
        Cout_p, Cin_p = d.D1p, d.Cp
        tile_m = 160 if Cout_p % 160 == 0 else (128 if Cout_p % 128 == 0 else 64)
        perm, seg, nseg = self._uniform_segments(B, (Cout_p // tile_m) * (Cin_p // 64), dev)
        dWd = ops.reduce_slabs(ops.wgrad_gemm(dh_sa, bufs["Xt"], B=B, T=T, KS=1, dil=0, perm=perm, seg_start=seg, nseg=nseg))
        grads["z"] = ops.sa_weights_backward(dWd, ctx.W_sa, ctx.mask, P["cosT"], P["sinT"], P["z"].shape[1],
                                             bwd_table=P.get("sa_tab_b"))
        flush(["subj_w", "sb_w", "sb_b", "z"])
        finish_deferred_sums()
        join_side()
        for work in pending:
            work.wait()                       # makes the current stream wait for RCCL's; no host sync
        return grads


def subject_segments(sidx: np.ndarray, S: int, r: int):
    """K-segments of the per-subject weight gradient: samples sorted by subject (stable), each subject's run cut
    into `r` nearly equal slices, segments listed slice-major (segment j*S + s = slice j of subject s) so that the
    r slabs of one subject are `r` equally strided blocks for the ordered slab sum.
    Returns (perm int32 [B], seg_start int32 [r*S + 1])."""
    order = np.argsort(sidx, kind="stable").astype(np.int32)
    bounds = np.searchsorted(sidx[order], np.arange(S + 1)).astype(np.int64)
    lo, hi = bounds[:-1], bounds[1:]
    cuts = [lo + ((hi - lo) * j) // r for j in range(r + 1)]              # r + 1 arrays of S cut points
    seg = np.empty(r * S + 1, dtype=np.int32)
    parts, pos = [], 0
    for j in range(r):
        for s in range(S):
            a_, b_ = int(cuts[j][s]), int(cuts[j + 1][s])
            seg[j * S + s] = pos
            parts.append(order[a_:b_])
            pos += b_ - a_
    seg[r * S] = pos
    perm = np.concatenate(parts).astype(np.int32) if parts else order
    return perm, seg


# ----------------------------------------------------------------------------------------------- loss
@dataclass
class ClipCtx:
    Bm: int
    Bn: int
    col0: int
    G: torch.Tensor
    rscale: torch.Tensor
    Yt: torch.Tensor
    Zt: torch.Tensor
    row_elems: int
    dtemp: torch.Tensor
    cscale: Optional[torch.Tensor] = None     # per-column factor taken out of G (see sda_clip_grad): the dZ GEMM's acc_scale


def clip_forward(Yt: torch.Tensor, Zt: torch.Tensor, temp: torch.Tensor, *, Bm: int, Bn: int, T: int, col0: int = 0,
                 reduction: str = "mean", B_global: Optional[int] = None, dist_group=None, want_grad: bool = True,
                 ysq: Optional[torch.Tensor] = None):
    """CLIP loss (loss.py:58-79) on RL embeddings: Yt holds the Bm (global) speech rows, Zt the Bn local
    brain rows.  Returns (loss_local_share, logits, ranks_count, ctx).  With `dist_group`, row statistics
    and the diagonal are merged across ranks so that the negatives span the global batch."""
    st = clip_block_stats(Yt, Zt, temp, Bm=Bm, Bn=Bn, T=T, col0=col0, ysq=ysq)
    if dist_group is None:                       # one process: the block IS the whole matrix, its row lse came with the logits
        row_lse, diag = st.row_lse, st.diag
    else:
        from .distributed import merge_row_softmax_stats
        row_lse, diag = merge_row_softmax_stats(st.row_max, st.row_sum, dist_group, diag=st.diag)   # diag: zero where not owned
    return clip_block_finish(st, row_lse, diag, reduction=reduction, B_global=B_global)


@dataclass
class ClipBlockStats:
    """One rank's column block before the cross-rank merge: logits (Bm, Bn) and what the merge needs."""
    Yt: torch.Tensor
    Zt: torch.Tensor
    temp: torch.Tensor
    Bm: int
    Bn: int
    col0: int
    row_elems: int
    ysq: torch.Tensor
    zsq: torch.Tensor
    logits: torch.Tensor
    row_max: torch.Tensor
    row_sum: torch.Tensor
    col_lse: torch.Tensor
    diag: torch.Tensor
    row_lse: torch.Tensor          # lse of each row over this block's columns only


def clip_block_stats(Yt, Zt, temp, *, Bm: int, Bn: int, T: int, col0: int = 0, ysq=None) -> ClipBlockStats:
    """First half of clip_forward: norms, the (Bm x Bn) logits block of this rank's brain columns, per-row (max, sum exp)
    over the block, per-column lse over all rows, the positives' logits.  Rank-local: no collective."""
    Fp = Zt.shape[1]
    row_elems = L.rows_tp(T) * Fp
    if ysq is None:                              # (under DP the caller gathers the per-rank norms instead)
        ysq = ops.rows_sumsq(Yt, Bm, row_elems, row_elems)
    zsq = ops.ROW_NORMS.get(Zt, Bn)              # left by the encoder's last conv when Zt is its output buffer
    if zsq is None:
        zsq = ops.rows_sumsq(Zt, Bn, row_elems, row_elems)
    S = ops.matmul_nt_splitk(Yt, Zt, Bm, Bn, row_elems, row_elems)
    logits, row_max, row_sum, col_lse, diag, row_lse = ops.clip_logits_stats(S, ysq, zsq, temp, Bm, Bn, col0)
    return ClipBlockStats(Yt, Zt, temp, Bm, Bn, col0, row_elems, ysq, zsq, logits, row_max, row_sum, col_lse, diag, row_lse)


def clip_block_finish(st: ClipBlockStats, row_lse, diag, *, reduction: str = "mean", B_global: Optional[int] = None):
    """Second half, given the row lse over the columns of ALL ranks and the positives' logits (after the merge): this rank's
    share of the loss and of d loss / d temp, the gradient coefficient matrix, the retrieval rank counts."""
    Yt, Zt, temp, Bm, Bn, col0, row_elems = st.Yt, st.Zt, st.temp, st.Bm, st.Bn, st.col0, st.row_elems
    logits, col_lse, ysq, zsq = st.logits, st.col_lse, st.ysq, st.zsq
    Bg = B_global if B_global is not None else Bm
    inv_norm = 1.0 / (2.0 * Bg) if reduction == "mean" else 0.5
    # G is an MFMA operand (16-bit in the 16-bit modes): it holds the O(1) part of dL/dlogits * exp(temp) / (|Y||Z|), the
    # rest — 1e-8 at the 8-GPU shapes, far outside fp16 — comes back as a per-column fp32 factor in the dZ GEMM's epilogue
    G, rscale, cscale, scalars = ops.clip_grad(logits, row_lse, col_lse, ysq, zsq, temp, inv_norm, col0, Yt.dtype)
    cnt = ops.clip_ranks(logits, diag, col0)
    ctx = ClipCtx(Bm=Bm, Bn=Bn, col0=col0, G=G, rscale=rscale, Yt=Yt, Zt=Zt, row_elems=row_elems, dtemp=scalars[1:2],
                  cscale=cscale)
    return scalars[0:1], logits, cnt, ctx


def clip_backward(ctx: ClipCtx, dZt: torch.Tensor, dloss: Optional[torch.Tensor] = None) -> torch.Tensor:
    """dZ = dloss * (diag(c) G^T Y - diag(r) Z)  (gradient of the loss share w.r.t. the local brain embeddings)."""
    return ops.clip_dz(ctx.G, ctx.Yt, ctx.Zt, dZt, ctx.rscale, ctx.cscale, Bm=ctx.Bm, Bn=ctx.Bn, row_elems=ctx.row_elems, out_scale=dloss)
