"""Thin torch-tensor wrappers over the C ABI (include/sd_amd.h).  torch is used only for device
memory and the current HIP stream; every computation happens inside libsdamd.so."""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

import torch

from . import lib as L

_DT = {torch.float32: L.F32, torch.bfloat16: L.BF16, torch.float16: L.F16}
COMPUTE_DTYPES = tuple(_DT)


def dt_code(dtype: torch.dtype) -> int:
    try:
        return _DT[dtype]
    except KeyError:
        raise L.SdaError(f"unsupported compute dtype {dtype}; use torch.float32, torch.bfloat16 or torch.float16")


def _p(t: Optional[torch.Tensor]):
    return None if t is None else t.data_ptr()


# The stream every launch goes to = torch's current stream of the current device.  torch.cuda.current_stream().cuda_stream builds
# a Stream object through three Python layers (4-5 us; ~165 launches per step); the two C entry points below are what it ends in.
_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)
_cur_device = getattr(torch._C, "_cuda_getDevice", None)


def _st():
    if _raw_stream is not None and _cur_device is not None:
        return _raw_stream(_cur_device())
    return torch.cuda.current_stream().cuda_stream


def stream_create_cumask(words) -> int:
    """A HIP stream restricted to the CUs set in `words` (32 CUs per word); returns the hipStream_t as an integer for
    torch.cuda.ExternalStream.  Created in this process; kept in _CREATED_STREAMS until stream_destroy_all()."""
    arr = (C.c_uint32 * len(words))(*[int(w) & 0xFFFFFFFF for w in words])
    out = C.c_void_p()
    L.check(L.load().sda_stream_create_cumask(arr, len(words), C.byref(out)), "stream_create_cumask")
    _CREATED_STREAMS.append(int(out.value))
    return int(out.value)


def stream_create_priority(priority: int) -> int:
    """A non-blocking HIP stream of `priority` (-1 high, 0 normal, 1 LOW: the last is below anything torch's stream pool
    hands out); returns the hipStream_t as an integer for torch.cuda.ExternalStream (destroyed by stream_destroy_all())."""
    out = C.c_void_p()
    L.check(L.load().sda_stream_create_priority(int(priority), C.byref(out)), "stream_create_priority")
    _CREATED_STREAMS.append(int(out.value))
    return int(out.value)


_CREATED_STREAMS = []       # hipStream_t handles made through the C ABI (torch's ExternalStream does not own them)


def stream_destroy_all():
    """Destroy every stream stream_create_cumask / stream_create_priority handed out (call once the ExternalStreams that
    wrap them are no longer used: end of a run).  Idempotent."""
    lib = L.load()
    while _CREATED_STREAMS:
        lib.sda_stream_destroy(C.c_void_p(_CREATED_STREAMS.pop()))


def _need_cuda(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise L.SdaError("libsdamd kernels need device tensors (no CPU fallback)")


def upload_small(array, device) -> torch.Tensor:
    """Host numpy array -> new device tensor without a memcpy (payload travels in kernel arguments)."""
    import numpy as np
    array = np.ascontiguousarray(array)
    nbytes = array.nbytes
    padded = (nbytes + 3) // 4 * 4
    raw = np.zeros(padded, dtype=np.uint8)
    raw[:nbytes] = array.view(np.uint8).reshape(-1)
    out = torch.empty(padded, dtype=torch.uint8, device=device)
    L.check(L.load().sda_upload_words(out.data_ptr(), raw.ctypes.data, padded // 4, _st()), "upload_words")
    tdt = torch.from_numpy(np.empty(0, dtype=array.dtype)).dtype
    return out[:nbytes].view(tdt).reshape(array.shape)


class UploadCache:
    """Small host->device tables (index tables, masks, descriptor tables) keyed by CONTENT.  An upload from
    pageable memory costs the stream ~0.2 ms of idle time (staging round trip when the stream reaches it), and
    most of these tables repeat (one dropout mask per sensor; descriptor tables; recurring batches), so a hit
    returns the device tensor uploaded earlier and nothing is enqueued."""

    def __init__(self, capacity: int = 1024):
        import collections
        self.capacity = capacity
        self.items = collections.OrderedDict()

    def upload(self, key, array, device) -> torch.Tensor:
        import numpy as np
        array = np.ascontiguousarray(array)
        k = (key, str(array.dtype), array.shape, array.tobytes(), str(device))
        hit = self.items.get(k)
        if hit is not None:
            self.items.move_to_end(k)
            return hit
        dev = upload_small(array, device)
        self.items[k] = dev
        if len(self.items) > self.capacity:
            self.items.popitem(last=False)
        return dev


UPLOADER = UploadCache()


def new_rows(B: int, T: int, Cp: int, dtype, device) -> torch.Tensor:
    """Zero-initialised RL buffer: (rows_alloc, Cp)."""
    return torch.zeros((L.rows_alloc(B, T), Cp), dtype=dtype, device=device)


def new_rows_uninit(B: int, T: int, Cp: int, dtype, device) -> torch.Tensor:
    """RL buffer whose VALID rows are left uninitialised (the producing kernel writes all of them); the pad rows in
    front of every sample and the slack behind the last one are zeroed (two small fills instead of a full memset)."""
    buf = torch.empty((L.rows_alloc(B, T), Cp), dtype=dtype, device=device)
    return zero_pad_rows(buf, B, T)


def zero_pad_rows(buf: torch.Tensor, B: int, T: int) -> torch.Tensor:
    """Zero the pad rows in front of every sample and the slack behind the last one of an RL buffer (one small launch)."""
    L.check(L.load().sda_zero_pad_rows(_p(buf), B, T, buf.shape[1], dt_code(buf.dtype), _st()), "zero_pad_rows")
    return buf


def zeros(shape, dtype, device) -> torch.Tensor:
    """A fresh zero tensor filled by sda_fill_zero (no framework fill kernel inside the step)."""
    out = torch.empty(shape, dtype=dtype, device=device)
    if out.numel():
        L.check(L.load().sda_fill_zero(out.data_ptr(), out.numel() * out.element_size(), _st()), "fill_zero")
    return out


def gather_samples(table: torch.Tensor, idx: torch.Tensor, B: int, T: int) -> torch.Tensor:
    """table: row-layout buffer holding N samples back to back ((N * Tp [+ slack], Cp), pad rows zero); idx: B int64 sample
    indices on the device.  Returns a fresh row-layout buffer (rows_alloc(B, T), Cp) whose sample b is table sample idx[b]."""
    _need_cuda(table, idx)
    if idx.dtype != torch.int64 or idx.numel() != B:
        raise L.SdaError("gather_samples: idx must hold B int64 indices")
    Cp = table.shape[1]
    out = new_rows_uninit(B, T, Cp, table.dtype, table.device)
    L.check(L.load().sda_gather_samples(_p(table), _p(idx), _p(out), B, L.rows_tp(T) * Cp * table.element_size(), _st()), "gather_samples")
    return out


def clip_merge_rows(allp: torch.Tensor):
    """allp: the all-gathered (world, 3, Bg) fp32 table of per-rank (row max, row sum exp, positive's logit) -> (lse, diag)."""
    world, k, Bg = allp.shape
    if k != 3 or allp.dtype != torch.float32 or not allp.is_contiguous():
        raise L.SdaError("clip_merge_rows: a contiguous fp32 (world, 3, B_global) table")
    out = torch.empty((2, Bg), dtype=torch.float32, device=allp.device)
    L.check(L.load().sda_clip_merge_rows(_p(allp), world, Bg, _p(out[0]), _p(out[1]), _st()), "clip_merge_rows")
    return out[0], out[1]


def scalar_mul(a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    """a * b[0] for small fp32 device tensors (a: n elements, b: one)."""
    out = torch.empty_like(a)
    L.check(L.load().sda_scalar_mul(_p(a), _p(b), _p(out), a.numel(), _st()), "scalar_mul")
    return out


def rows_view(buf: torch.Tensor, B: int, C: int, T: int) -> torch.Tensor:
    """(B, C, T) strided view of an RL buffer (zero copy): element (b, c, t) = buf[b*Tp + PAD + t, c]."""
    Tp, Cp = L.rows_tp(T), buf.shape[1]
    return buf.as_strided((B, C, T), (Tp * Cp, 1, Cp), L.ROW_PAD * Cp)


def pack_rows(src: torch.Tensor, dst: torch.Tensor, ones_channel: Optional[int] = None):
    """(B, C, T) fp32 -> row layout; with `ones_channel` that padding channel is 1 on every valid row."""
    _need_cuda(src, dst)
    B, Cc, T = src.shape
    src = src.contiguous().float()
    if ones_channel is None:
        L.check(L.load().sda_pack_rows(_p(src), _p(dst), B, Cc, T, dst.shape[1], dt_code(dst.dtype), _st()), "pack_rows")
    else:
        L.check(L.load().sda_pack_rows_ones(_p(src), _p(dst), B, Cc, T, dst.shape[1], ones_channel, dt_code(dst.dtype), _st()),
                "pack_rows_ones")


def unpack_rows(src: torch.Tensor, B: int, Cc: int, T: int) -> torch.Tensor:
    _need_cuda(src)
    out = torch.empty((B, Cc, T), dtype=torch.float32, device=src.device)
    L.check(L.load().sda_unpack_rows(_p(src), _p(out), B, Cc, T, src.shape[1], dt_code(src.dtype), _st()), "unpack_rows")
    return out


def rows_sumsq(x: torch.Tensor, B: int, row_elems: int, pitch: int) -> torch.Tensor:
    out = torch.empty(B, dtype=torch.float32, device=x.device)
    scratch = torch.empty(B * 64, dtype=torch.float32, device=x.device)
    L.check(L.load().sda_rows_sumsq(_p(x), _p(out), _p(scratch), B, row_elems, pitch, dt_code(x.dtype), _st()), "rows_sumsq")
    return out


def rows_sumsq_from_row_parts(parts: torch.Tensor, B: int, T: int) -> torch.Tensor:
    """Per-sample sum of squares from the per-row partial sums conv_gemm(row_sumsq=parts) wrote."""
    out = torch.empty(B, dtype=torch.float32, device=parts.device)
    L.check(L.load().sda_rows_sumsq_from_row_parts(_p(parts), parts.shape[1], _p(out), B, T, _st()), "rows_sumsq_from_row_parts")
    return out


def rows_sumsq_from_stats(stats: torch.Tensor, B: int) -> torch.Tensor:
    """Per-sample sum of squares of a conv output from the per-tile statistics its epilogue wrote (plane 1)."""
    out = torch.empty(B, dtype=torch.float32, device=stats.device)
    L.check(L.load().sda_rows_sumsq_from_stats(_p(stats), stats.shape[0] // B, stats.shape[2], _p(out), B, _st()),
            "rows_sumsq_from_stats")
    return out


class _NormCache:
    """Squared norms of RL buffers whose producer already summed them (the encoder's last conv).  An entry is valid only
    while the PRODUCER'S tensor object is alive (a freed buffer's address can be handed to an unrelated tensor), for
    the same storage at the same torch version counter: an in-place torch op on the tensor (or any view of it)
    invalidates it; the producing kernel refreshes it on every forward."""

    def __init__(self):
        self.items = {}

    def put(self, buf: torch.Tensor, sumsq: torch.Tensor):
        import weakref
        if len(self.items) > 16:
            self.items = {k: v for k, v in self.items.items() if v[0]() is not None}
        self.items[buf.data_ptr()] = (weakref.ref(buf), buf._version, buf.shape[0], sumsq)

    def get(self, buf: torch.Tensor, B: int):
        hit = self.items.get(buf.data_ptr())
        if hit is None:
            return None
        owner = hit[0]()
        if (owner is None or owner.data_ptr() != buf.data_ptr() or owner._version != hit[1] or buf._version != hit[1]
                or hit[2] != buf.shape[0] or hit[3].numel() != B):
            return None
        return hit[3]


ROW_NORMS = _NormCache()


def pack_conv_weight(w: torch.Tensor, Cout_p: int, Cin_p: int, dtype, mode: int = 0, glu_half: int = 0,
                     glu_half_p: int = 0, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """w: (nW, Cout, Cin, KS) or (Cout, Cin, KS) fp32 -> packed operand (nW, KS, rows, cols)."""
    if w.dim() == 3:
        w = w.unsqueeze(0)
    nW, Cout, Cin, KS = w.shape
    w = w.contiguous()
    rows, cols = (Cout_p, Cin_p) if mode == 0 else (Cin_p, Cout_p)
    if out is None:
        out = torch.empty((nW, KS, rows, cols), dtype=dtype, device=w.device)
    L.check(L.load().sda_pack_conv_weight(_p(w), _p(out), nW, Cout, Cin, KS, Cout_p, Cin_p, mode, glu_half, glu_half_p,
                                          dt_code(dtype), _st()), "pack_conv_weight")
    return out


class PackPlan:
    """A fixed list of operand packs (weights -> compute dtype / padded layout, biases -> padded fp32) executed by
    ONE kernel launch.  Sources are the fp32 master parameters (stable addresses), destinations persistent
    buffers; the descriptor table lives on the device and is rebuilt only if a source pointer changes."""

    def __init__(self, dtype, device):
        self.dtype, self.device = dtype, device
        self.items = []          # (key, src tensor, dst tensor, desc fields)
        self.out = {}
        self._table = None
        self._sig = None

    def add_weight(self, key, w, Cout_p, Cin_p, mode=0, glu_half=0, glu_half_p=0, glu_tile=0):
        if w.dim() == 3:
            w = w.unsqueeze(0)
        nW, Cout, Cin, KS = w.shape
        rows, cols = (Cout_p, Cin_p) if mode == 0 else (Cin_p, Cout_p)
        dst = torch.empty((nW, KS, rows, cols), dtype=self.dtype, device=self.device)
        self.items.append((key, w, dst, dict(nW=nW, Cout=Cout, Cin=Cin, KS=KS, Cout_p=Cout_p, Cin_p=Cin_p, mode=mode,
                                             glu_half=glu_half, glu_half_p=glu_half_p, glu_tile=glu_tile, is_vector=0,
                                             total=dst.numel())))
        self.out[key] = dst
        return dst

    def add_vector(self, key, v, Cp, glu_half=0, glu_half_p=0, glu_tile=0):
        dst = torch.empty(Cp, dtype=torch.float32, device=self.device)
        self.items.append((key, v, dst, dict(nW=1, Cout=v.numel(), Cin=1, KS=1, Cout_p=Cp, Cin_p=1, mode=0,
                                             glu_half=glu_half, glu_half_p=glu_half_p, glu_tile=glu_tile, is_vector=1, total=Cp)))
        self.out[key] = dst
        return dst

    def run(self, sources):
        """sources: {key: current fp32 tensor} (same shapes as at add time)."""
        sig = tuple(sources[k].data_ptr() for k, *_ in self.items)
        if sig != self._sig:
            arr = (L.PackDesc * len(self.items))()
            for i, (k, _, dst, f) in enumerate(self.items):
                src = sources[k]
                if not src.is_contiguous():
                    raise L.SdaError(f"pack plan: parameter {k} is not contiguous")
                arr[i].src, arr[i].dst = src.data_ptr(), dst.data_ptr()
                for name, val in f.items():
                    setattr(arr[i], name, val)
            import numpy as np
            self._table = upload_small(np.frombuffer(bytes(arr), dtype=np.uint8), self.device)
            self._sig = sig
            self._max_total = max(f["total"] for *_, f in self.items)
        L.check(L.load().sda_pack_multi(_p(self._table), len(self.items), self._max_total, dt_code(self.dtype), _st()), "pack_multi")
        return self.out


def reduce_unpack_wgrad(slabs, Cout, Cin, KS, glu_half=0, glu_half_p=0) -> torch.Tensor:
    """slabs (nseg, KS, Cout_p, Cin_p) fp32 -> parameter-layout gradient (Cout, Cin, KS) fp32."""
    out = torch.empty((Cout, Cin, KS), dtype=torch.float32, device=slabs.device)
    L.check(L.load().sda_reduce_unpack_wgrad(_p(slabs), slabs.shape[0], _p(out), Cout, Cin, KS, slabs.shape[2], slabs.shape[3],
                                             glu_half, glu_half_p, _st()), "reduce_unpack_wgrad")
    return out


def pack_vector(v: torch.Tensor, Cp: int, glu_half: int = 0, glu_half_p: int = 0) -> torch.Tensor:
    out = torch.empty(Cp, dtype=torch.float32, device=v.device)
    L.check(L.load().sda_pack_vector(_p(v.contiguous()), _p(out), v.numel(), Cp, glu_half, glu_half_p, _st()), "pack_vector")
    return out


def unpack_conv_wgrad(g: torch.Tensor, nW, Cout, Cin, KS, Cout_p, Cin_p, glu_half=0, glu_half_p=0) -> torch.Tensor:
    out = torch.empty((nW, Cout, Cin, KS), dtype=torch.float32, device=g.device)
    L.check(L.load().sda_unpack_conv_wgrad(_p(g), _p(out), nW, Cout, Cin, KS, Cout_p, Cin_p, glu_half, glu_half_p, _st()),
            "unpack_conv_wgrad")
    return out


def unpack_vector(g: torch.Tensor, Cc: int, glu_half=0, glu_half_p=0) -> torch.Tensor:
    out = torch.empty(Cc, dtype=torch.float32, device=g.device)
    L.check(L.load().sda_unpack_vector(_p(g), _p(out), Cc, g.numel(), glu_half, glu_half_p, _st()), "unpack_vector")
    return out


def param_gemm(A: torch.Tensor, B: torch.Tensor, out: Optional[torch.Tensor] = None, out_dtype=torch.float32) -> torch.Tensor:
    """out[b] = A[b] @ B[b] for fp32 operands of ANY strides — (M, K) @ (K, N), or batched (b, M, K) @ (b, K, N) where a 2-d
    operand is shared by every batch member (views, transposes, slices: nothing is copied or made contiguous).  `out` may be
    a view too (fp32, or the compute dtype: the product is then rounded on the way out); by default a new fp32 tensor."""
    _need_cuda(A, B, out)
    if A.dtype != torch.float32 or B.dtype != torch.float32:
        raise L.SdaError("param_gemm: operands must be fp32")
    batch = A.shape[0] if A.dim() == 3 else (B.shape[0] if B.dim() == 3 else 1)
    M, K = A.shape[-2], A.shape[-1]
    N = B.shape[-1]
    if B.shape[-2] != K or (A.dim() == 3 and B.dim() == 3 and A.shape[0] != B.shape[0]):
        raise L.SdaError(f"param_gemm: shapes {tuple(A.shape)} x {tuple(B.shape)} do not multiply")
    batched = A.dim() == 3 or B.dim() == 3
    if out is None:
        out = torch.empty(((batch, M, N) if batched else (M, N)), dtype=out_dtype, device=A.device)
    if tuple(out.shape[-2:]) != (M, N) or (out.dim() == 3) != batched or (batched and out.shape[0] != batch):
        raise L.SdaError(f"param_gemm: output shape {tuple(out.shape)} does not match ({batch}, {M}, {N})")
    a = L.PgemmArgs()
    a.A, a.B, a.C = A.data_ptr(), B.data_ptr(), out.data_ptr()
    a.M, a.N, a.K, a.batch = M, N, K, batch
    a.a_i, a.a_k, a.a_b = A.stride(-2), A.stride(-1), (A.stride(0) if A.dim() == 3 else 0)
    a.b_k, a.b_j, a.b_b = B.stride(-2), B.stride(-1), (B.stride(0) if B.dim() == 3 else 0)
    a.c_i, a.c_j, a.c_b = out.stride(-2), out.stride(-1), (out.stride(0) if out.dim() == 3 else 0)
    a.c_dtype = dt_code(out.dtype)
    L.check(L.load().sda_param_gemm(C.byref(a), _st()), "param_gemm")
    return out


def copy3d(dst: torch.Tensor, src: torch.Tensor) -> torch.Tensor:
    """dst[...] = src[...] for fp32 views of equal shape (up to 3 dims) and any strides — one small kernel instead of a
    framework copy (column inserts, permuted packs of a parameter)."""
    _need_cuda(dst, src)
    if dst.shape != src.shape or dst.dim() > 3 or dst.dtype != torch.float32 or src.dtype != torch.float32:
        raise L.SdaError("copy3d: fp32 views of equal shape with at most three dimensions")
    pad = 3 - dst.dim()
    n = [1] * pad + list(dst.shape)
    d = [0] * pad + list(dst.stride())
    s_ = [0] * pad + list(src.stride())
    L.check(L.load().sda_copy3d(dst.data_ptr(), d[0], d[1], d[2], src.data_ptr(), s_[0], s_[1], s_[2], n[0], n[1], n[2], _st()), "copy3d")
    return dst


class KernelTimer:
    """Optional HIP-event timing of individual launches on the current stream (bench.py's roofline leg).
    Off by default: `ops.TIMER = KernelTimer()` turns it on, `ops.TIMER = None` off."""

    def __init__(self, families=("conv_gemm",)):
        self.records = []           # (key, algorithmic_flops, start_event, end_event)
        self.families = tuple(families)   # which kernel families get event pairs ("conv_gemm", "wgrad_gemm")

    def summary(self):
        torch.cuda.synchronize()
        out = {}
        for key, flops, e0, e1 in self.records:
            n, f, ms = out.get(key, (0, 0.0, 0.0))
            out[key] = (n + 1, f + flops, ms + e0.elapsed_time(e1))
        return out


TIMER: Optional[KernelTimer] = None


def conv_tile_co(Cout_p: int, KS: int = 3, stats: bool = False) -> int:
    """Output-channel tile sda_conv_gemm picks (labels of the kernel timer; mirrors dispatch_conv in conv_gemm.hip)."""
    if KS == 1 and Cout_p % 128 == 0 and not stats:
        return 128
    return 160 if Cout_p % 160 == 0 else (128 if Cout_p % 128 == 0 else 64)


def conv_gemm(x, w, y, *, B, T, KS, dil, bias=None, res=None, y_pre=None, widx=None, stats=None, gelu=False,
              alg_dims=None, flags=0, bn_x=None, bn_coef=None, glu_bwd=None, gelu_bwd_u=None, row_sumsq=None):
    """RL conv: x (rows, Cin_p), w (nW, KS, Cout_p, Cin_p) packed, y (rows, Cout_p).
    alg_dims = (Cin, Cout) unpadded, only used to count algorithmic FLOPs when the timer is on.
    gelu_bwd_u (flat 1x1 only, SDA_EPI_GELU_BWD): y = round(conv) * GELU'(gelu_bwd_u), `stats` = per-unit column sums.
    row_sumsq (flat 1x1 only, SDA_EPI_ROW_SUMSQ): float (rows, Cout_p / 128) buffer of per-row partial sums of squares."""
    _need_cuda(x, w, y)
    a = L.ConvArgs()
    if gelu_bwd_u is not None:
        if bn_x is not None or gelu_bwd_u.shape != y.shape or stats is None:
            raise L.SdaError("conv_gemm: gelu_bwd_u needs the shape of y and a stats buffer")
        flags |= L.EPI_GELU_BWD
        bn_x = gelu_bwd_u
    if row_sumsq is not None:
        if stats is not None or row_sumsq.dtype != torch.float32 or row_sumsq.shape[0] < B * L.rows_tp(T) \
                or y.shape[1] % 128 or row_sumsq.shape[1] != y.shape[1] // 128:
            raise L.SdaError("conv_gemm: row_sumsq must be float (rows, Cout_p / 128) and excludes stats")
        flags |= L.EPI_ROW_SUMSQ
        stats = row_sumsq
    a.x, a.w, a.bias, a.res, a.y, a.y_pre = _p(x), _p(w), _p(bias), _p(res), _p(y), _p(y_pre)
    a.widx, a.stats, a.partial = _p(widx), _p(stats), None
    a.bn_x, a.bn_coef = _p(bn_x), _p(bn_coef)
    glu = bool(flags & L.EPI_GLU)        # y (and y_pre = the gate) are half as wide as the conv's output channels
    Cout_p = 2 * y.shape[1] if glu else y.shape[1]
    if glu_bwd is not None:              # (out, gate) of the GLU this conv's output is the gradient of: y = [d value | d gate]
        flags |= L.EPI_GLU_BWD
        Cout_p = y.shape[1] // 2
        if glu_bwd[0].shape[1] != Cout_p or glu_bwd[1].shape != glu_bwd[0].shape or stats is None:
            raise L.SdaError("conv_gemm: glu_bwd needs (out, gate) of the conv's width and a stats buffer")
        a.glu_out, a.glu_gate = _p(glu_bwd[0]), _p(glu_bwd[1])
    a.B, a.T, a.Cin_p, a.Cout_p, a.KS, a.dil = B, T, x.shape[1], Cout_p, KS, dil
    if glu and (res is not None or stats is not None or bn_x is not None or (y_pre is not None and y_pre.shape != y.shape)):
        raise L.SdaError("conv_gemm: EPI_GLU takes no residual / statistics and a gate buffer of y's shape")
    if bn_x is not None and gelu_bwd_u is None and (bn_x.shape != y.shape or bn_coef is None or bn_coef.numel() != 4 * y.shape[1] or stats is None):
        raise L.SdaError("conv_gemm: bn_x needs the shape of y, a [4][Cout_p] coefficient table and a stats buffer")
    if w.shape[-1] != x.shape[1] or w.shape[-2] != Cout_p or w.shape[-3] != KS:
        raise L.SdaError(f"conv_gemm: weight {tuple(w.shape)} does not match x {tuple(x.shape)} / y {tuple(y.shape)}")
    a.x_pitch, a.w_pitch = x.shape[1], w.shape[-1]
    a.x_row0, a.x_sample_rows, a.x_rows_limit = L.ROW_PAD, L.rows_tp(T), x.shape[0]
    if x.shape[0] < L.rows_alloc(B, T) or y.shape[0] < L.rows_alloc(B, T):
        raise L.SdaError("conv_gemm: RL buffers too small for (B, T)")
    a.w_rows_limit, a.ksplit = Cout_p, 1
    a.flags, a.dtype = (L.EPI_GELU if gelu else 0) | flags, dt_code(x.dtype)
    if TIMER is not None and "conv_gemm" in TIMER.families:
        cin, cout = alg_dims if alg_dims is not None else (x.shape[1], y.shape[1])
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        L.check(L.load().sda_conv_gemm(C.byref(a), _st()), "conv_gemm")
        e1.record()
        TIMER.records.append((("conv_gemm", str(x.dtype).replace("torch.", ""), conv_tile_co(Cout_p, KS, stats is not None), KS),
                              2.0 * B * T * KS * cin * cout, e0, e1))
        return y
    L.check(L.load().sda_conv_gemm(C.byref(a), _st()), "conv_gemm")
    return y


def n_t_tiles(T: int) -> int:
    return (T + 127) // 128


def conv_stats_rows(B: int, T: int, KS: int, Cout_p: int, flags: int = 0) -> int:
    """Rows of the per-tile statistics buffer a conv_gemm launch with these parameters writes."""
    return L.load().sda_conv_stats_rows(B, T, KS, Cout_p, flags)


# the similarity matmul of 16-bit operands on sim_gemm.hip's 256 x 256 tiles (SDA_SIM_GEMM=0: conv_gemm's split-K matrix mode
# with 128 x 128 tiles, which fp32 storage always uses)
SIM_GEMM_TILES256 = os.environ.get("SDA_SIM_GEMM", "1") != "0"


def matmul_nt_splitk(xm: torch.Tensor, wm: torch.Tensor, M: int, N: int, K: int, pitch: int) -> torch.Tensor:
    """S[i][j] = sum_k xm[i][k] * wm[j][k] (both K-contiguous rows with `pitch`), fp32 (M, pad64(N)) result.
    Runs conv_gemm in split-K mode + ordered slab reduction (loss.py:68)."""
    Np = L.pad_channels(N)
    if SIM_GEMM_TILES256 and K <= pitch:
        # 16-bit storage: 256 x 256 output tiles, every operand byte of a K slice through LDS once (csrc/sim_gemm.hip)
        ks = L.load().sda_sim_gemm_ksplit(M, N, K, dt_code(xm.dtype))
        if ks > 0:
            partial = torch.empty((ks, M, Np), dtype=torch.float32, device=xm.device)
            L.check(L.load().sda_sim_gemm(_p(xm), _p(wm), _p(partial), M, N, Np, K, pitch, ks, dt_code(xm.dtype), _st()), "sim_gemm")
            if ks == 1:
                return partial[0]
            out = torch.empty((M, Np), dtype=torch.float32, device=xm.device)
            L.check(L.load().sda_reduce_slabs(_p(partial), _p(out), ks, M * Np, _st()), "reduce_slabs")
            return out
    slab = 32 if xm.dtype == torch.float32 else 64
    nslab = K // slab
    tile_co = 160 if Np % 160 == 0 else (128 if Np % 128 == 0 else 64)
    tiles = ((M + 127) // 128) * (Np // tile_co)
    ksplit = max(1, min(nslab, (512 + tiles - 1) // tiles))
    if nslab <= 64:                 # a short contraction (SpatialAttention's weights: 2048 deep, 270 x 256 outputs): the slab sum
        ksplit = min(ksplit, 16)    # has few threads, each walking every slab — 64 slabs cost 30 us on the step's start chain
    partial = torch.empty((ksplit, M, Np), dtype=torch.float32, device=xm.device)
    a = L.ConvArgs()
    a.x, a.w, a.bias, a.res, a.y, a.y_pre, a.widx, a.stats = _p(xm), _p(wm), None, None, None, None, None, None
    a.partial = _p(partial)
    a.B, a.T, a.Cin_p, a.Cout_p, a.KS, a.dil = 1, M, K, Np, 1, 0
    a.x_pitch, a.w_pitch, a.x_row0, a.x_sample_rows, a.x_rows_limit = pitch, pitch, 0, 0, M
    a.w_rows_limit, a.ksplit, a.flags, a.dtype = N, ksplit, 0, dt_code(xm.dtype)
    L.check(L.load().sda_conv_gemm(C.byref(a), _st()), "conv_gemm(split-K)")
    if ksplit == 1:
        return partial[0]
    out = torch.empty((M, Np), dtype=torch.float32, device=xm.device)
    L.check(L.load().sda_reduce_slabs(_p(partial), _p(out), ksplit, M * Np, _st()), "reduce_slabs")
    return out


def bn_finalize(partial, ntiles, count, gamma, beta, running_mean, running_var, Cp, training, eps=1e-5, momentum=0.1,
                want_bwd_coef=False, batches_tracked=None):
    """Returns (mean, rstd, scale, shift[, bwd_coef]); bwd_coef = [4][Cp] (gamma, beta, mean, rstd), the table the
    data-gradient conv reads in its BatchNorm-backward statistics mode.  batches_tracked: the module's int64
    num_batches_tracked buffer, incremented by the same launch in training mode."""
    if batches_tracked is not None and (batches_tracked.dtype != torch.int64 or not batches_tracked.is_cuda):
        raise L.SdaError("bn_finalize: batches_tracked must be an int64 device tensor")
    dev = gamma.device
    buf = torch.empty((8 if want_bwd_coef else 4, Cp), dtype=torch.float32, device=dev)
    mean, rstd, scale, shift = buf[0], buf[1], buf[2], buf[3]
    coef = buf[4:] if want_bwd_coef else None
    L.check(L.load().sda_bn_finalize(_p(partial), ntiles, float(count), _p(gamma), _p(beta), eps, momentum,
                                     _p(running_mean), _p(running_var), _p(mean), _p(rstd), _p(scale), _p(shift),
                                     _p(coef), gamma.numel(), Cp, int(training), _p(batches_tracked), _st()), "bn_finalize")
    return (mean, rstd, scale, shift, coef) if want_bwd_coef else (mean, rstd, scale, shift)


def bn_gelu_forward(x, y, scale, shift, B, T):
    L.check(L.load().sda_bn_gelu_forward(_p(x), _p(y), _p(scale), _p(shift), B, T, x.shape[1], dt_code(x.dtype), _st()),
            "bn_gelu_forward")
    return y


def reduce_scratch(Cp, device):
    return torch.empty(L.load().sda_reduce_scratch_floats(Cp), dtype=torch.float32, device=device)


def bn_gelu_backward(dy, x, mean, rstd, gamma, beta, dx, B, T, scratch, count=None, allreduce=None, tile_stats=None, dy_is_dg=False):
    """Returns (dgamma, dbeta) summed over `count` rows (global sums when `allreduce` is given) and fills dx.
    tile_stats: per-tile (sum dg, sum dg*xhat) already produced by the conv that wrote dy (conv_gemm(bn_x=...));
    without it the sums take a pass of their own over dy and x."""
    Cp = x.shape[1]
    sums = torch.empty((2, Cp), dtype=torch.float32, device=x.device)
    dgamma, dbeta = sums[0], sums[1]
    lib = L.load()
    if tile_stats is not None and allreduce is None:          # single process: sums + coefficients in one launch
        coef = torch.empty(6 * Cp, dtype=torch.float32, device=x.device)
        fn = lib.sda_bn_gelu_backward_from_stats_dg if dy_is_dg else lib.sda_bn_gelu_backward_from_stats
        L.check(fn(_p(tile_stats), tile_stats.shape[0], _p(dy), _p(x), _p(mean), _p(rstd),
                                                    _p(gamma), _p(beta), gamma.numel(), float(count if count is not None else B * T),
                                                    _p(dgamma), _p(dbeta), _p(coef), _p(dx), B, T, Cp, dt_code(x.dtype), _st()),
                "bn_gelu_backward_from_stats")
        return dgamma, dbeta
    if tile_stats is not None:
        L.check(lib.sda_reduce_stats(_p(tile_stats), tile_stats.shape[0], _p(dbeta), _p(dgamma), Cp, _st()), "reduce_stats")
    else:
        L.check(lib.sda_bn_gelu_backward_reduce(_p(dy), _p(x), _p(mean), _p(rstd), _p(gamma), _p(beta), gamma.numel(), _p(scratch),
                                                _p(dgamma), _p(dbeta), B, T, Cp, dt_code(x.dtype), _st()), "bn_gelu_backward_reduce")
    if allreduce is not None:
        allreduce(sums)
    coef = torch.empty(6 * Cp, dtype=torch.float32, device=x.device)
    if dy_is_dg and tile_stats is None:
        raise L.SdaError("bn_gelu_backward: dy_is_dg needs the statistics rows of the conv that wrote dg")
    fn = lib.sda_bn_gelu_backward_apply_dg if dy_is_dg else lib.sda_bn_gelu_backward_apply
    L.check(fn(_p(dy), _p(x), _p(mean), _p(rstd), _p(gamma), _p(beta), gamma.numel(), _p(dgamma),
                                           _p(dbeta), float(count if count is not None else B * T), _p(coef), _p(dx), B, T, Cp,
                                           dt_code(x.dtype), _st()), "bn_gelu_backward_apply")
    return dgamma, dbeta


def reduce_stats(stats: torch.Tensor) -> torch.Tensor:
    """Per-tile rows (n, 2, Cp) -> fp32 (2 * Cp,) = [column sums of slot 0 | of slot 1], summed in fixed order."""
    Cp = stats.shape[2]
    out = torch.empty(2 * Cp, dtype=torch.float32, device=stats.device)
    L.check(L.load().sda_reduce_stats(_p(stats), stats.shape[0], _p(out), _p(out[Cp:]), Cp, _st()), "reduce_stats")
    return out


def glu_forward(x, y, B, T):
    L.check(L.load().sda_glu_forward(_p(x), _p(y), B, T, y.shape[1], dt_code(x.dtype), _st()), "glu_forward")
    return y


def glu_backward(x, dy, dx, B, T):
    L.check(L.load().sda_glu_backward(_p(x), _p(dy), _p(dx), B, T, dy.shape[1], dt_code(x.dtype), _st()), "glu_backward")
    return dx


def gelu_backward(u, dz, du, B, T):
    L.check(L.load().sda_gelu_backward(_p(u), _p(dz), _p(du), B, T, u.shape[1], dt_code(u.dtype), _st()), "gelu_backward")
    return du


def glu_backward_colsum(x, dy, dx, B, T, scratch):
    """dx = GLU backward; returns the fp32 column sums of dx (2*Ch) = bias gradient of the conv that made x."""
    cs = torch.empty(2 * dy.shape[1], dtype=torch.float32, device=x.device)
    L.check(L.load().sda_glu_backward_colsum(_p(x), _p(dy), _p(dx), _p(cs), _p(scratch), B, T, dy.shape[1], dt_code(x.dtype), _st()),
            "glu_backward_colsum")
    return cs


def _deferred_colsum(scratch, B, T, width, two):
    """finish() for a column-sum kernel launched without its final reduction: sums the partial rows left in `scratch`
    on whatever stream is current when it is called (the caller orders that stream after the producing launch)."""
    nb = L.load().sda_reduce_scratch_rows(B, T)

    def finish():
        scratch.record_stream(torch.cuda.current_stream(scratch.device))     # allocated on the producer's stream, read on this one
        cs = torch.empty((2 if two else 1) * width, dtype=torch.float32, device=scratch.device)
        L.check(L.load().sda_reduce_stats(_p(scratch), nb, _p(cs), _p(cs[width:]) if two else None, width, _st()), "reduce_stats")
        return cs
    return finish


def glu_backward_colsum_og(out, gate, dy, dx, B, T, scratch, defer=False):
    """GLU backward after a fused forward (conv_gemm(flags=EPI_GLU)): `out` = value * sigmoid(gate) and `gate` instead of the
    [value | gate] buffer; same dx ([d value | d gate]) and column sums as glu_backward_colsum.  defer=True: returns finish()
    instead of the sums (the final reduction of the per-workgroup partial rows is then the caller's to place; `scratch` must
    be a buffer of this call's own)."""
    Ch = dy.shape[1]
    cs = None if defer else torch.empty(2 * Ch, dtype=torch.float32, device=dy.device)
    L.check(L.load().sda_glu_backward_colsum_og(_p(out), _p(gate), _p(dy), _p(dx), _p(cs), _p(scratch), B, T, Ch,
                                                dt_code(dy.dtype), _st()), "glu_backward_colsum_og")
    return _deferred_colsum(scratch, B, T, Ch, True) if defer else cs


def gelu_backward_colsum(u, dz, du, B, T, scratch, defer=False):
    cs = None if defer else torch.empty(u.shape[1], dtype=torch.float32, device=u.device)
    L.check(L.load().sda_gelu_backward_colsum(_p(u), _p(dz), _p(du), _p(cs), _p(scratch), B, T, u.shape[1], dt_code(u.dtype), _st()),
            "gelu_backward_colsum")
    return _deferred_colsum(scratch, B, T, u.shape[1], False) if defer else cs


def colsum(x, B, T, scratch):
    out = torch.empty(x.shape[1], dtype=torch.float32, device=x.device)
    L.check(L.load().sda_colsum(_p(x), _p(out), _p(scratch), B, T, x.shape[1], dt_code(x.dtype), _st()), "colsum")
    return out


def wgrad_gemm(dy, x, *, B, T, KS, dil, perm=None, seg_start=None, nseg=1, alg_dims=None, flat_rows=False):
    """fp32 slabs (nseg, KS, Cout_p, Cin_p) of dy^T x over the RL rows of the samples in each segment.
    alg_dims = (Cin, Cout) unpadded, only used to count algorithmic FLOPs when the timer is on.
    flat_rows: the caller guarantees that dy's pad rows (the 16 rows in front of every sample) are zero — true for every
    row-layout buffer the kernels of this library write (they only ever write valid rows of zero-initialised buffers) —
    so a segment of consecutive samples is contracted as one run of rows in whole K-chunks (SDA_WGRAD_FLAT_ROWS)."""
    a = L.WgradArgs()
    g = torch.empty((nseg, KS, dy.shape[1], x.shape[1]), dtype=torch.float32, device=x.device)
    a.dy, a.x, a.g, a.out_e, a.sub, a.rscale, a.out_scale = _p(dy), _p(x), _p(g), None, None, None, None
    a.perm, a.seg_start = _p(perm), _p(seg_start)
    a.nseg, a.B, a.T, a.Cout_p, a.Cin_p, a.KS, a.dil = nseg, B, T, dy.shape[1], x.shape[1], KS, dil
    a.dy_pitch, a.x_pitch, a.out_pitch = dy.shape[1], x.shape[1], 0
    a.row0, a.sample_rows, a.rows_limit, a.dy_zero_row = L.ROW_PAD, L.rows_tp(T), x.shape[0], 0
    a.co_valid, a.dtype, a.acc_scale = 0, dt_code(x.dtype), None
    a.flags = L.WGRAD_FLAT_ROWS if flat_rows else 0      # (with `perm`: every sample as whole chunks across its own padding)
    if nseg > 1 and seg_start is None:
        raise L.SdaError("wgrad_gemm: nseg > 1 needs seg_start")
    if TIMER is not None and "wgrad_gemm" in TIMER.families:   # events go on the CURRENT stream (the side stream in backward)
        cin, cout = alg_dims if alg_dims is not None else (x.shape[1], dy.shape[1])
        tile_m = 160 if dy.shape[1] % 160 == 0 else (128 if dy.shape[1] % 128 == 0 else 64)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        L.check(L.load().sda_wgrad_gemm(C.byref(a), _st()), "wgrad_gemm")
        e1.record()
        TIMER.records.append((("wgrad_gemm", str(x.dtype).replace("torch.", ""), tile_m, KS),
                              2.0 * B * T * KS * cin * cout, e0, e1))
        return g
    L.check(L.load().sda_wgrad_gemm(C.byref(a), _st()), "wgrad_gemm")
    return g


def reduce_slabs(slabs: torch.Tensor) -> torch.Tensor:
    n = slabs[0].numel()
    if slabs.shape[0] == 1:
        return slabs[0]
    out = torch.empty_like(slabs[0])
    L.check(L.load().sda_reduce_slabs(_p(slabs), _p(out), slabs.shape[0], n, _st()), "reduce_slabs")
    return out


def matmul_tn_typed(G, Ym, out, sub, rscale, *, M_rows, N_valid, K_cols, pitch, out_scale=None, acc_scale=None):
    """out[j][k] = out_scale * (acc_scale[j] * sum_i G[i][j] * Ym[i][k] - rscale[j] * sub[j][k])   (typed rows with `pitch`)."""
    a = L.WgradArgs()
    a.acc_scale = _p(acc_scale)
    a.dy, a.x, a.g, a.out_e, a.sub, a.rscale, a.out_scale = _p(G), _p(Ym), None, _p(out), _p(sub), _p(rscale), _p(out_scale)
    a.perm, a.seg_start = None, None
    a.nseg, a.B, a.T, a.Cout_p, a.Cin_p, a.KS, a.dil = 1, 1, M_rows, G.shape[1], K_cols, 1, 0
    a.dy_pitch, a.x_pitch, a.out_pitch = G.shape[1], pitch, pitch
    if G.shape[0] < M_rows + 1:
        raise L.SdaError("matmul_tn_typed: G needs one trailing all-zero row")
    a.row0, a.sample_rows, a.rows_limit, a.dy_zero_row = 0, 0, M_rows, M_rows
    a.co_valid, a.dtype = N_valid, dt_code(Ym.dtype)
    L.check(L.load().sda_wgrad_gemm(C.byref(a), _st()), "wgrad_gemm(typed)")
    return out


def clip_dz(G, Yt, Zt, out, rscale, cscale, *, Bm, Bn, row_elems, out_scale=None):
    """out[j][k] = out_scale * (cscale[j] * sum_i G[i][j] Yt[i][k] - rscale[j] Zt[j][k]) — the loss's embedding gradient.
    One GPU's shapes (Bm <= 256, 16-bit) run on the streaming kernel (loss_gemm.hip); anything else on wgrad_gemm's typed output."""
    if L.load().sda_clip_dz_supported(Bm, Bn, row_elems, dt_code(Yt.dtype)):
        L.check(L.load().sda_clip_dz(_p(G), G.shape[1], _p(Yt), _p(Zt), _p(out), _p(cscale), _p(rscale), _p(out_scale), Bm, Bn, row_elems,
                                     dt_code(Yt.dtype), _st()), "clip_dz")
        return out
    return matmul_tn_typed(G, Yt, out, Zt, rscale, M_rows=Bm, N_valid=Bn, K_cols=row_elems, pitch=row_elems, out_scale=out_scale,
                           acc_scale=cscale)


def sa_gemm_tables(cos_t: torch.Tensor, sin_t: torch.Tensor):
    """Operand tables of SpatialAttention's two contractions, built from the (K2, C) cos/sin buffers:
    forward  a[o][c] = sum_m Re z[o][m] cos[m][c] + Im z[o][m] sin[m][c]  ->  rows c of [cos | sin] interleaved in m,
    backward dz[o][m] = sum_c da[o][c] (cos[m][c], sin[m][c])            ->  rows (2m, 2m+1), sensors padded to 64.
    The owner of the buffers (models.SpatialAttention) caches them and rebuilds when the buffers change."""
    K2, Cc = cos_t.shape
    both = torch.stack([cos_t, sin_t], dim=-1)                                    # (K2, C, 2)
    fwd = both.permute(1, 0, 2).reshape(Cc, 2 * K2).contiguous()
    Cq = (Cc + 63) // 64 * 64
    bwd = torch.zeros((2 * K2, Cq), dtype=torch.float32, device=cos_t.device)
    bwd[:, :Cc] = both.permute(0, 2, 1).reshape(2 * K2, Cc)
    return fwd, bwd


def sa_weights_forward(z, cos_t, sin_t, mask, D1p, Cp, dtype, fwd_table=None):
    """SpatialAttention weights (models.py:49-58) + dropout mask (81-84): W fp32 (D1, C) and the packed operand.
    The (D1 x 2K2) . (2K2 x C) contraction runs on the fp32 matrix path (conv_gemm, split-K matrix mode)."""
    D1, K2 = z.shape
    Cc = cos_t.shape[1]
    zr = torch.view_as_real(z).contiguous().view(D1, 2 * K2)
    W = torch.empty((D1, Cc), dtype=torch.float32, device=z.device)
    Wp = torch.empty((1, 1, D1p, Cp), dtype=dtype, device=z.device)
    if fwd_table is not None and (2 * K2) % 64 == 0:   # the matrix path contracts in whole 64-element slabs
        a = matmul_nt_splitk(zr, fwd_table, D1, Cc, 2 * K2, 2 * K2)                          # (D1, pad64(C)) fp32
        L.check(L.load().sda_sa_softmax_pack(_p(a), a.shape[1], _p(mask), _p(W), _p(Wp), D1, Cc, D1p, Cp, dt_code(dtype), _st()),
                "sa_softmax_pack")
        return W, Wp
    scratch = torch.empty(L.load().sda_sa_scratch_floats(D1, K2, Cc), dtype=torch.float32, device=z.device)
    L.check(L.load().sda_sa_weights_forward(_p(zr), _p(cos_t), _p(sin_t), _p(mask), _p(W), _p(Wp), _p(scratch), D1, K2, Cc, D1p, Cp,
                                            dt_code(dtype), _st()), "sa_weights_forward")
    return W, Wp


def sa_weights_backward(dWd, W, mask, cosT, sinT, K2, bwd_table=None):
    """dWd fp32 (rows >= D1, pitch dWd.shape[-1]) -> dz complex64 (D1, K2).  With `bwd_table` (sa_gemm_tables) the
    (D1 x C) . (C x 2K2) contraction runs on the fp32 matrix path; otherwise the stand-alone kernel is used."""
    D1, Cc = W.shape
    if bwd_table is not None and (2 * K2) % 64 == 0:
        bwd = bwd_table
        Cq = bwd.shape[1]
        da = torch.empty((D1, Cq), dtype=torch.float32, device=W.device)
        L.check(L.load().sda_sa_softmax_backward(_p(dWd), dWd.shape[-1], _p(W), _p(mask), _p(da), Cq, D1, Cc, _st()),
                "sa_softmax_backward")
        dz = matmul_nt_splitk(da, bwd, D1, 2 * K2, Cq, Cq)                              # (D1, 2*K2) = (re, im) interleaved
        return torch.view_as_complex(dz[:, : 2 * K2].reshape(D1, K2, 2))
    dz = torch.empty((D1, K2, 2), dtype=torch.float32, device=W.device)
    L.check(L.load().sda_sa_weights_backward(_p(dWd), _p(W), _p(mask), _p(cosT), _p(sinT), _p(dz), D1, K2, Cc, dWd.shape[-1],
                                             _st()), "sa_weights_backward")
    return torch.view_as_complex(dz)


def clip_logits_stats(S, ysq, zsq, temp, Bm, Bn, col0):
    dev = S.device
    logits = torch.empty((Bm, Bn), dtype=torch.float32, device=dev)
    # (row max, row sum, positive's logit) as the three rows of ONE buffer: under data parallelism that buffer is what the
    # all-gather of the row statistics sends — no stack / copy in front of the collective
    st3 = torch.empty((3, Bm), dtype=torch.float32, device=dev)
    row_max, row_sum, diag = st3[0], st3[1], st3[2]                    # (diag: every row is written, zero where the positive lives elsewhere)
    col_lse = torch.empty(Bn, dtype=torch.float32, device=dev)
    row_lse = torch.empty(Bm, dtype=torch.float32, device=dev)        # lse over THIS block of columns
    L.check(L.load().sda_clip_logits_stats(_p(S), S.shape[1], _p(ysq), _p(zsq), _p(temp), _p(logits), _p(row_max), _p(row_sum),
                                           _p(col_lse), _p(diag), _p(row_lse), Bm, Bn, col0, _st()), "clip_logits_stats")
    return logits, row_max, row_sum, col_lse, diag, row_lse


def clip_grad(logits, row_lse, col_lse, ysq, zsq, temp, inv_norm, col0, dtype):
    Bm, Bn = logits.shape
    dev = logits.device
    G = torch.empty((Bm + 1, L.pad_channels(Bn)), dtype=dtype, device=dev)      # + one zero row (wgrad_gemm's t >= T stand-in):
                                                                                 # it and the padding columns are written by the kernel
    rscale = torch.empty(Bn, dtype=torch.float32, device=dev)
    cscale = torch.empty(Bn, dtype=torch.float32, device=dev)
    colpart = torch.empty(2 * Bn, dtype=torch.float32, device=dev)
    scalars = torch.empty(2, dtype=torch.float32, device=dev)
    L.check(L.load().sda_clip_grad(_p(logits), _p(row_lse), _p(col_lse), _p(ysq), _p(zsq), _p(temp), float(inv_norm), col0,
                                   _p(G), G.shape[1], _p(rscale), _p(cscale), _p(colpart), _p(scalars), Bm, Bn, dt_code(dtype), _st()),
            "clip_grad")
    return G, rscale, cscale, scalars


def clip_ranks(logits, diag, col0):
    Bm, Bn = logits.shape
    cnt = torch.empty(Bm, dtype=torch.int32, device=logits.device)
    L.check(L.load().sda_clip_ranks(_p(logits), _p(diag), _p(cnt), Bm, Bn, col0, _st()), "clip_ranks")
    return cnt


# ---------------------------------------------------------------------------------------------------------------
# wav2vec 2.0 embedder stages (csrc/w2v2.hip) and the raw GEMM form of conv_gemm they share
# ---------------------------------------------------------------------------------------------------------------
def gemm_view(x_ptr: int, w_ptr: int, y_ptr: int, *, rows: int, K: int, Cout_p: int, x_pitch: int, w_pitch: int, x_row0: int,
              x_rows_limit: int, dtype, bias=None, res_ptr: Optional[int] = None, gelu: bool = False,
              w_rows_limit: Optional[int] = None, batch: int = 1, sample_rows: Optional[int] = None, widx=None):
    """y[x_row0 + r][0:Cout_p] = sum_k x[x_row0 + r][k] * w[co][k] (+ bias)(+ res)(GELU) for r < rows, on raw device
    addresses: conv_gemm with kernel size 1 in matrix mode.  Row r of x starts at x_ptr + (x_row0 + r) * x_pitch elements
    and is K elements long, so x_pitch < K gives overlapping rows (an im2col view of a strided Conv1d); w row co starts at
    w_ptr + co * w_pitch.  y rows have Cout_p elements.  The caller guarantees every address touched is allocated."""
    a = L.ConvArgs()
    a.x, a.w, a.bias, a.res, a.y, a.y_pre = x_ptr, w_ptr, _p(bias), res_ptr, y_ptr, None
    a.widx, a.stats, a.partial, a.bn_x, a.bn_coef = _p(widx), None, None, None, None
    a.B, a.T, a.Cin_p, a.Cout_p, a.KS, a.dil = batch, rows, K, Cout_p, 1, 0
    a.x_pitch, a.w_pitch = x_pitch, w_pitch
    # `batch` independent problems of `rows` rows each, `sample_rows` view rows apart (x and y alike); widx[b] picks the
    # b-th problem's weight matrix out of w [nW][Cout_p][w_pitch]
    a.x_row0, a.x_sample_rows, a.x_rows_limit = x_row0, (rows + L.ROW_PAD if sample_rows is None else sample_rows), x_rows_limit
    a.w_rows_limit, a.ksplit = (Cout_p if w_rows_limit is None else w_rows_limit), 1
    a.flags, a.dtype = (L.EPI_GELU if gelu else 0), dt_code(dtype)
    L.check(L.load().sda_conv_gemm(C.byref(a), _st()), "conv_gemm(view)")


def w2v_conv0(wave, w, bias, gamma, beta, y, T, C, K, stride, eps=1e-5):
    L.check(L.load().sda_w2v_conv0(_p(wave), wave.numel(), _p(w), _p(bias), _p(gamma), _p(beta), _p(y), T, C, y.shape[1], K, stride,
                                   eps, dt_code(y.dtype), _st()), "w2v_conv0")
    return y


def layernorm_rows(x, y, gamma, beta, T, C, eps=1e-5, gelu=False):
    L.check(L.load().sda_layernorm_rows(_p(x), _p(y), _p(gamma), _p(beta), T, C, x.shape[1], eps, int(gelu), dt_code(x.dtype), _st()),
            "layernorm_rows")
    return y


def w2v_group_split(h, xg, T, gw, G, lead):
    L.check(L.load().sda_w2v_group_split(_p(h), _p(xg), T, h.shape[1], gw, xg.shape[2], G, xg.shape[1], lead, dt_code(h.dtype), _st()),
            "w2v_group_split")


def w2v_group_merge_add(h, yg, out, T, gw, G, bias=None, gelu=False):
    L.check(L.load().sda_w2v_group_merge_add(_p(h), _p(yg), _p(out), T, h.shape[1], gw, yg.shape[2], G, yg.shape[1],
                                             _p(bias), int(gelu), dt_code(h.dtype), _st()), "w2v_group_merge_add")
    return out


def w2v_attention(q_ptr, k_ptr, vt, out, T, heads, head_dim, qk_pitch, scale):
    L.check(L.load().sda_w2v_attention(q_ptr, k_ptr, _p(vt), _p(out), T, heads, head_dim, qk_pitch, vt.shape[1], out.shape[1],
                                       scale, dt_code(out.dtype), _st()), "w2v_attention")
    return out


def w2v_mean4(a, b, c, d, T, Cc):
    out = torch.empty((T, Cc), dtype=torch.float32, device=a.device)
    L.check(L.load().sda_w2v_mean4(_p(a), _p(b), _p(c), _p(d), _p(out), T, Cc, a.shape[1], dt_code(a.dtype), _st()), "w2v_mean4")
    return out


def linear_rows(x, w, y, T, *, bias=None, res=None, gelu=False, scratch=None):
    """y = f(x W^T + bias) + res on ONE row-layout sample of T frames (w packed (1, 1, Cout_p, Cin_p)).  Few frames make
    few output tiles (a 1024-wide layer at T = 299 is 24 workgroups): then the contraction is split over workgroups
    (sda_conv_gemm's split-K slabs) and a small epilogue kernel sums the slabs and applies bias / GELU / residual."""
    Cout_p, Cin_p = w.shape[-2], w.shape[-1]
    tiles = ((T + 127) // 128) * (Cout_p // conv_tile_co(Cout_p, 1))
    nslab = Cin_p // (32 if x.dtype == torch.float32 else 64)
    ksplit = min(nslab, max(1, 256 // tiles))
    if ksplit < 2:
        return conv_gemm(x, w, y, B=1, T=T, KS=1, dil=0, bias=bias, res=res, gelu=gelu)
    need = ksplit * T * Cout_p
    if scratch is None or scratch.numel() < need:
        scratch = torch.empty(need, dtype=torch.float32, device=x.device)
    a = L.ConvArgs()
    a.x, a.w, a.bias, a.res, a.y, a.y_pre, a.widx, a.stats = _p(x), _p(w), None, None, None, None, None, None
    a.partial, a.bn_x, a.bn_coef = _p(scratch), None, None
    a.B, a.T, a.Cin_p, a.Cout_p, a.KS, a.dil = 1, T, Cin_p, Cout_p, 1, 0
    a.x_pitch, a.w_pitch = x.shape[1], Cin_p
    a.x_row0, a.x_sample_rows, a.x_rows_limit = L.ROW_PAD, L.rows_tp(T), x.shape[0]
    a.w_rows_limit, a.ksplit, a.flags, a.dtype = Cout_p, ksplit, 0, dt_code(x.dtype)
    L.check(L.load().sda_conv_gemm(C.byref(a), _st()), "conv_gemm(split-K)")
    L.check(L.load().sda_splitk_epilogue(_p(scratch), ksplit, _p(bias), _p(res), _p(y), T, Cout_p, int(gelu), dt_code(x.dtype), _st()),
            "splitk_epilogue")
    return y
