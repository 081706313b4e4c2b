"""Which pairs of the backward's kernels gain from running side by side?  (diagnostic)
Each kernel of a ConvBlock's backward at config-2 shapes alone, then pairs on two streams, both streams kept busy for N launches:
wall time per (one launch of each) against the sum and the maximum of the two alone times.
    W  = wgrad_gemm 320 -> 320 k3 (one workgroup per CU)     W2 = the 640-channel one
    Dt = data-gradient conv on the tile kernel, BatchNorm-backward sums in its epilogue (what the step runs)
    Df = the same on flat tiles (two workgroups per CU); Df1 = one workgroup per CU
    P  = bn_gelu_backward apply from the conv's statistics rows (HBM-bound pass)
    C  = glu_backward_colsum (HBM-bound pass)
    U  = reduce_unpack_wgrad (slab sum of a weight gradient)"""
import sys, os, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
from speech_decoding_amd import ops, lib as L

dev, dt = "cuda:0", torch.bfloat16
B, T, C = 256, 360, 320
N = 10


def rows(c):
    t = ops.new_rows(B, T, c, dt, dev)
    ops.rows_view(t, B, c, T).normal_()
    return t


x, dy, dy2, h = rows(C), rows(C), rows(2 * C), rows(C)
w = torch.randn(C, C, 3, device=dev) / math.sqrt(3 * C)
wT = ops.pack_conv_weight(w.transpose(0, 1).flip(2).contiguous(), C, C, dt)          # data-gradient operand
w2 = torch.randn(2 * C, C, 3, device=dev) / math.sqrt(3 * C)
w2T = ops.pack_conv_weight(w2.transpose(0, 1).flip(2).contiguous(), C, 2 * C, dt)
mean, rstd = torch.zeros(C, device=dev), torch.ones(C, device=dev)
gamma, beta = torch.ones(C, device=dev), torch.zeros(C, device=dev)
coef = torch.cat([gamma, beta, mean, rstd]).contiguous()
scratch = ops.reduce_scratch(2 * C, dev)


def seg_for(ntiles, wgs=256):
    nseg = 8 * max(1, round(wgs / (8 * ntiles)))
    return torch.from_numpy(np.floor(np.linspace(0, B, nseg + 1)).astype(np.int32)).to(dev), nseg


def mk_w(cout):
    d = dy if cout == C else dy2
    seg, nseg = seg_for((cout // 160) * (C // 64))
    return lambda: ops.wgrad_gemm(d, x, B=B, T=T, KS=3, dil=4, perm=None, seg_start=seg, nseg=nseg, flat_rows=True)


def mk_d(flags, cin=C):
    src, wp = (dy, wT) if cin == C else (dy2, w2T)
    out = ops.new_rows(B, T, C, dt, dev)
    st = torch.zeros((ops.conv_stats_rows(B, T, 3, C, flags), 2, C), device=dev)
    return lambda: ops.conv_gemm(src, wp, out, B=B, T=T, KS=3, dil=4, stats=st, bn_x=h, bn_coef=coef, flags=flags), st, out


slabs = mk_w(C)()
d_t, st_t, g_t = mk_d(0)
d_f, _, _ = mk_d(L.CONV_FLAT_TILES)
d_f1, _, _ = mk_d(L.CONV_FLAT_TILES | L.CONV_ONE_PER_CU)
d2_t, _, _ = mk_d(0, 2 * C)
d2_f, _, _ = mk_d(L.CONV_FLAT_TILES, 2 * C)
d_t()
dxp = ops.new_rows(B, T, C, dt, dev)
p_ = lambda: ops.bn_gelu_backward(g_t, h, mean, rstd, gamma, beta, dxp, B, T, scratch, tile_stats=st_t)
gate, outv, dc2 = rows(C), rows(C), ops.new_rows(B, T, 2 * C, dt, dev)
c_ = lambda: ops.glu_backward_colsum_og(outv, gate, dy, dc2, B, T, scratch)
u_ = lambda: ops.reduce_unpack_wgrad(slabs, C, C, 3)
K = {"W": mk_w(C), "W2": mk_w(2 * C), "Dt": d_t, "Df": d_f, "Df1": d_f1, "D2t": d2_t, "D2f": d2_f, "P": p_, "C": c_, "U": u_}


def wall(pairs):
    for s, f in pairs:
        with torch.cuda.stream(s):
            f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    main = torch.cuda.current_stream()
    e0.record(main)
    for s, _ in pairs:
        s.wait_event(e0)
    for s, f in pairs:
        with torch.cuda.stream(s):
            for _ in range(N):
                f()
    for s, _ in pairs:
        ev = torch.cuda.Event()
        ev.record(s)
        main.wait_event(ev)
    e1.record(main)
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / N * 1e3


s1 = torch.cuda.Stream(priority=-1)
s2 = torch.cuda.Stream()
alone = {}
for rep in range(2):
    for k, f in K.items():
        alone[k] = wall([(s1, f)])
print("alone (us): " + "  ".join(f"{k} {v:.1f}" for k, v in alone.items()), flush=True)
for a, b in [("W", "P"), ("W", "C"), ("W", "U"), ("W", "Dt"), ("W", "Df"), ("W", "Df1"), ("W2", "D2t"), ("W2", "D2f"), ("Dt", "U"), ("Df", "U"),
             ("Df", "P"), ("W", "W")]:
    t = wall([(s1, K[b]), (s2, K[a])])
    print(f"{a:>3} || {b:<3}: {t:7.1f} us   (sum {alone[a] + alone[b]:6.1f}, max {max(alone[a], alone[b]):6.1f})", flush=True)
