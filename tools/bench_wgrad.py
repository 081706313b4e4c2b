"""Micro-benchmark of wgrad_gemm alone at the config-2 shapes of the training step (diagnostic): the kernel-3 weight gradients,
the two 1x1 ones (conv_final1 / conv_final2) and the loss's dZ product.  SDA_WGRAD_PF=<n> sets the L2-prefetch distance."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from speech_decoding_amd import ops, engine as E, lib as L

def timeit(fn, n=20, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3   # us

dev = "cuda:0"
B, T = 256, 360
dt = torch.float32 if os.environ.get("DTYPE") == "fp32" else torch.bfloat16
print("SDA_WGRAD_PF =", os.environ.get("SDA_WGRAD_PF", "(default)"))
for (cin, cout, KS, dil, wgs) in [(320, 320, 3, 4, 256), (320, 320, 3, 4, 512), (320, 640, 3, 2, 256), (320, 640, 3, 2, 512),
                                  (640, 1024, 1, 0, 256), (640, 1024, 1, 0, 512), (320, 640, 1, 0, 256), (256, 320, 3, 1, 256)]:
    x = ops.new_rows(B, T, cin, dt, dev); x.normal_()
    dy = ops.new_rows(B, T, cout, dt, dev); dy.normal_()
    tile_m = 160 if cout % 160 == 0 else (128 if cout % 128 == 0 else 64)
    tn = 64 if KS == 3 else (128 if cin % 128 == 0 else 64)
    ntiles = (cout // tile_m) * (cin // tn)
    nseg = 8 * max(1, round(wgs / (8 * ntiles)))
    seg = torch.from_numpy(np.floor(np.linspace(0, B, nseg + 1)).astype(np.int32)).to(dev)
    fl = 2.0 * B * T * KS * cin * cout
    us = timeit(lambda: ops.wgrad_gemm(dy, x, B=B, T=T, KS=KS, dil=dil, perm=None, seg_start=seg, nseg=nseg, flat_rows=True))
    print(f"wgrad {cin:4d}->{cout:4d} k{KS} tiles {ntiles:3d} nseg {nseg:3d} ({ntiles * nseg:4d} wgs) {us:8.1f} us  {fl / us / 1e6:7.1f} TF", flush=True)
F = 1024
if dt == torch.float32: sys.exit(0)
for Bm, Bn in [(256, 256), (2048, 256)]:
    Yt = ops.new_rows(Bm, T, F, dt, dev); Zt = ops.new_rows(Bn, T, F, dt, dev)
    ops.rows_view(Yt, Bm, F, T).normal_(); ops.rows_view(Zt, Bn, F, T).normal_()
    temp = torch.tensor([5.1], device=dev)
    re = L.rows_tp(T) * F
    loss, logits, cnt, ctx = E.clip_forward(Yt, Zt, temp, Bm=Bm, Bn=Bn, T=T)
    dZ = ops.new_rows(Bn, T, F, dt, dev)
    t = timeit(lambda: E.clip_backward(ctx, dZ))
    print(f"dZ gemm Bm={Bm} Bn={Bn}: {t:8.1f} us  ({2.0 * Bm * Bn * re / t / 1e6:.0f} TF, {(Bm + 2 * Bn) * re * 2 / t / 1e6:.2f} TB/s algorithmic)", flush=True)
    t = timeit(lambda: ops.matmul_nt_splitk(Yt, Zt, Bm, Bn, re, re))
    print(f"similarity Bm={Bm} Bn={Bn}: {t:8.1f} us  ({2.0 * Bm * Bn * re / t / 1e6:.0f} TF, {(Bm + Bn) * re * 2 / t / 1e6:.2f} TB/s algorithmic)", flush=True)
