"""Micro-benchmark of the loss stages at config-2 (Bm = 256) and config-3 (Bm = 2048, Bn = 256) shapes."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from speech_decoding_amd import ops, engine as E, lib as L

def timeit(fn, n=10, warm=2):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3

dev = "cuda:0"; T, F = 360, 1024
for Bm, Bn in [(256, 256), (2048, 256)]:
    dt = torch.bfloat16
    Yt = ops.new_rows(Bm, T, F, dt, dev); Zt = ops.new_rows(Bn, T, F, dt, dev)
    ops.rows_view(Yt, Bm, F, T).normal_(); ops.rows_view(Zt, Bn, F, T).normal_()
    temp = torch.tensor([5.1], device=dev)
    re = L.rows_tp(T) * F
    print(f"Bm={Bm} Bn={Bn}")
    print("  rows_sumsq(Y) %.1f us" % timeit(lambda: ops.rows_sumsq(Yt, Bm, re, re)))
    print("  sim gemm     %.1f us  (%.0f TF)" % ((t := timeit(lambda: ops.matmul_nt_splitk(Yt, Zt, Bm, Bn, re, re))), 2.0 * Bm * Bn * re / t / 1e6))
    loss, logits, cnt, ctx = E.clip_forward(Yt, Zt, temp, Bm=Bm, Bn=Bn, T=T)
    print("  clip_forward total %.1f us" % timeit(lambda: E.clip_forward(Yt, Zt, temp, Bm=Bm, Bn=Bn, T=T)))
    dZ = ops.new_rows(Bn, T, F, dt, dev)
    print("  dZ gemm      %.1f us  (%.0f TF)" % ((t := timeit(lambda: E.clip_backward(ctx, dZ))), 2.0 * Bm * Bn * re / t / 1e6))
    X = torch.randn(Bn, F, T, device=dev)
    print("  pack_rows(Y local) %.1f us" % timeit(lambda: ops.pack_rows(X, Zt)))
