"""The four 1 x 1 convs of the config-2 step alone on the three kernels: conv_gemm's tile per workgroup, conv1_flat
(SDA_CONV_FLAT_TILES) and conv1_wide (SDA_CONV_WIDE_TILES), with the epilogue each has in the step and plain (diagnostic)."""
import sys, os, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from speech_decoding_amd import ops, lib as L


def timeit(fn, n=20, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


dev, dt = "cuda:0", torch.bfloat16
B, T = 256, 360
for (name, cin, cout, kind) in [("f1 fwd", 320, 640, "fwd"), ("f2 fwd", 640, 1024, "fwd_rsq"), ("f2 dgrad", 1024, 640, "gbwd"), ("f1 dgrad", 640, 320, "plain")]:
    x = ops.new_rows(B, T, cin, dt, dev); x.normal_()
    wp = ops.pack_conv_weight(torch.randn(cout, cin, 1, device=dev) / math.sqrt(cin), cout, cin, dt)
    y, u = ops.new_rows(B, T, cout, dt, dev), ops.new_rows(B, T, cout, dt, dev)
    u.normal_()
    bias = torch.zeros(cout, device=dev)
    fl = 2.0 * B * T * cin * cout
    for label, flags in [("tile", 0), ("flat", L.CONV_FLAT_TILES), ("wide", L.CONV_WIDE_TILES)]:
        def run(plain=False):
            if plain or kind == "plain":
                return ops.conv_gemm(x, wp, y, B=B, T=T, KS=1, dil=0, flags=flags)
            if kind == "fwd":
                return ops.conv_gemm(x, wp, y, B=B, T=T, KS=1, dil=0, bias=bias, y_pre=u, gelu=True, flags=flags)
            if kind == "fwd_rsq":
                if flags:
                    parts = torch.empty((x.shape[0], cout // 128), device=dev)
                    return ops.conv_gemm(x, wp, y, B=B, T=T, KS=1, dil=0, bias=bias, y_pre=u, gelu=True, row_sumsq=parts, flags=flags)
                st = torch.empty((B * ops.n_t_tiles(T), 2, cout), device=dev)
                return ops.conv_gemm(x, wp, y, B=B, T=T, KS=1, dil=0, bias=bias, y_pre=u, gelu=True, stats=st)
            st = torch.empty((ops.conv_stats_rows(B, T, 1, cout, flags | L.EPI_GELU_BWD), 2, cout), device=dev)
            return ops.conv_gemm(x, wp, y, B=B, T=T, KS=1, dil=0, gelu_bwd_u=u, stats=st, flags=flags)
        us, usp = timeit(run), timeit(lambda: run(True))
        print(f"{name:9s} {cin:4d}->{cout:4d} {label:5s} step form {us:7.1f} us {fl / us / 1e6:6.1f} TF | plain {usp:7.1f} us {fl / usp / 1e6:6.1f} TF", flush=True)
