"""Micro-benchmark of the 1x1 convs of the step (diagnostic): conv_final1/2 forward and data gradient, SubjectBlock GEMM."""
import sys, os, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
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

def main():
    dev, dtype = "cuda:0", torch.bfloat16
    B, T = 256, 360
    for name, cin, cout, gelu, pre, widx in [("x0 (composed SubjectBlock)", 256, 320, False, False, True), ("conv_final1 fwd", 320, 640, True, True, False),
                                             ("conv_final2 fwd", 640, 1024, True, True, False), ("conv_final2 dgrad", 1024, 640, False, False, False),
                                             ("conv_final1 dgrad", 640, 320, False, False, False)]:
        x = ops.new_rows(B, T, cin, dtype, dev); x.normal_()
        S = 27 if widx else 1
        w = torch.randn(S, cout, cin, 1, device=dev) / math.sqrt(cin)
        wp = ops.pack_conv_weight(w, cout, cin, dtype)
        y, yp = ops.new_rows(B, T, cout, dtype, dev), ops.new_rows(B, T, cout, dtype, dev)
        bias = torch.zeros(cout, device=dev)
        idx = torch.randint(0, S, (B,), dtype=torch.int32, device=dev) if widx else None
        fl = 2.0 * B * T * cin * cout
        mb = B * T * (cin + cout * (2 if pre else 1)) * 2 / 1e6
        for tag, flags in (("single", 0), ("paired", L.CONV_PAIR_TILES)):
            if widx and flags:
                continue
            us = timeit(lambda: ops.conv_gemm(x, wp, y, B=B, T=T, KS=1, dil=0, bias=bias, gelu=gelu, y_pre=yp if pre else None, widx=idx, flags=flags))
            print(f"{name:28s} {cin:4d}->{cout:4d} {tag:7s} {us:7.1f} us  {fl / us / 1e6:6.1f} TF  {mb:5.0f} MB  {mb / us:5.2f} TB/s", flush=True)

if __name__ == "__main__":
    main()
