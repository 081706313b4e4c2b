import sys, os, math
sys.path.insert(0, os.getcwd())
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
dev, dtype = "cuda:0", torch.bfloat16
B, T = 256, 360
for name, cin, cout in [("f2 fwd", 640, 1024), ("f2 dgrad", 1024, 640), ("f1 fwd", 320, 640)]:
    x = ops.new_rows(B, T, cin, dtype, dev); x.normal_()
    w = torch.randn(1, cout, cin, 1, device=dev) / math.sqrt(cin)
    wp = ops.pack_conv_weight(w, cout, cin, dtype)
    y, yp = ops.new_rows(B, T, cout, dtype, dev), ops.new_rows(B, T, cout, dtype, dev)
    bias = torch.zeros(cout, device=dev)
    stats = torch.empty((B * ops.n_t_tiles(T), 2, cout), device=dev)
    fl = 2.0 * B * T * cin * cout
    for tag, kw in [("gelu+pre+stats", dict(bias=bias, gelu=True, y_pre=yp, stats=stats)), ("gelu+pre", dict(bias=bias, gelu=True, y_pre=yp)),
                    ("gelu", dict(bias=bias, gelu=True)), ("plain", dict()), ("no_epi", dict(flags=256)), ("no_main", dict(bias=bias, gelu=True, y_pre=yp, flags=512)),
                    ("neither", dict(flags=768))]:
        us = timeit(lambda: ops.conv_gemm(x, wp, y, B=B, T=T, KS=1, dil=0, **kw))
        print(f"{name:10s} {tag:16s} {us:7.1f} us {fl/us/1e6:7.1f} TF", flush=True)
