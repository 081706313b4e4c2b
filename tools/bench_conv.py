"""Micro-benchmark of conv_gemm / wgrad_gemm at config-2 shapes (diagnostic; not part of the product)."""
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
    return e0.elapsed_time(e1) / n * 1e3   # us

def main():
    dev = "cuda:0"
    B, T = 256, 360
    for dtype in ((torch.float32,) if os.environ.get("DTYPE") == "fp32" else (torch.bfloat16,)):
        for (cin, cout, KS, dil) in [(320, 320, 3, 4), (320, 640, 3, 2)]:
            x = ops.new_rows(B, T, cin, dtype, dev); x.normal_()
            w = torch.randn(cout, cin, KS, device=dev) / math.sqrt(KS * cin)
            wp = ops.pack_conv_weight(w, cout, cin, dtype)
            y = ops.new_rows(B, T, cout, dtype, dev)
            bias = torch.zeros(cout, device=dev)
            stats = torch.zeros((B * ops.n_t_tiles(T), 2, cout), device=dev)
            res = x if cin == cout else None
            fl = 2.0 * B * T * KS * cin * cout
            for name, kw in [("full", dict(bias=bias, res=res, stats=stats)), ("plain", dict()),
                             ("no_epi", dict(flags=256)), ("no_main", dict(bias=bias, res=res, stats=stats, flags=512)),
                             ("neither", dict(flags=768)), ("pair_full", dict(bias=bias, res=res, stats=stats, flags=8192)),
                             ("flat", dict(bias=bias, res=res, stats=stats, flags=16384)), ("flat_plain", dict(flags=16384)),
                             ("flat_noepi", dict(flags=16384 | 256)),

                             ("flat_noho", dict(bias=bias, res=res, stats=stats, flags=16384 | 64)),       # 64: no priority hand-over
                             ("flat_noepi_noho", dict(flags=16384 | 256 | 64)),
                             ("flat_1024", dict(bias=bias, res=res, stats=stats, flags=16384 | 1024)),
                             ("flat_2048", dict(bias=bias, res=res, stats=stats, flags=16384 | 2048)),
                             ("flat_3072", dict(bias=bias, res=res, stats=stats, flags=16384 | 3072)),
                             ("flat_1024_noho", dict(bias=bias, res=res, stats=stats, flags=16384 | 1024 | 64)),
                             ("flat_2048_noho", dict(bias=bias, res=res, stats=stats, flags=16384 | 2048 | 64)),
                             ("flat_noepi_nodma", dict(flags=16384 | 256 | 16)),       # diagnostic builds: no LDS-DMA in the K loop
                             ("flat_noepi_nodma_nolds", dict(flags=16384 | 256 | 8))]:  # ... and no fragment reads either (MFMA only)
                if KS != 3 and name.startswith("flat"):
                    continue
                us = timeit(lambda: ops.conv_gemm(x, wp, y, B=B, T=T, KS=KS, dil=dil, **kw))
                print(f"conv {str(dtype)[6:]:8s} {cin}->{cout} k{KS} {name:28s} {us:8.1f} us  {fl/us/1e6:7.1f} TF", flush=True)
            if KS == 3 or cout == 1024:
                dy = ops.new_rows(B, T, cout, dtype, dev); dy.normal_()
                tile_m = 160 if cout % 160 == 0 else 128
                ntiles = (cout // tile_m) * (cin // 64)
                nseg = max(1, min(B, round(int(os.environ.get("WGS", 256)) / ntiles)))
                import numpy as np
                seg = torch.from_numpy(np.floor(np.linspace(0, B, nseg + 1)).astype(np.int32)).to(dev)
                perm = torch.arange(B, dtype=torch.int32, device=dev)
                us = timeit(lambda: ops.wgrad_gemm(dy, x, B=B, T=T, KS=KS, dil=dil, perm=perm, seg_start=seg, nseg=nseg))
                print(f"wgrad(perm) {str(dtype)[6:]:8s} {cin}->{cout} k{KS} nseg={nseg:3d} {us:8.1f} us  {fl/us/1e6:7.1f} TF", flush=True)
                us = timeit(lambda: ops.wgrad_gemm(dy, x, B=B, T=T, KS=KS, dil=dil, perm=None, seg_start=seg, nseg=nseg))
                print(f"wgrad {str(dtype)[6:]:8s} {cin}->{cout} k{KS} nseg={nseg:3d} {us:8.1f} us  {fl/us/1e6:7.1f} TF", flush=True)

if __name__ == "__main__":
    main()
