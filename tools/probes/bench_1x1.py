"""The row-layout 1 x 1 convs of the config-2 step alone (f1, f2 forward; their data gradients), each with its K loop /
epilogue switched off in turn (diagnostic flags 256 = no epilogue, 512 = no K loop).  Diagnostic, not part of the product."""
import sys, os, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from speech_decoding_amd import ops


def timeit(fn, n=20, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def main():
    dev, dt = "cuda:0", torch.bfloat16
    B, T = 256, 360
    for (name, cin, cout, fwd) in [("f1 fwd", 320, 640, True), ("f2 fwd", 640, 1024, True), ("f2 dgrad", 1024, 640, False),
                                   ("f1 dgrad", 640, 320, False)]:
        x = ops.new_rows(B, T, cin, dt, dev); x.normal_()
        w = torch.randn(cout, cin, 1, device=dev) / math.sqrt(cin)
        wp = ops.pack_conv_weight(w, cout, cin, dt)
        y, u = ops.new_rows(B, T, cout, dt, dev), ops.new_rows(B, T, cout, dt, dev)
        bias = torch.zeros(cout, device=dev)
        stats = torch.zeros((B * ops.n_t_tiles(T), 2, cout), device=dev)
        full = dict(bias=bias, y_pre=u, gelu=True, stats=stats) if (fwd and cout == 1024) else \
            dict(bias=bias, y_pre=u, gelu=True) if fwd else dict()
        fullf = {k: v for k, v in full.items() if k != "stats"}
        if "stats" in full:
            fullf["row_sumsq"] = torch.zeros((x.shape[0], cout // 128), device=dev)
        fl = 2.0 * B * T * cin * cout
        wr = B * (T + 16) * cout * 2 * (2 if fwd else 1) / 1e6
        rd = B * (T + 16) * cin * 2 / 1e6
        for vn, kw in [("full", full), ("plain", dict()), ("no_epi", dict(flags=256)), ("no_main", dict(flags=512, **full)),
                       ("neither", dict(flags=768)),
                       ("flat", dict(flags=16384, **fullf)), ("flat_plain", dict(flags=16384)), ("flat_no_epi", dict(flags=16384 | 256)),
                       ("flat_no_main", dict(flags=16384 | 512, **fullf)), ("flat_same_order", dict(flags=16384 | 1024, **fullf)),
                       ("flat_no_prio", dict(flags=16384 | 2048, **fullf)), ("flat_no_prio_same", dict(flags=16384 | 2048 | 1024, **fullf)),
                       ("flat_1percu", dict(flags=16384 | 32768, **fullf)), ("flat_1percu_no_epi", dict(flags=16384 | 32768 | 256)),
                       ("flat_1percu_no_main", dict(flags=16384 | 32768 | 512, **fullf)),
                       ("flat_wrap_store", dict(flags=16384 | 32, **fullf)), ("flat_wrap_store_no_gelu", dict(flags=16384 | 32 | 16, **fullf)),
                       ("flat_no_main_wrap_store", dict(flags=16384 | 32 | 512, **fullf)),
                       ("flat_no_dma", dict(flags=16384 | 128, **fullf)), ("flat_no_dma_no_gelu", dict(flags=16384 | 128 | 16, **fullf)),
                       ("flat_no_dma_no_epi", dict(flags=16384 | 128 | 256)),
                       ("flat_no_store", dict(flags=16384 | 8, **fullf)), ("flat_no_gelu", dict(flags=16384 | 16, **fullf)),
                       ("flat_no_store_no_gelu", dict(flags=16384 | 24, **fullf)),
                       ("flat_no_main_no_store", dict(flags=16384 | 512 | 8, **fullf)), ("flat_no_main_no_gelu", dict(flags=16384 | 512 | 16, **fullf))]:
            us = timeit(lambda: ops.conv_gemm(x, wp, y, B=B, T=T, KS=1, dil=0, **kw))
            print(f"{name:9s} {cin:4d}->{cout:4d} {vn:8s} {us:7.1f} us  {fl / us / 1e6:6.1f} TF  (reads {rd:.0f} MB, writes {wr:.0f} MB"
                  f" -> {(rd + wr) / us:.2f} TB/s)", flush=True)


if __name__ == "__main__":
    main()
