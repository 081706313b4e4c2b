"""Micro-benchmark of the HBM-bound row passes at config-2 shapes (diagnostic; not part of the product)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from speech_decoding_amd import ops

def timeit(fn, n=30, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3   # us

def main():
    dev = "cuda:0"
    B, T, C = 256, 360, 320
    for dtype in (torch.bfloat16, torch.float32):
        es = 2 if dtype != torch.float32 else 4
        x = ops.new_rows(B, T, C, dtype, dev); x.normal_()
        dy = ops.new_rows(B, T, C, dtype, dev); dy.normal_()
        y = ops.new_rows(B, T, C, dtype, dev)
        x2 = ops.new_rows(B, T, 2 * C, dtype, dev); x2.normal_()
        dx2 = ops.new_rows(B, T, 2 * C, dtype, dev)
        f = lambda *s: torch.randn(*s, device=dev)
        scale, shift, mean, gamma, beta = f(C), f(C), f(C), f(C), f(C)
        rstd = torch.rand(C, device=dev) + 0.5
        scratch = ops.reduce_scratch(2 * C, dev)
        stats = torch.randn(B * ops.n_t_tiles(T), 2, C, device=dev)
        mb = B * T * C * es / 1e6
        cases = [
            ("bn_gelu_forward", 2 * mb, lambda: ops.bn_gelu_forward(x, y, scale, shift, B, T)),
            ("bn_gelu_backward(from stats)", 3 * mb, lambda: ops.bn_gelu_backward(dy, x, mean, rstd, gamma, beta, y, B, T, scratch, tile_stats=stats)),
            ("bn_gelu_backward(own sums)", 5 * mb, lambda: ops.bn_gelu_backward(dy, x, mean, rstd, gamma, beta, y, B, T, scratch)),
            ("glu_forward", 3 * mb, lambda: ops.glu_forward(x2, y, B, T)),
            ("glu_backward_colsum", 5 * mb, lambda: ops.glu_backward_colsum(x2, dy, dx2, B, T, scratch)),
            ("gelu_backward_colsum", 3 * mb, lambda: ops.gelu_backward_colsum(x, dy, y, B, T, scratch)),
            ("glu_backward", 5 * mb, lambda: ops.glu_backward(x2, dy, dx2, B, T)),
            ("gelu_backward", 3 * mb, lambda: ops.gelu_backward(x, dy, y, B, T)),
            ("colsum", mb, lambda: ops.colsum(x, B, T, scratch)),
            # what the chip gives library kernels on the same buffers (the practical ceiling for 1:1 / 2:1 / 0:1 traffic)
            ("torch copy_ (1 read : 1 write)", 2 * mb, lambda: y.copy_(x)),
            ("torch add (2 reads : 1 write)", 3 * mb, lambda: torch.add(x, dy, out=y)),
            ("torch zero_ (write only)", mb, lambda: y.zero_()),
            ("torch sum (read only)", mb, lambda: x.sum()),
        ]
        for name, mbytes, fn in cases:
            us = timeit(fn)
            print(f"{str(dtype)[6:]:9s} {name:30s} {us:7.1f} us  {mbytes:6.0f} MB  {mbytes / us:5.2f} TB/s", flush=True)

if __name__ == "__main__":
    main()
