"""pack_rows at the step's two shapes (diagnostic)."""
import sys, os
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
dev = "cuda:0"
for name, B, C, T, Cp in [("X 208 -> 256", 256, 208, 360, 256), ("Y 1024", 256, 1024, 360, 1024)]:
    x = torch.randn(B, C, T, device=dev)
    for dt in (torch.bfloat16, torch.float32):
        buf = ops.new_rows(B, T, Cp, dt, dev)
        us = timeit(lambda: ops.pack_rows(x, buf))
        mb = (B * C * T * 4 + B * T * Cp * buf.element_size()) / 1e6
        print(f"{name:14s} {str(dt)[6:]:9s} {us:7.1f} us  {mb:6.0f} MB  {mb / us:5.2f} TB/s", flush=True)
