"""reduce_unpack_wgrad alone at the step's shapes (diagnostic): the K-split slab sums that follow every weight-gradient GEMM."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from speech_decoding_amd import ops


def timeit(fn, n=50, warm=5):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for (cout, cin, ks, nseg) in [(320, 320, 3, 25), (640, 320, 3, 12), (320, 320, 3, 51), (1024, 640, 1, 3), (640, 320, 1, 12)]:
    slabs = torch.randn(nseg, ks, cout, cin, device="cuda:0")
    us = timeit(lambda: ops.reduce_unpack_wgrad(slabs, cout, cin, ks))
    mb = slabs.numel() * 4 / 1e6
    print(f"reduce_unpack {cout}x{cin}x{ks} nseg={nseg:3d}: {us:6.1f} us  ({mb:.0f} MB -> {mb / us / 1e3 * 1e3:.0f} GB/s)", flush=True)
