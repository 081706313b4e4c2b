"""Does the similarity matmul's time depend on WHERE its operands sit?  (diagnostic)  The same 256 x 256 x 385 024 split-K
product with the two packed operands at different distances from each other / different base alignments."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from speech_decoding_amd import ops, lib as L

dev = "cuda:0"; T, F, B = 360, 1024, 256
dt = torch.bfloat16
re = L.rows_tp(T) * F
n = L.rows_alloc(B, T) * F


def timeit(fn, n_=10, warm=2):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n_): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n_ * 1e3


pool = torch.empty(3 * n + (64 << 20), dtype=dt, device=dev)
pool.normal_()
flush = torch.empty(512 << 20, dtype=torch.uint8, device=dev)
for off_y, off_z in [(0, n), (0, n + 64), (0, n + 2048), (0, n + 65536), (0, n + 4096 * 37), (128, n + 128), (0, 2 * n), (4096, n + (1 << 20)), (0, n + (3 << 20) + 8192)]:
    Yt = pool[off_y: off_y + n].view(-1, F)
    Zt = pool[off_z: off_z + n].view(-1, F)
    warm = timeit(lambda: ops.matmul_nt_splitk(Yt, Zt, B, B, re, re))
    def cold():
        flush.zero_()                         # push the operands out of the Infinity Cache, as in the step
        ops.matmul_nt_splitk(Yt, Zt, B, B, re, re)
    c = timeit(cold, 6, 1) - timeit(lambda: flush.zero_(), 6, 1)
    print(f"Y at +{off_y * 2:>9d} B, Z at +{off_z * 2:>10d} B ({(off_z - off_y) * 2 % (1 << 20):>8d} mod 1 MiB): back to back {warm:7.1f} us, cache-cold {c:7.1f} us", flush=True)
