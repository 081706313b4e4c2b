"""Throughput of the GPU collate at the bench's batch (diagnostic): sda_collate_rows on a resident (B, C, T) batch
(Gwilliams2022Collator, gwilliams2022.py:640-661) and sda_collate_windows on resident recordings (ResidentSegments: the
window gather of gwilliams2022.py:129-142 fused with it) — rows/s, GB/s of algorithmic traffic (one read + one write of the
batch) and the share of a 7 ms training step one batch costs."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from speech_decoding_amd.collate import ResidentSegments, robust_scale_clamp

dev = "cuda:0"


def timeit(fn, n=20, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for (B, C, T, nb, name) in [(256, 208, 360, 60, "config 2 (208 ch x 360, batch 256)"), (512, 60, 360, 60, "config 4 per rank (60 ch, batch 512)"),
                            (512, 306, 1000, 100, "config 5 per rank (306 ch x 1000, batch 512)")]:
    X = torch.randn(B, C, T, device=dev) * 3 + 0.5
    us = timeit(lambda: robust_scale_clamp(X, nb, 20.0, True))
    rows, byt = B * C, 2.0 * B * C * T * 4
    print(f"collate_rows    {name:46s} {us:8.1f} us  {rows / us:8.2f} M rows/s  {byt / us / 1e3:7.1f} GB/s", flush=True)
    sessions = [torch.randn(C, 60000, device=dev) for _ in range(8)]
    rs = ResidentSegments(sessions, T, nb, 20.0, True)
    rng = np.random.RandomState(0)
    sidx, on = rng.randint(0, 8, B), rng.randint(0, 60000 - T, B)
    us2 = timeit(lambda: rs.batch(sidx, on))
    print(f"collate_windows {name:46s} {us2:8.1f} us  {rows / us2:8.2f} M rows/s  {byt / us2 / 1e3:7.1f} GB/s  (incl. the host's index upload)", flush=True)
