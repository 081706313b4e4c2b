"""Latency of the small collectives a data-parallel step puts on its critical path, through RCCL at world size 1 (what one
GPU can measure: the launch + kernel cost of the collective itself, not the xGMI hop): the 2.5 KB SyncBN statistics
all-reduce (20 per step), the 3 * B_global-float row-statistics all-gather of the loss, the mask broadcast.
    MASTER_ADDR=127.0.0.1 MASTER_PORT=29533 RANK=0 WORLD_SIZE=1 python tools/rccl_latency.py"""
import datetime, os, time
import torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
dist.init_process_group("nccl", rank=0, world_size=1, timeout=datetime.timedelta(seconds=60), device_id=dev)

def timeit(fn, n=200, warm=20):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter(); e0.record()
    for _ in range(n): fn()
    e1.record(); host = (time.perf_counter() - t0) / n * 1e6
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3, host

bn = torch.zeros(2 * 320, device=dev)
rows = torch.zeros(3 * 2048, device=dev); out = torch.zeros(3 * 2048, device=dev)
mask = torch.zeros(208, device=dev)
for name, fn in [("all_reduce 2 x 320 floats (SyncBN statistics)", lambda: dist.all_reduce(bn)),
                 ("all_gather 3 x 2048 floats (loss row statistics)", lambda: dist.all_gather_into_tensor(out, rows)),
                 ("broadcast 208 floats (dropout mask)", lambda: dist.broadcast(mask, src=0))]:
    dev_us, host_us = timeit(fn)
    print(f"{name:52s} {dev_us:7.1f} us on the stream, {host_us:7.1f} us of host time per call")
dist.destroy_process_group()
