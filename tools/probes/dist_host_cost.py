"""Host time of one small all-reduce through torch.distributed at world size 1 (what the data-parallel step pays 20+ times),
under the ProcessGroupNCCL knobs that might cheapen it.   python tools/probes/dist_host_cost.py [variant]"""
import datetime, os, sys, time
variant = sys.argv[1] if len(sys.argv) > 1 else "default"
if variant == "lean":
    os.environ["TORCH_NCCL_TRACE_BUFFER_SIZE"] = "0"
    os.environ["TORCH_NCCL_ENABLE_MONITORING"] = "0"
    os.environ["TORCH_NCCL_AVOID_RECORD_STREAMS"] = "1"
    os.environ["TORCH_NCCL_ASYNC_ERROR_HANDLING"] = "0"
    os.environ["TORCH_NCCL_CUDA_EVENT_CACHE"] = "1"
import torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29534")
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
dist.init_process_group("nccl", rank=0, world_size=1, timeout=datetime.timedelta(seconds=60), device_id=dev)
pg = dist.group.WORLD
bn = torch.zeros(2 * 320, device=dev)

def host(fn, n=2000, warm=100):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n): fn()
    h = (time.perf_counter() - t0) / n * 1e6
    torch.cuda.synchronize()
    return h

opts = dist.AllreduceOptions()
print(variant, "dist.all_reduce            %.1f us" % host(lambda: dist.all_reduce(bn)))
print(variant, "dist.all_reduce async+wait %.1f us" % host(lambda: dist.all_reduce(bn, async_op=True).wait()))
print(variant, "pg.allreduce([t]).wait()   %.1f us" % host(lambda: pg.allreduce([bn], opts).wait()))
print(variant, "torch.empty(640)           %.1f us" % host(lambda: torch.empty(640, device=dev)))
e = torch.cuda.Event()
s2 = torch.cuda.Stream()
print(variant, "event record + wait        %.1f us" % host(lambda: (e.record(), s2.wait_event(e))))
dist.destroy_process_group()
