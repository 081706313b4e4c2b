"""GPU and host time of EVERY step of a free-running loop (no synchronisation inside; diagnostic): an event after each step's
optimiser launch gives the GPU-side step period, time.perf_counter the host's enqueue time.  Shows periodic slow episodes."""
import os, sys, time, warnings, gc
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from speech_decoding_amd.layout import synthetic_positions
from speech_decoding_amd import BrainEncoder, CLIPLoss, load_config
from speech_decoding_amd import loss as sda_loss
from speech_decoding_amd.optim import FusedAdam
C, S, T, F, B = 208, 27, 360, 1024, 256
dev = "cuda:0"
cfg = load_config(overrides=[f"num_subjects={S}", "compute_dtype=bf16"])
cfg["sensor_positions"] = synthetic_positions(C, 0).numpy()
with warnings.catch_warnings():
    warnings.simplefilter("ignore")
    enc = BrainEncoder(cfg).to(dev).train()
lossf = CLIPLoss(cfg).to(dev)
opt = FusedAdam(list(enc.parameters()) + list(lossf.parameters()), lr=3e-4)
X = torch.randn(B, C, T, device=dev); Y = torch.randn(B, F, T, device=dev)
rng = np.random.RandomState(0)
one = torch.ones((), device=dev)
def step():
    subj = torch.from_numpy(rng.randint(0, S, size=B).astype(np.int32))
    lossf.prefetch(Y, enc.compute_dtype)
    Z = enc(X, subj); loss = lossf(Y, Z); sda_loss.retrieval_ranks(Y, Z)
    opt.zero_grad(set_to_none=True); loss.backward(gradient=one); opt.step()
# the step's chain on a high-priority stream, as train.py / bench.py run it (speech_decoding_amd/streams.py);
# SDA_MAIN_PRIO=default keeps torch's default stream, SDA_MAIN_PRIO=0 an explicit normal-priority stream (diagnostic)
_prio = os.environ.get("SDA_MAIN_PRIO", "-1")
if _prio != "default":
    torch.cuda.set_stream(torch.cuda.Stream(device=dev, priority=int(_prio)))
for _ in range(8): step()
torch.cuda.synchronize(); gc.collect(); gc.freeze()
N = int(sys.argv[1]) if len(sys.argv) > 1 else 120
SETTLE = int(sys.argv[2]) if len(sys.argv) > 2 else 0      # steps between the heap collection and the measured loop (bench.py: 2)
for _ in range(SETTLE): step()
if SETTLE: torch.cuda.synchronize()
evs, host = [], []
e = torch.cuda.Event(enable_timing=True); e.record(); evs.append(e)
for _ in range(N):
    t0 = time.perf_counter(); step(); host.append((time.perf_counter() - t0) * 1e3)
    e = torch.cuda.Event(enable_timing=True); e.record(); evs.append(e)
torch.cuda.synchronize()
gpu = [evs[i].elapsed_time(evs[i + 1]) for i in range(N)]
print("gpu :", " ".join(f"{t:.1f}" for t in gpu))
print("host:", " ".join(f"{t:.1f}" for t in host))
print(f"gpu median {np.median(gpu):.3f} mean {np.mean(gpu):.3f}; host median {np.median(host):.3f} mean {np.mean(host):.3f}; gc counts {gc.get_count()}")
