"""Host-side cost of one training step (Python + launch overhead) against its GPU time (diagnostic): is the host far
enough ahead of the GPU that its time never shows?  Per step (bench.py's step, FusedAdam): host-only time (enqueue
without waiting), the GPU-drained time, and per phase how long the host takes to enqueue it."""
import os, sys, time, warnings, gc
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from speech_decoding_amd.layout import synthetic_positions
from speech_decoding_amd import BrainEncoder, CLIPLoss, load_config
from speech_decoding_amd import loss as sda_loss
from speech_decoding_amd.optim import FusedAdam
C, S, T, F = 208, 27, 360, 1024
dev = "cuda:0"
cfg = load_config(overrides=[f"num_subjects={S}", "compute_dtype=bf16"])
cfg["sensor_positions"] = synthetic_positions(C, 0).numpy()
with warnings.catch_warnings():
    warnings.simplefilter("ignore")
    enc = BrainEncoder(cfg).to(dev).train()
lossf = CLIPLoss(cfg).to(dev)
params = list(enc.parameters()) + list(lossf.parameters())
opt = FusedAdam(params, lr=3e-4)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
X = torch.randn(B, C, T, device=dev); Y = torch.randn(B, F, T, device=dev)
rng = np.random.RandomState(0)
one = torch.ones((), device=dev)
ph = {"prefetch": 0.0, "enc_fwd": 0.0, "loss+ranks": 0.0, "backward": 0.0, "adam": 0.0}
def step(acc=None):
    t = [time.perf_counter()]
    subj = torch.from_numpy(rng.randint(0, S, size=B).astype(np.int32))
    lossf.prefetch(Y, enc.compute_dtype); t.append(time.perf_counter())
    Z = enc(X, subj); t.append(time.perf_counter())
    loss = lossf(Y, Z); sda_loss.retrieval_ranks(Y, Z); t.append(time.perf_counter())
    opt.zero_grad(set_to_none=True); loss.backward(gradient=one); t.append(time.perf_counter())
    opt.step(); t.append(time.perf_counter())
    if acc is not None:
        for k, a, b in zip(acc, t, t[1:]): acc[k] += b - a
for _ in range(8): step()
torch.cuda.synchronize(); gc.collect(); gc.freeze()
N = 30
t0 = time.perf_counter()
for _ in range(N): step(ph)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"B={B}: host-only {1e3*(t1-t0)/N:.2f} ms/step, with GPU drain {1e3*(t2-t0)/N:.2f} ms/step")
print("host time per phase (ms):", {k: round(1e3 * v / N, 3) for k, v in ph.items()})
# the same with the GPU idle at every step's start (host latency fully exposed): step time = GPU time + whatever the host adds
torch.cuda.synchronize(); ts = []
for _ in range(10):
    torch.cuda.synchronize(); a = time.perf_counter(); step(); torch.cuda.synchronize(); ts.append(time.perf_counter() - a)
print(f"one step from an idle GPU: {1e3 * np.median(ts):.2f} ms")
