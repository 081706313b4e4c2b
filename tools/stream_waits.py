"""How long does the main stream sit idle at each cross-stream wait of a config-2 training step?  (diagnostic)
Also reports the main stream's total idle time per step: step wall time minus the sum of event-bracketed segments is not
available without a trace, so the waits are the part that scheduling can change."""
import os, sys, time, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from speech_decoding_amd import BrainEncoder, CLIPLoss, load_config
from speech_decoding_amd.layout import synthetic_positions
from speech_decoding_amd import loss as sda_loss
from speech_decoding_amd.optim import FusedAdam

C, S, T, F, B = 208, 27, 360, 1024, 256
dev = torch.device("cuda", 0)
torch.manual_seed(0); np.random.seed(0)
cfg = load_config(overrides=[f"num_subjects={S}", "compute_dtype=bf16", "dataset=Gwilliams2022"])
cfg["sensor_positions"] = synthetic_positions(C, seed=0).numpy()
with warnings.catch_warnings():
    warnings.simplefilter("ignore")
    enc = BrainEncoder(cfg).to(dev).train()
lossf = CLIPLoss(cfg).to(dev).train()
opt = FusedAdam(list(enc.parameters()) + list(lossf.parameters()), lr=3e-4)
g = torch.Generator(device=dev).manual_seed(1)
X = torch.randn(B, C, T, generator=g, device=dev)
Y = torch.randn(B, F, T, generator=g, device=dev)
rng = np.random.RandomState(0)
marks = []

def step(mark=False):
    subj = torch.from_numpy(rng.randint(0, S, size=B).astype(np.int32))
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(5)] if mark else None
    if mark: ev[0].record()
    lossf.prefetch(Y, enc.compute_dtype)
    Z = enc(X, subj)
    if mark: ev[1].record()
    loss = lossf(Y, Z)
    sda_loss.retrieval_ranks(Y, Z)
    if mark: ev[2].record()
    opt.zero_grad(set_to_none=True)
    loss.backward()
    if mark: ev[3].record()
    opt.step()
    if mark: ev[4].record(); marks.append(ev)

for _ in range(5): step()
torch.cuda.synchronize()
enc.engine.probe = []
n = 20
t0 = time.perf_counter()
for _ in range(n): step(True)
torch.cuda.synchronize()
wall = (time.perf_counter() - t0) / n * 1e3
by = {}
for label, e0, e1 in enc.engine.probe:
    by.setdefault(label, []).append(e0.elapsed_time(e1) * 1e3)
print(f"step {wall:.3f} ms (with probes)")
for k, v in by.items():
    print(f"  wait '{k}': {len(v) / n:.0f} per step, mean {np.mean(v):7.1f} us, median {np.median(v):7.1f}, max {np.max(v):7.1f}")
ph = np.array([[m[i].elapsed_time(m[i + 1]) * 1e3 for i in range(4)] for m in marks])
gap = np.array([marks[i][4].elapsed_time(marks[i + 1][0]) * 1e3 for i in range(len(marks) - 1)])
print("  main-stream phases (us, median): forward %.0f  loss+ranks %.0f  backward %.0f  adam %.0f  | between steps %.0f" %
      (*np.median(ph, axis=0), np.median(gap)))
