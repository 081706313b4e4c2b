"""What-if (timing only, results are NOT checked): the config-2 training step captured into ONE HIP graph and replayed,
against the same step launched eagerly, interleaved in one process.  The naive capture bakes the subject indices and
Adam's step count into the graph, so its numbers are wrong after the first replay — only its duration means anything.
Usage: python tools/graph_probe.py [rounds]"""
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
params = list(enc.parameters()) + list(lossf.parameters())
opt = FusedAdam(params, lr=3e-4)
g = torch.Generator(device=dev).manual_seed(1)
X = torch.randn(B, C, T, generator=g, device=dev)
Y = torch.randn(B, F, T, generator=g, device=dev)
rng = np.random.RandomState(0)
enc.set_drop_centre(3)

def step(subj=None):
    if subj is None:
        subj = torch.from_numpy(rng.randint(0, S, size=B).astype(np.int32))
    lossf.prefetch(Y, enc.compute_dtype)
    Z = enc(X, subj)
    loss = lossf(Y, Z)
    sda_loss.retrieval_ranks(Y, Z)
    opt.zero_grad(set_to_none=True)
    loss.backward()
    opt.step()
    return loss

for _ in range(8): step()
torch.cuda.synchronize()
import gc
gc.collect(); gc.freeze()

def timed_eager(n=20):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): step()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3

fixed = torch.from_numpy(rng.randint(0, S, size=B).astype(np.int32))
graph = torch.cuda.CUDAGraph()
try:
    opt.zero_grad(set_to_none=True)
    with torch.cuda.graph(graph):
        step(fixed)
    torch.cuda.synchronize()
    ok = True
except Exception as e:                                   # noqa: BLE001
    import traceback
    traceback.print_exc()
    ok = False

def timed_graph(n=20):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): graph.replay()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 4
res = {"eager": [], "graph": []}
for r in range(rounds):
    res["eager"].append(timed_eager())
    if ok:
        res["graph"].append(timed_graph())
for k, v in res.items():
    if v:
        print(f"{k:6s} median {np.median(v):.3f} ms  min {min(v):.3f}  all {[round(x, 3) for x in v]}")
