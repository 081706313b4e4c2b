"""Wall time of each of the first N config-2 training steps (synchronised per step): where does warm-up end? (diagnostic)"""
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
import gc
for kv in sys.argv[2:]:
    k, v = kv.split("=")
    if k == "gc":
        (gc.enable if eval(v) else gc.disable)()
        continue
    setattr(enc.engine, k, eval(v))
lossf = CLIPLoss(cfg).to(dev).train()
opt = FusedAdam(list(enc.parameters()) + list(lossf.parameters()), lr=3e-4)
g = torch.Generator(device=dev).manual_seed(1)
X = torch.randn(B, C, T, generator=g, device=dev)
Y = torch.randn(B, F, T, generator=g, device=dev)
rng = np.random.RandomState(0)
ts = []
for i in range(int(sys.argv[1]) if len(sys.argv) > 1 else 40):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    subj = torch.from_numpy(rng.randint(0, S, size=B).astype(np.int32))
    lossf.prefetch(Y, enc.compute_dtype)
    Z = enc(X, subj)
    loss = lossf(Y, Z)
    sda_loss.retrieval_ranks(Y, Z)
    opt.zero_grad(set_to_none=True)
    loss.backward()
    opt.step()
    torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
print(" ".join(f"{t:.1f}" for t in ts))
print("allocator:", {k: v for k, v in torch.cuda.memory_stats().items() if k in ("num_alloc_retries", "num_device_alloc", "num_device_free")})
