"""Host-side cost of one training step (launch + Python overhead) vs GPU time (diagnostic)."""
import os, sys, time, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from speech_decoding_amd.layout import synthetic_positions
from speech_decoding_amd import BrainEncoder, CLIPLoss, load_config
from speech_decoding_amd import loss as sda_loss
C, S, T, F = 208, 27, 360, 1024
dev = "cuda:0"
cfg = load_config(overrides=[f"num_subjects={S}", "compute_dtype=bf16"])
cfg["sensor_positions"] = synthetic_positions(C, 0).numpy()
with warnings.catch_warnings():
    warnings.simplefilter("ignore")
    enc = BrainEncoder(cfg).to(dev).train()
lossf = CLIPLoss(cfg).to(dev)
params = list(enc.parameters()) + list(lossf.parameters())
opt = torch.optim.Adam(params, lr=3e-4)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
X = torch.randn(B, C, T, device=dev); Y = torch.randn(B, F, T, device=dev)
subj = torch.randint(0, S, (B,), dtype=torch.int32)
def step():
    Z = enc(X, subj); loss = lossf(Y, Z); sda_loss.retrieval_ranks(Y, Z)
    opt.zero_grad(set_to_none=True); loss.backward(); opt.step()
for _ in range(3): step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10): step()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"B={B}: host-only {1e3*(t1-t0)/10:.2f} ms/step, with GPU drain {1e3*(t2-t0)/10:.2f} ms/step")
