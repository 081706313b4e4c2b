"""Per-phase GPU time of one training step (diagnostic): encoder fwd / loss fwd+ranks / backward / Adam."""
import os, sys, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from speech_decoding_amd.layout import synthetic_positions
from speech_decoding_amd import BrainEncoder, CLIPLoss, load_config
from speech_decoding_amd import loss as sda_loss
C, S, T, F = 208, 27, 360, 1024
dev = "cuda:0"
dtype = sys.argv[1] if len(sys.argv) > 1 else "bf16"
cfg = load_config(overrides=[f"num_subjects={S}", f"compute_dtype={dtype}"])
cfg["sensor_positions"] = synthetic_positions(C, 0).numpy()
with warnings.catch_warnings():
    warnings.simplefilter("ignore")
    enc = BrainEncoder(cfg).to(dev).train()
lossf = CLIPLoss(cfg).to(dev)
opt = torch.optim.Adam(list(enc.parameters()) + list(lossf.parameters()), lr=3e-4)
B = 256
X = torch.randn(B, C, T, device=dev); Y = torch.randn(B, F, T, device=dev)
subj = torch.randint(0, S, (B,), dtype=torch.int32)
names = ["enc_fwd", "loss_fwd", "ranks", "backward", "adam"]
acc = {n: 0.0 for n in names}
def ev():
    e = torch.cuda.Event(enable_timing=True); e.record(); return e
N = 20
for it in range(5 + N):
    torch.cuda.synchronize()
    e = [ev()]
    Z = enc(X, subj); torch.cuda.synchronize(); e.append(ev())
    loss = lossf(Y, Z); torch.cuda.synchronize(); e.append(ev())
    sda_loss.retrieval_ranks(Y, Z); torch.cuda.synchronize(); e.append(ev())
    opt.zero_grad(set_to_none=True); loss.backward(); torch.cuda.synchronize(); e.append(ev())
    opt.step(); torch.cuda.synchronize(); e.append(ev()); torch.cuda.synchronize()
    if it >= 5:
        for i, n in enumerate(names):
            acc[n] += e[i].elapsed_time(e[i + 1])
print(dtype, {n: round(v / N, 3) for n, v in acc.items()}, "sum", round(sum(acc.values()) / N, 3))
