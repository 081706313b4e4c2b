"""Time one training step at the BASELINE.json per-GPU shapes (diagnostic): configs 2, 4, 5."""
import os, sys, time, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from speech_decoding_amd.layout import synthetic_positions
from speech_decoding_amd import BrainEncoder, CLIPLoss, load_config
from speech_decoding_amd.optim import FusedAdam
from speech_decoding_amd.streams import use_training_stream

def run(name, C, S, T, B, F=1024, dtype="bf16", steps=8):
    dev = "cuda:0"
    use_training_stream(dev)             # the step's chain on a high-priority stream, as bench.py / train.py run it
    cfg = load_config(overrides=[f"num_subjects={S}", f"compute_dtype={dtype}"])
    cfg["sensor_positions"] = synthetic_positions(C, 0).numpy()
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        enc = BrainEncoder(cfg).to(dev).train()
    lossf = CLIPLoss(cfg).to(dev)
    opt = FusedAdam(list(enc.parameters()) + list(lossf.parameters()), lr=3e-4)
    X = torch.randn(B, C, T, device=dev); Y = torch.randn(B, F, T, device=dev)
    subj = torch.randint(0, S, (B,), dtype=torch.int32)
    for i in range(3 + steps):
        if i == 3:
            torch.cuda.synchronize(); t0 = time.perf_counter()
        lossf.prefetch(Y, enc.compute_dtype)
        loss = lossf(Y, enc(X, subj))
        opt.zero_grad(set_to_none=True); loss.backward(); opt.step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    print(f"{name}: C={C} S={S} T={T} B={B} {dtype}: {dt*1e3:.2f} ms/step, {B/dt:.0f} seg/s, loss {float(loss):.3f}, "
          f"mem {torch.cuda.max_memory_allocated()/2**30:.1f} GiB", flush=True)

if __name__ == "__main__":
    run("config2", 208, 27, 360, 256)
    run("config4/GPU", 60, 1, 360, 512)
    run("config5/GPU", 306, 100, 1000, 512)
