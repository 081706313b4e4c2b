"""A/B of engine switches on the config-2 training step in ONE process, interleaved rounds (box-to-box and run-to-run
variance is larger than most of the effects being compared).  Usage: python tools/ab_step.py [rounds]"""
import itertools, os, sys, time, warnings
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

def step():
    subj = torch.from_numpy(rng.randint(0, S, size=B).astype(np.int32))
    lossf.prefetch(Y, enc.compute_dtype)
    Z = enc(X, subj)
    loss = lossf(Y, Z)
    sda_loss.retrieval_ranks(Y, Z)
    opt.zero_grad(set_to_none=True)
    loss.backward()
    opt.step()

def timed(n=10):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): step()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3

# variants: name:attr=value,attr=value ... (or ';' between the pairs when a value contains a comma) (engine attributes; values are Python literals)
specs = sys.argv[2:] or ["base:", "flat_fwd_off:flat_tiles_forward=False"]
variants = {}
for spec in specs:
    name, _, rest = spec.partition(":")
    variants[name] = {kv.split("=")[0]: eval(kv.split("=")[1]) for kv in rest.split(";" if ";" in rest else ",") if kv}
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 4
eng = enc.engine
eng.main_priority = 0
defaults = {k: getattr(eng, k) for v in variants.values() for k in v}
eng.main_priority = 0                   # pseudo-attribute of this tool: priority of the stream the step is issued on
streams = {0: torch.cuda.current_stream(dev), -1: torch.cuda.Stream(device=dev, priority=-1)}
def apply(v):
    for k, d in defaults.items(): setattr(eng, k, d)
    for k, x in v.items(): setattr(eng, k, x)
    eng._seg_cache.clear()
    eng._side.clear()
res = {k: [] for k in variants}
for k, v in variants.items():           # warm-up of every variant (workspaces, attribute calls)
    apply(v)
    with torch.cuda.stream(streams[eng.main_priority]):
        for _ in range(3): step()
    torch.cuda.synchronize()
import gc
gc.collect(); gc.freeze()               # (a generation-2 collection inside a timed round is worth 80 ms)
for r in range(rounds):
    for k, v in variants.items():
        apply(v)
        with torch.cuda.stream(streams[eng.main_priority]):
            res[k].append(timed())
for k, v in res.items():
    print(f"{k:10s} median {np.median(v):.3f} ms  min {min(v):.3f}  all {[round(x, 3) for x in v]}")
