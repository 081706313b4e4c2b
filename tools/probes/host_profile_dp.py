"""Where the HOST's time goes in a data-parallel step (diagnostic): the step of tools/step_series.py with the data-parallel
path on at world size 1 (SDA_DP_SINGLE_RANK=1, emulate_world(N)), 30 steps under cProfile.
    python tools/probes/host_profile_dp.py [N=2]"""
import os, sys, time, warnings, gc, socket, cProfile, pstats
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch, torch.distributed as dist
N = int(sys.argv[1]) if len(sys.argv) > 1 else 2
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
if N > 1:
    os.environ["SDA_DP_SINGLE_RANK"] = "1"
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0)); port = sk.getsockname()[1]
    from speech_decoding_amd import distributed as sd
    sd.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1, device_id=dev)
    sd.emulate_world(N)
    from speech_decoding_amd import loss as _l
    _l.EMULATE_COPY_REMOTE = False
from speech_decoding_amd.layout import synthetic_positions
from speech_decoding_amd import BrainEncoder, CLIPLoss, load_config
from speech_decoding_amd import loss as sda_loss
from speech_decoding_amd.optim import FusedAdam
from speech_decoding_amd.distributed import allreduce_gradients
C, S, T, F, B = 208, 27, 360, 1024, 256
cfg = load_config(overrides=[f"num_subjects={S}", "compute_dtype=bf16"])
cfg["sensor_positions"] = synthetic_positions(C, 0).numpy()
with warnings.catch_warnings():
    warnings.simplefilter("ignore")
    enc = BrainEncoder(cfg).to(dev).train()
lossf = CLIPLoss(cfg).to(dev)
params = list(enc.parameters()) + list(lossf.parameters())
opt = FusedAdam(params, lr=3e-4)
X = torch.randn(B, C, T, device=dev); Y = torch.randn(B, F, T, device=dev)
rng = np.random.RandomState(0)
one = torch.ones((), device=dev)
def step():
    subj = torch.from_numpy(rng.randint(0, S, size=B).astype(np.int32))
    lossf.prefetch(Y, enc.compute_dtype)
    Z = enc(X, subj); loss = lossf(Y, Z); sda_loss.retrieval_ranks(Y, Z)
    opt.zero_grad(set_to_none=True); loss.backward(gradient=one)
    if N > 1:
        allreduce_gradients(list(lossf.parameters()) if enc.grads_are_reduced else params)
    opt.step()
torch.cuda.set_stream(torch.cuda.Stream(device=dev, priority=-1))
for _ in range(8): step()
torch.cuda.synchronize(); gc.collect(); gc.freeze()
t0 = time.perf_counter()
for _ in range(30): step()
h = (time.perf_counter() - t0) / 30 * 1e3
torch.cuda.synchronize()
print(f"host per step {h:.2f} ms (not profiled)")
# (backward in the calling thread, so that the profiler sees inside it)
with torch.autograd.set_multithreading_enabled(False):
    for _ in range(3): step()
    pr = cProfile.Profile(); pr.enable()
    for _ in range(30): step()
    pr.disable(); torch.cuda.synchronize()
st = pstats.Stats(pr); st.sort_stats("tottime").print_stats(22)
st.sort_stats("cumtime").print_stats(30)
