#!/usr/bin/env python3
"""bench.py — training throughput of the contrastive hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--dtype bf16|fp32] [--batch B_per_gpu]

One step = BrainEncoder forward + CLIPLoss + top-k ranks + backward + (N > 1: gradient all-reduce) + Adam
on one synthetic Gwilliams2022-shaped batch (208 sensors x 360 samples, 27 subjects, F = 1024,
256 segments per GPU; BASELINE.json configs[1] at N = 1, configs[2] at N = 8) that is already resident
in HBM.  Weak scaling: the per-GPU batch is fixed, negatives span the global batch.
Prints ONE JSON line on rank 0 (contract in the task statement) with `roofline` for the dominant kernel
(conv_gemm k=3, measured with HIP events on the launch stream) and, at N = 1, `cpu_baseline` (the CPU
oracle timed on this box's host cores on one config-② step).
"""
import argparse
import json
import os
import sys
import time
import warnings

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

C, S, T, F, D1, D2, K = 208, 27, 360, 1024, 270, 320, 32
PEAK_TFLOPS = {"bf16": 2500.0, "fp32": 157.3}      # dense MFMA peaks, MI355X_MICROARCH.md


def cpu_baseline(batch: int):
    """Oracle (CPU restatement of the reference, pinned on the golden fixtures) timed on the host cores:
    ONE config-② training step (forward + loss + backward + Adam) after a small warm-up."""
    from oracle import brain_oracle as O
    threads = torch.get_num_threads()
    loc = O.synthetic_positions(C, seed=0)
    P = O.seeded_params(C, S, D1, D2, F, K, seed=0, loc=loc)
    temp = torch.tensor([5.1])

    def one(B, seed):
        X, Y, subj = O.synthetic_batch(B, C, T, F, S, seed=seed)
        t0 = time.perf_counter()
        loss, Z, logits, grads = O.train_step(P, temp, X, Y, subj, loc=loc, drop_centre=3)
        params = [P[k] for k in grads if k != "temp" and grads[k] is not None]
        for p, k in zip(params, [k for k in grads if k != "temp" and grads[k] is not None]):
            p.grad = grads[k]
        opt = torch.optim.Adam([p.requires_grad_(True) for p in params], lr=3e-4)
        opt.step()
        O.topk_accuracy(Z, Y)
        for p in params:
            p.requires_grad_(False)
            p.grad = None
        return time.perf_counter() - t0

    one(16, 1)
    dt = one(batch, 2)
    return {"value": round(batch / dt, 3), "unit": "segments/s", "cores": threads, "kind": "port",
            "sample": f"1 training step (fwd+loss+top-k+bwd+Adam) at batch {batch}, 208ch x 360, fp32, "
                      f"{threads} torch threads, {dt:.1f} s"}


def pmc_traffic(dtype: str, tile_co: int, ks: int, kind: str = "conv_gemm"):
    """HBM bytes per launch of the dominant kernel from the committed PMC summary (separate rocprofv3 --pmc
    passes of this same command; tools/pmc_summary.py).  None when no summary matches."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_summary.json")))
    if not files:
        return None
    kernels = json.load(open(files[-1])).get("kernels", {})
    ctype = "unsigned short" if dtype == "bf16" else "float"
    pre = f"void sda::{kind}_kernel<{ctype}, {tile_co}, {ks},"
    hits = [v for k, v in kernels.items() if k.startswith(pre)]
    if not hits:
        return None
    n = sum(h["launches"] for h in hits)
    return round(sum(h["hbm_bytes_per_launch"] * h["launches"] for h in hits) / n)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--batch", type=int, default=256, help="segments per GPU")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-timer", action="store_true")
    ap.add_argument("--time-wgrad", action="store_true", help="also put HIP-event pairs around the weight-gradient GEMMs")
    a = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        if world == 1 and a.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node N for --gpus N > 1")
    import torch.distributed as dist
    # rehearsal-only overrides (several ranks on a one-GPU box): SDA_FORCE_DEVICE=0 SDA_DIST_BACKEND=gloo
    local = int(os.environ.get("SDA_FORCE_DEVICE", local))
    backend = os.environ.get("SDA_DIST_BACKEND", "nccl")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    from speech_decoding_amd import Classifier, BrainEncoder, CLIPLoss, load_config, ops
    from speech_decoding_amd.layout import synthetic_positions
    from speech_decoding_amd import loss as sda_loss
    from speech_decoding_amd.distributed import allreduce_gradients, broadcast_parameters

    torch.manual_seed(0)
    np.random.seed(0)
    loc = synthetic_positions(C, seed=0)        # (the oracle is imported by the cpu_baseline leg only)
    cfg = load_config(overrides=[f"num_subjects={S}", f"compute_dtype={a.dtype}", "dataset=Gwilliams2022"])
    cfg["sensor_positions"] = loc.numpy()
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        enc = BrainEncoder(cfg).to(dev).train()
    lossf = CLIPLoss(cfg).to(dev).train()
    broadcast_parameters(enc)
    broadcast_parameters(lossf)
    params = list(enc.parameters()) + list(lossf.parameters())
    from speech_decoding_amd.optim import FusedAdam        # same rule as torch.optim.Adam, one launch
    opt = FusedAdam(params, lr=float(cfg.lr))

    # synthetic data pool resident in HBM: X ~ N(0,1) clamped ±20; Y = P·X + 0.5·eps (learnable structure)
    B = a.batch
    g = torch.Generator(device=dev).manual_seed(1234 + rank)
    Pm = torch.randn(F, C, generator=torch.Generator().manual_seed(7)).to(dev) / np.sqrt(C)
    pool = []
    for i in range(3):
        X = torch.randn(B, C, T, generator=g, device=dev).clamp_(-20, 20)
        Y = torch.einsum("fc,bct->bft", Pm, X) + 0.5 * torch.randn(B, F, T, generator=g, device=dev)
        subj = torch.randint(0, S, (B,), generator=torch.Generator().manual_seed(100 * rank + i), dtype=torch.int32)
        pool.append((X, Y.contiguous(), subj))

    ranks_acc = []

    def step(i):
        X, Y, subj = pool[i % len(pool)]
        lossf.prefetch(Y, enc.compute_dtype)                 # pack Y (+ all-gather it under DP) while the encoder runs
        Z = enc(X, subj)
        loss = lossf(Y, Z)
        ranks_acc.append(sda_loss.retrieval_ranks(Y, Z))     # Classifier semantics (train.py:193-194), kept on device
        opt.zero_grad(set_to_none=True)
        loss.backward()
        if world > 1:      # encoder gradients were all-reduced inside backward (overlapped); temp is left
            allreduce_gradients(list(lossf.parameters()) if enc.grads_are_reduced else params)
        opt.step()
        return loss

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for i in range(a.warmup):
        step(i)
    ranks_acc.clear()
    # HIP-event timing of the dominant kernel's launches runs inside the timed region, on its last
    # `timed_tail` steps only: event pairs around every launch cost ~10 % of a step (they get in the way of
    # the two-stream overlap in backward), so instrumenting all K steps would distort `value`.
    timed_tail = 0 if a.no_kernel_timer else min(5, a.steps)
    timer = ops.KernelTimer(("conv_gemm", "wgrad_gemm") if a.time_wgrad else ("conv_gemm",)) if timed_tail else None
    fence()
    t0 = time.perf_counter()
    for i in range(a.steps):
        if timer is not None and i == a.steps - timed_tail:
            ops.TIMER = timer
        loss = step(a.warmup + i)
    fence()
    dt = time.perf_counter() - t0
    ops.TIMER = None
    tmax = torch.tensor([dt], device=dev)
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = float(tmax.item())

    cnt = torch.cat(ranks_acc[-3:]).float()
    top10 = float((cnt < 10).float().mean())
    final_loss = float(loss.detach())

    if rank == 0:
        out = {
            "metric": "train segments/sec (Gwilliams2022 208ch x 360, top-10 retrieval acc alongside)",
            "value": round(B * world * a.steps / dt, 2), "unit": "segments/s", "n_gpus": world, "steps": a.steps,
            "warmup": a.warmup, "ms_per_step": round(1e3 * dt / a.steps, 3), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": a.dtype, "data": "synthetic",
            "config": {"workload": f"Gwilliams2022-shaped MEG: {C} ch x {T} samples (3 s @120 Hz), {S} subjects, "
                                   f"F={F}, batch {B}/GPU, fwd+CLIP loss+top-k+bwd+Adam (BASELINE configs[1]"
                                   f"{' / configs[2]' if world == 8 else ''})",
                       "global_batch": B * world, "seq_len": T, "parallelism": f"dp{world}"},
            "top10_acc": round(top10, 4), "final_loss": round(final_loss, 4),
        }
        if timer is not None:
            summ = timer.summary()
            key = max(summ, key=lambda k: summ[k][2])             # dominant = most total time
            n, flops, ms = summ[key]
            ach = flops / (ms * 1e-3) / 1e12
            out["roofline"] = {"bound": "mfma", "achieved": round(ach, 2), "peak": PEAK_TFLOPS[a.dtype], "unit": "TFLOP/s",
                               "frac": round(ach / PEAK_TFLOPS[a.dtype], 4),
                               "traffic": pmc_traffic(a.dtype, key[2], key[3], key[0]),
                               "kernel": f"{key[0]}<{key[1]},TILE={key[2]},KS={key[3]}>", "launches": n,
                               "avg_us": round(1e3 * ms / n, 2)}
            out["kernel_time_ms_per_step"] = {f"{k[0]}<{k[1]},{k[2]},{k[3]}>": round(v[2] / timed_tail, 3) for k, v in summ.items()}
        if world == 1 and not a.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(B)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
