#!/usr/bin/env python3
"""bench.py — training throughput of the contrastive hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--dtype bf16|fp16|fp32] [--batch B_per_gpu]

One step = BrainEncoder forward + CLIPLoss + top-k ranks + backward + (N > 1: gradient all-reduce) + Adam
on one synthetic Gwilliams2022-shaped batch (208 sensors x 360 samples, 27 subjects, F = 1024,
256 segments per GPU; BASELINE.json configs[1] at N = 1, configs[2] at N = 8) that is already resident
in HBM.  Weak scaling: the per-GPU batch is fixed, negatives span the global batch.

`--config 4 | 5` runs the per-GPU shape of BASELINE configs[3] / configs[4] instead (60 ch x 360, 1 subject, 512 / GPU; 306 ch x
1000, 100 subjects, 512 / GPU, fp16); `--emulate-world 8` runs configs[2]'s PER-RANK step on the one GPU (2048 speech rows, 1792
of them resident stand-ins for the all-gather's delivery; every collective issued through RCCL at world size 1).

`python bench.py --gpus N` with N > 1 starts its own ranks: a CHILD `python -m torch.distributed.run
--nproc-per-node N bench.py ...` is spawned before this process touches the GPU, and its exit code is returned.
Under `torch.distributed.run` (RANK/WORLD_SIZE set) it runs as one rank.

Timing: W untimed warm-up steps, then EXACTLY K steps between barrier + synchronize fences with NO instrumentation
inside; `value` = global segments / max-over-ranks time.  The roofline leg (HIP-event pairs around every launch of the
GEMM-class kernels, on the stream they are launched on) runs AFTER the timed region on a few extra steps, so it cannot
change `value`.  Prints ONE JSON line on rank 0 with `roofline` (dominant kernel) and, at N = 1, `cpu_baseline`
(the CPU oracle timed on this box's host cores, BASELINE.md §3 procedure on a bounded sample).
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time
import warnings

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

C, S, T, F, D1, D2, K = 208, 27, 360, 1024, 270, 320, 32
# per-GPU shapes of the BASELINE.json configs that run on one MI355X (`--config N`, circled numbers of SURVEY.md §8d):
#   2 = configs[1] (the headline), 4 = configs[3] per rank (60 ch, 1 subject, 512 / GPU), 5 = configs[4] per rank
#   (306 ch x 1000 samples, 100 subjects, 512 / GPU, fp16); configs[2] per rank = config 2 with --emulate-world 8
CONFIGS = {2: dict(C=208, S=27, T=360, batch=256, dtype="bf16", name="configs[1]"),
           4: dict(C=60, S=1, T=360, batch=512, dtype="bf16", name="configs[3] per-rank shape (60 ch, batch 4096 global = 512 x 8)"),
           5: dict(C=306, S=100, T=1000, batch=512, dtype="fp16", name="configs[4] per-rank shape (306 ch x 5 s @200 Hz, batch 4096 global = 512 x 8)")}
PEAK_TFLOPS = {"bf16": 2500.0, "fp16": 2500.0, "fp32": 157.3}      # dense MFMA peaks, MI355X_MICROARCH.md


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--dtype", default=None, choices=["bf16", "fp16", "fp32"], help="default: the config's (bf16; fp16 for config 5)")
    ap.add_argument("--batch", type=int, default=None, help="segments per GPU (default: the config's, 256 / 512 / 512)")
    ap.add_argument("--config", type=int, default=2, choices=sorted(CONFIGS), help="BASELINE config whose per-GPU shape is run")
    ap.add_argument("--emulate-world", type=int, default=1, help="N = 1 only: run the per-rank step of an N-rank job — N x batch "
                    "speech rows ((N - 1) x batch of them resident stand-ins for what the all-gather would have delivered), "
                    "every collective issued for real through RCCL at world size 1.  A compute-side bound, not a scaling curve")
    ap.add_argument("--emulate-no-copy", action="store_true", help="--emulate-world: leave the stand-in rows in place instead of "
                    "re-delivering them (a 1.3 GB device copy beside the forward) every step")
    ap.add_argument("--no-feed-leg", action="store_true", help="skip the extra loop fed by ResidentSegmentFeed + ShardedRandomSampler")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-timer", action="store_true")
    ap.add_argument("--timer-steps", type=int, default=3, help="instrumented steps run after the timed region")
    ap.add_argument("--no-prefetch-ahead", action="store_true", help="N > 1: gather each step's speech rows at its own start")
    ap.add_argument("--no-host-sync-leg", action="store_true", help="skip the extra loop that reads loss/ranks back every step")
    ap.add_argument("--coll-timer-steps", type=int, default=0, help="N > 1: extra steps (after the timed region) with HIP-event "
                    "brackets around every collective; adds `collectives_us_per_step` to the JSON line")
    return ap.parse_args()


def spawn_ranks(a):
    """--gpus N from a bare `python bench.py`: start N ranks as a child process (nothing here has touched the GPU:
    torch is not even imported yet) and hand back its exit code."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={a.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


# ----------------------------------------------------------------------------------------------- CPU baseline
def host_cores():
    """Cores this process may use: CPU affinity, cgroup CPU quota and physical (non-SMT) cores, whichever is smallest."""
    aff = sorted(os.sched_getaffinity(0))
    n = len(aff)
    phys = set()
    try:
        cur = {}
        for line in open("/proc/cpuinfo"):
            if ":" in line:
                k, v = [x.strip() for x in line.split(":", 1)]
                cur[k] = v
            elif cur:
                if int(cur.get("processor", -1)) in aff:
                    phys.add((cur.get("physical id", "0"), cur.get("core id", cur.get("processor"))))
                cur = {}
        if phys:
            n = min(n, len(phys))
    except OSError:
        pass
    model = "?"
    try:
        model = next(l.split(":", 1)[1].strip() for l in open("/proc/cpuinfo") if l.startswith("model name"))
    except (OSError, StopIteration):
        pass
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(float(txt[0]) / float(txt[1]) + 0.5)))
            else:
                q = int(txt[0])
                if q > 0:
                    per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                    n = min(n, max(1, int(q / per + 0.5)))
        except (OSError, ValueError, IndexError):
            pass
    return max(1, n), model


def cpu_baseline(batch: int, budget_s: float = 75.0):
    """Oracle (CPU restatement of the reference, pinned on the golden fixtures) timed on the host cores, following
    BASELINE.md §3: torch threads = physical cores available to this process, 2 warm-up steps, then timed training
    steps (forward + loss + top-k as a matmul + backward + Adam) at config ② (208 ch, batch 256, 27 subjects) and
    config ① (60 ch, batch 64, 1 subject); bounded: each config stops after 5 timed steps or its share of `budget_s` seconds."""
    import torch
    from oracle import brain_oracle as O
    cores, model = host_cores()
    torch.set_num_threads(cores)
    temp = torch.tensor([5.1])

    def run(Cc, Ss, B, max_steps, budget):
        loc = O.synthetic_positions(Cc, seed=0)
        P = O.seeded_params(Cc, Ss, D1, D2, F, K, seed=0, loc=loc)
        names = None
        opt = None
        times = []

        def one(Bx, seed):
            nonlocal names, opt
            X, Y, subj = O.synthetic_batch(Bx, Cc, T, F, Ss, seed=seed)
            t0 = time.perf_counter()
            loss, Z, logits, grads = O.train_step(P, temp, X, Y, subj, loc=loc, drop_centre=3)
            if names is None:
                names = [k for k in grads if k != "temp" and grads[k] is not None]
                opt = torch.optim.Adam([P[k].requires_grad_(True) for k in names], lr=3e-4)
            for k in names:
                P[k].grad = grads[k]
            opt.step()
            O.topk_accuracy(Z, Y)
            for k in names:
                P[k].grad = None
            return time.perf_counter() - t0

        one(12, 1)                                    # 2 warm-up steps (thread pool, allocator, mkldnn primitives): a small one
        one(B, 2)                                     # and one at the measured size
        t_start = time.perf_counter()
        while len(times) < max_steps and (not times or time.perf_counter() - t_start + times[-1] < budget):
            times.append(one(B, 10 + len(times)))
        return B * len(times) / sum(times), times

    v2, t2 = run(C, S, batch, 5, budget_s)           # >= 5 timed steps
    v1, t1 = run(60, 1, 64, 5, budget_s / 4)
    par = [l.strip() for l in torch.__config__.parallel_info().splitlines() if "get_num_threads" in l or "OpenMP" in l or "MKL" in l.upper()]
    return {"value": round(v2, 3), "unit": "segments/s", "cores": cores, "kind": "port",
            "sample": f"config 2 (208ch x 360, batch {batch}, 27 subj): {len(t2)} timed training steps of "
                      f"{', '.join(f'{x:.1f}' for x in t2)} s after 2 warm-up steps (one at full size); config 1 (60ch, batch 64, 1 subj): "
                      f"{len(t1)} steps -> {v1:.2f} segments/s; fp32, step = fwd+loss+top-k(matmul)+bwd+Adam",
            "config1_value": round(v1, 3), "cpu_model": model, "torch_threads": cores,
            "parallel_info": "; ".join(par)[:300]}


def pmc_traffic(dtype: str, tile_co: int, ks: int, kind: str = "conv_gemm"):
    """HBM bytes per launch of the dominant kernel, launch-weighted over its instantiations, from the newest committed
    PMC summary (separate `rocprofv3 --pmc` passes of this same command; tools/pmc_summary.py).  Returns
    (bytes or None, source file or None): the counters cannot be collected from inside this process."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_summary.json")))
    if not files:
        return None, None
    kernels = json.load(open(files[-1])).get("kernels", {})
    ctype = {"bf16": "unsigned short", "fp16": "_Float16", "fp32": "float"}[dtype]
    hits = [v for k, v in kernels.items() if f"{kind}_kernel<{ctype}, {tile_co}, {ks}," in k]
    if kind == "conv_gemm" and ks == 3 and tile_co == 160:          # the same family's flat-tile instantiations (conv3_flat.hip)
        hits += [v for k, v in kernels.items() if f"conv3_flat_kernel<{ctype}," in k]
    if not hits:
        return None, os.path.relpath(files[-1], ROOT)
    n = sum(h["launches"] for h in hits)
    return round(sum(h["hbm_bytes_per_launch"] * h["launches"] for h in hits) / n), os.path.relpath(files[-1], ROOT)


def main():
    global C, S, T
    a = parse_args()
    shape = CONFIGS[a.config]
    C, S, T = shape["C"], shape["S"], shape["T"]
    a.batch = a.batch or shape["batch"]
    a.dtype = a.dtype or shape["dtype"]
    if a.emulate_world > 1 and a.gpus > 1:
        sys.exit("bench.py: --emulate-world is a one-GPU measurement (use --gpus N on a node that has N)")
    if a.gpus > 1 and "RANK" not in os.environ:
        sys.exit(spawn_ranks(a))

    # ONE JSON line on stdout, whatever the libraries below choose to print there (RCCL's version banner goes to stdout):
    # everything written to file descriptor 1 from here on lands on stderr, the line itself goes to the saved descriptor
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    import numpy as np
    import torch
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    import torch.distributed as dist
    # rehearsal-only overrides (several ranks on a one-GPU box): SDA_FORCE_DEVICE=0 SDA_DIST_BACKEND=gloo
    local = int(os.environ.get("SDA_FORCE_DEVICE", local))
    backend = os.environ.get("SDA_DIST_BACKEND", "nccl")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if not os.environ.get("SDA_DEFAULT_STREAM"):      # (diagnostic: SDA_DEFAULT_STREAM=1 keeps torch's default stream)
        from speech_decoding_amd.streams import use_training_stream
        use_training_stream(dev)                      # the step's chain on a high-priority stream (side streams stay normal)
    emu = a.emulate_world if world == 1 else 1
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend == "nccl":
            from speech_decoding_amd.distributed import init_process_group as sda_init_pg
            sda_init_pg("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)
    elif emu > 1:
        # one rank, data-parallel code path ON: every collective of the step goes through the backend (RCCL) at world size 1
        os.environ["SDA_DP_SINGLE_RANK"] = "1"
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        kw = dict(device_id=dev) if backend == "nccl" else {}
        from speech_decoding_amd.distributed import init_process_group as sda_init_pg
        sda_init_pg(backend, init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1, **kw)
        from speech_decoding_amd import distributed as sda_dist
        from speech_decoding_amd import loss as _l
        sda_dist.emulate_world(emu)
        _l.EMULATE_COPY_REMOTE = not a.emulate_no_copy
    dp = world > 1 or emu > 1

    from speech_decoding_amd import BrainEncoder, CLIPLoss, load_config, ops
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
    from speech_decoding_amd.amp import LossScaler
    scaler = LossScaler.for_dtype(enc.compute_dtype, global_batch=a.batch * world * emu, T=T)   # static loss scale for fp16 (grows with
                                                                                           # the global batch); a no-op for bf16 / fp32

    # synthetic data pool resident in HBM: X ~ N(0,1) clamped ±20; Y = P·X + 0.5·eps (learnable structure).
    # Subject indices are drawn FRESH every step (as a data loader delivers them), so the per-step index uploads are real.
    B = a.batch
    g = torch.Generator(device=dev).manual_seed(1234 + rank)
    Pm = torch.randn(F, C, generator=torch.Generator().manual_seed(7)).to(dev) / np.sqrt(C)
    pool = []
    for i in range(4):
        X = torch.randn(B, C, T, generator=g, device=dev).clamp_(-20, 20)
        Y = torch.einsum("fc,bct->bft", Pm, X) + 0.5 * torch.randn(B, F, T, generator=g, device=dev)
        pool.append((X, Y.contiguous()))
    # a fifth batch the steps never train on: the retrieval accuracy reported below is measured on IT (eval mode), i.e. on
    # unseen segments — the quality half of BASELINE's metric — not on the batches the model has just been fitted to
    Xh = torch.randn(B, C, T, generator=g, device=dev).clamp_(-20, 20)
    Yh = (torch.einsum("fc,bct->bft", Pm, Xh) + 0.5 * torch.randn(B, F, T, generator=g, device=dev)).contiguous()
    subj_rng = np.random.RandomState(100 + rank)
    subj_h = torch.from_numpy(np.random.RandomState(900 + rank).randint(0, S, size=B).astype(np.int32))

    ranks_acc = []
    one = torch.ones((), dtype=torch.float32, device=dev)

    # Under data parallelism the speech rows of the NEXT batch are packed and all-gathered (197 MB per rank and step at
    # config 3) while THIS step's backward runs, like a data loader that is one batch ahead: five milliseconds of cover
    # instead of the forward's two and a half.  The loss keeps two packed buffers in rotation for exactly this overlap.
    ahead = dp and not a.no_prefetch_ahead
    primed = [False]

    def step(i, host_sync=False, batch=None, after_forward=None):
        if batch is None:
            X, Y = pool[i % len(pool)]
            subj = torch.from_numpy(subj_rng.randint(0, S, size=B).astype(np.int32))
        else:                                                # the feed leg: (X, Y, subject indices) from the input path
            X, Y, subj = batch
        if not (ahead and primed[0]) or batch is not None:
            lossf.prefetch(Y, enc.compute_dtype)             # pack Y (+ all-gather it under DP) while the encoder runs
        Z = enc(X, subj)
        loss = lossf(Y, Z)
        cnt = sda_loss.retrieval_ranks(Y, Z)                 # Classifier semantics (train.py:193-194)
        if ahead and batch is None:
            lossf.prefetch(pool[(i + 1) % len(pool)][1], enc.compute_dtype)
            primed[0] = True
        if after_forward is not None:
            after_forward()
        if host_sync:                                        # what train.py does every step: loss.item() + top-k on the host
            float(loss.detach())
            cnt = cnt.cpu()
        ranks_acc.append(cnt)
        opt.zero_grad(set_to_none=True)
        scaler.scale(loss).backward(gradient=one)            # (a resident d loss / d loss = 1: autograd would fill a fresh one per step)
        scaler.unscale_(params)
        if dp:             # encoder gradients were all-reduced inside backward (overlapped); temp is left
            allreduce_gradients(list(lossf.parameters()) if enc.grads_are_reduced else params)
        opt.step()
        return loss

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    host_enqueue = [0.0]

    def timed(n, first, **kw):
        fence()
        t0 = time.perf_counter()
        for i in range(n):
            loss = step(first + i, **kw)
        host_enqueue[0] = (time.perf_counter() - t0) / max(1, n)     # the host's share: it runs ahead of the GPU when this < the step
        fence()
        dt = time.perf_counter() - t0
        tmax = torch.tensor([dt], device=dev)
        if world > 1:
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        return float(tmax.item()), loss

    settle = min(4, max(0, a.warmup - 1))                    # warm-up steps kept for AFTER the heap is collected (below)
    for i in range(a.warmup - settle):
        step(i)
    # CPython's cyclic collector runs a FULL (generation-2) pass once the objects a freshly built model and its first steps
    # allocated cross its threshold — measured at step ~23 of this loop, 87 ms of host time, then five slower steps
    # (tools/step_warmup.py): 3 ms per step if it lands inside a 20-step timed region.  A long-running training job pays it
    # once per many thousand steps; here the heap is collected and frozen after warm-up (the training driver does the same
    # after its first steps), so the timed region measures steps, not the interpreter's housekeeping.
    # The collection itself leaves the interpreter cold (the next step takes the host 7 ms to enqueue instead of 4.6, and the
    # four after it are 0.1-0.4 ms slow on the GPU side: tools/step_series.py N SETTLE): it runs after the FIRST warm-up step —
    # by then the model, the workspaces and every cached launch descriptor exist — and up to four of the W warm-up steps follow it.
    import gc
    gc.collect()
    gc.freeze()
    for i in range(a.warmup - settle, a.warmup):
        step(i)
    ranks_acc.clear()
    dt, loss = timed(a.steps, a.warmup)                      # the contract number: K clean steps
    host_main = host_enqueue[0]
    cnt = torch.cat([c.to(dev) for c in ranks_acc[-3:]]).float()
    top10_train = float((cnt < 10).float().mean())           # on the batches the last three steps trained on (for reference)
    final_loss = float(loss.detach())
    # held-out retrieval accuracy (Classifier semantics, models.py:208-248; global candidates under data parallelism), outside
    # the timed region: eval-mode forward of the unseen batch
    enc.eval()
    with torch.no_grad():
        cnt_h = sda_loss.retrieval_ranks(Yh, enc(Xh, subj_h)).float()
    enc.train()
    top10 = float((cnt_h < 10).float().mean())
    top1 = float((cnt_h < 1).float().mean())
    if world > 1:
        acc = torch.tensor([top10, top1], device=dev)
        dist.all_reduce(acc)
        top10, top1 = float(acc[0]) / world, float(acc[1]) / world
    # (fp16 runs with a STATIC loss scale here — no per-step host check inside the timed region; train.py checks every step.
    # An overflow would have poisoned the weights: verify after the fact that it did not)
    if not all(bool(torch.isfinite(torch.view_as_real(p) if p.is_complex() else p).all()) for p in params):
        raise SystemExit("bench.py: non-finite parameters after the timed region (fp16 overflow?) - the number is void")

    nxt = a.warmup + a.steps                                 # step indices stay contiguous across the legs: under N > 1 the
    dt_sync = None                                           # speech rows of step i + 1 were prefetched by step i
    if not a.no_host_sync_leg:                               # same step with the reference loop's per-step host readbacks
        n_sync = max(3, min(10, a.steps))
        dt_sync, _ = timed(n_sync, nxt, host_sync=True)
        dt_sync /= n_sync
        nxt += n_sync

    # The input path inside a timed loop (gwilliams2022.py:129-142,640-661, get_dataloaders.py:48-87): the
    # recordings resident in HBM, the reference's RandomSampler(replacement=True) cut into rank shards, and every step's batch
    # made by ONE kernel (window gather + baseline correction + robust scaling + clamp) plus the speech rows' gather from the
    # resident embedding table — on a stream of its own, one batch ahead of the step that consumes it, like a data loader.
    dt_feed = None
    if a.config == 2 and world == 1 and emu == 1 and not a.no_feed_leg:
        from speech_decoding_amd.data import ShardedRandomSampler, synthetic_resident_dataset
        n_feed = max(3, min(20, a.steps))
        # (as in the real dataset every task is heard by many subjects: ceil(S / 4) recordings per task, so that a batch spans
        # all S subjects like the pool's uniformly drawn indices — with two recordings per task only 8 of the 27 subjects
        # ever occur and the per-subject weight gradient of the SubjectBlock runs on a third of its workgroups)
        feed, train_idx, _ = synthetic_resident_dataset(cfg, dev, n_segments=4 * B, seed=1234, recs_per_task=int(os.environ.get("SDA_FEED_RECS", -(-S // 4))))
        feed.pack_embeddings(enc.compute_dtype)              # the embedding table resident in row layout: Y arrives packed
        sampler = ShardedRandomSampler(len(train_idx), B, n_feed + 5, rank, world, seed=4321)
        # The feed's stream has LOW priority (the step's chain is the critical path: a freed CU slot goes to it first) and starts
        # batch i + 1 behind step i's loss forward, i.e. beside the backward pass: the forward's k = 3 convs are ONE round of
        # persistent workgroups sized to the whole chip, and a slot taken from them by another stream's kernel is a straggler.
        # (SDA_FEED_AT=start / SDA_FEED_PRIO=0: diagnostics — the batch made beside the forward / at normal priority)
        feed_at = os.environ.get("SDA_FEED_AT", "backward")
        feed_prio = int(os.environ.get("SDA_FEED_PRIO", "1"))
        feed_stream = (torch.cuda.ExternalStream(ops.stream_create_priority(feed_prio), device=dev) if feed_prio > 0
                       else torch.cuda.Stream(device=dev, priority=feed_prio))
        main_stream = torch.cuda.current_stream(dev)
        it = iter(sampler)

        def produce(behind_main=False):
            idx = next(it, None)
            if idx is None:
                return None
            if feed_at == "inline":                          # on the step's own stream, in program order (diagnostic)
                Xf, Yf, sf = feed.batch(train_idx[idx.numpy()])
                ev = torch.cuda.Event()
                ev.record(main_stream)
                return Xf, Yf, sf, ev
            if behind_main:
                gate = torch.cuda.Event()
                gate.record(main_stream)
                feed_stream.wait_event(gate)
            with torch.cuda.stream(feed_stream):
                Xf, Yf, sf = feed.batch(train_idx[idx.numpy()])
                ev = torch.cuda.Event()
                ev.record(feed_stream)
            return Xf, Yf, sf, ev

        feed_dbg = os.environ.get("SDA_FEED_DEBUG", "")    # "discard": make every batch but train on the pool (diagnostic)

        def feed_step(i, nxt_batch):
            Xf, Yf, sf, ev = nxt_batch
            if feed_dbg == "discard":
                main_stream.wait_event(ev)
                nb = produce()
                step(i)
                return nb
            main_stream.wait_event(ev)
            Xf.record_stream(main_stream)
            Yf.record_stream(main_stream)
            box = []
            if feed_at == "start":
                box.append(produce())                        # batch i + 1 is made while step i runs, from its start
                step(i, batch=(Xf, Yf, sf))
            else:
                step(i, batch=(Xf, Yf, sf), after_forward=lambda: box.append(produce(behind_main=True)))
            return box[0]

        primed[0] = False
        pending = produce()
        n_warm = 4                                           # untimed: the feed's streams exist, the allocator has its blocks
        for i in range(n_warm):
            pending = feed_step(nxt + i, pending)
        fence()
        t0 = time.perf_counter()
        for i in range(n_feed):
            pending = feed_step(nxt + n_warm + i, pending)
        host_feed = (time.perf_counter() - t0) / n_feed      # the host's enqueue time per step (it runs ahead of the GPU)
        fence()
        dt_feed = (time.perf_counter() - t0) / n_feed
        nxt += n_warm + n_feed
        del feed, pending

    timer = None
    if not a.no_kernel_timer and a.timer_steps > 0:          # roofline leg, outside the timed region
        timer = ops.KernelTimer(("conv_gemm", "wgrad_gemm"))
        fence()
        ops.TIMER = timer
        for i in range(a.timer_steps):
            step(nxt + i)
        fence()
        ops.TIMER = None

    coll = None
    if dp and a.coll_timer_steps > 0:                        # what each collective costs the stream that waits for it
        from speech_decoding_amd.distributed import CollectiveTimer
        fence()
        ct = CollectiveTimer().install()
        base = nxt + (a.timer_steps if timer is not None else 0)
        for i in range(a.coll_timer_steps):
            step(base + i)
        fence()
        ct.uninstall()
        coll = {k: {"calls_per_step": round(n / a.coll_timer_steps, 2), "us_per_step": round(us / a.coll_timer_steps, 1)}
                for k, (n, us) in sorted(ct.summary().items())}

    if rank == 0:
        out = {
            "metric": "train segments/sec (Gwilliams2022 208ch x 360, top-10 retrieval acc alongside)",
            "value": round(B * world * a.steps / dt, 2), "unit": "segments/s", "n_gpus": world, "steps": a.steps,
            "warmup": a.warmup, "ms_per_step": round(1e3 * dt / a.steps, 3), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": a.dtype, "data": "synthetic",
            "config": {"workload": (f"{'Gwilliams2022-shaped MEG' if C != 60 else 'Brennan2018-shaped EEG'}: {C} ch x {T} samples, "
                                    f"{S} subjects, F={F}, batch {B}/GPU, fwd+CLIP loss+top-k+bwd+Adam (BASELINE {shape['name']}"
                                    f"{' / configs[2]' if world == 8 and a.config == 2 else ''})"
                                    + (f"; PER-RANK step of a {emu}-rank job on ONE GPU: {B * emu} speech rows of which {B * (emu - 1)} are "
                                       f"resident stand-ins for the all-gather's delivery"
                                       f"{' (re-delivered by a device copy every step)' if not a.emulate_no_copy else ''}, logits block "
                                       f"{B * emu} x {B}, SyncBN all-reduces / loss-statistics gather / gradient buckets issued through "
                                       f"RCCL at world size 1 (BASELINE configs[2] per-rank step when N = 8: a compute-side bound, not a scaling curve)"
                                       if emu > 1 else "")),
                       "global_batch": B * world * emu, "seq_len": T, "parallelism": f"dp{world}" + (f" (emulating dp{emu})" if emu > 1 else "")},
            "top10_acc": round(top10, 4), "top1_acc": round(top1, 4),
            "top10_note": f"held-out batch of {B * world} segments never trained on (eval mode, chance = {10.0 / (B * world * emu):.4f}); "
                          f"on the last three TRAINING batches: {top10_train:.4f}",
            "final_loss": round(final_loss, 4),
            "host_enqueue_ms_per_step": round(1e3 * host_main, 3),      # < ms_per_step: the host keeps ahead, the GPU is the limit
        }
        if dt_sync is not None:
            out["host_synced"] = {"value": round(B * world / dt_sync, 2), "ms_per_step": round(1e3 * dt_sync, 3),
                                  "note": "same step with loss.item() and the ranks read back on the host every step (train.py:194-196)"}
        if dt_feed is not None:
            out["with_feed"] = {"value": round(B * world / dt_feed, 2), "ms_per_step": round(1e3 * dt_feed, 3),
                                "host_enqueue_ms_per_step": round(1e3 * host_feed, 3),
                                "note": "same step fed by ResidentSegmentFeed + ShardedRandomSampler every step: window gather + baseline "
                                        "correction + robust scaling + clamp (one kernel) and the speech rows' gather, on a stream of "
                                        "their own one batch ahead (gwilliams2022.py:129-142,640-661; get_dataloaders.py:48-87)"}
        if timer is not None:
            summ = timer.summary()
            key = max(summ, key=lambda k: summ[k][2])             # dominant = most total time
            n, flops, ms = summ[key]
            ach = flops / (ms * 1e-3) / 1e12
            # (the committed PMC passes are of the headline configuration only)
            traffic, src = pmc_traffic(a.dtype, key[2], key[3], key[0]) if (a.config == 2 and emu == 1 and B == 256) else (None, None)
            out["roofline"] = {"bound": "mfma", "achieved": round(ach, 2), "peak": PEAK_TFLOPS[a.dtype], "unit": "TFLOP/s",
                               "frac": round(ach / PEAK_TFLOPS[a.dtype], 4), "traffic": traffic, "traffic_source": src,
                               "kernel": f"{key[0]}<{key[1]},TILE={key[2]},KS={key[3]}>", "launches": n,
                               "avg_us": round(1e3 * ms / n, 2),
                               "measured": f"HIP events around every launch in {a.timer_steps} extra steps after the timed region"}
            out["kernel_time_ms_per_step"] = {f"{k[0]}<{k[1]},{k[2]},{k[3]}>": round(v[2] / a.timer_steps, 3) for k, v in summ.items()}
            out["kernel_tflops"] = {f"{k[0]}<{k[1]},{k[2]},{k[3]}>": round(v[1] / (v[2] * 1e-3) / 1e12, 1) for k, v in summ.items() if v[2] > 0}
        if coll is not None:
            out["collectives_us_per_step"] = coll
        if world == 1 and emu == 1 and a.config == 2 and not a.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(B)
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    if dp:
        from speech_decoding_amd.distributed import shutdown
        lossf.drain()                # the speech rows gathered one batch ahead that no step will consume
        shutdown()
    torch.cuda.synchronize()
    ops.stream_destroy_all()         # streams made through the C ABI (the feed leg's low-priority one): nothing is queued any more


if __name__ == "__main__":
    main()
