#!/usr/bin/env python3
"""Training / evaluation driver for the MI355X build — same CLI surface and printed scalars as the
reference's `train.py`, own implementation.

    python train.py                          # configs/config.yaml
    python train.py dataset=Brennan2018 batch_size=64 epochs=3 compute_dtype=bf16
    python -m torch.distributed.run --nproc-per-node 8 train.py batch_size=256      # data parallel

What is kept from the reference loop (train.py:148-259):
  * models: BrainEncoder + Classifier + CLIPLoss, Adam over encoder parameters and the loss temperature;
  * per batch: Z = encoder(X, subject_idxs); loss = CLIPLoss(Y, Z); top-1/top-10 from Classifier(Z, Y);
  * update cadence: Gwilliams2022 steps on every batch, Brennan2018 ONCE per epoch with the last batch's
    loss (train.py:200-209);
  * evaluation in eval() mode on the whole test split as ONE batch (train.py:99,211-233);
  * the per-epoch print line and the optional W&B scalar names; `model_last.pt` = encoder.state_dict().
What is not: the M/EEG + wav2vec2 dataset classes (out of scope, they need MNE, the raw recordings and
un-downloadable weights).  `--data synthetic` (default) builds a seeded stand-in with the same tensor
shapes; a real dataset object can be passed to `run()` as (train_batches, test_batch) callables.
"""
from __future__ import annotations

import os
import sys
import time
from typing import Callable, Iterable, Optional, Tuple

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from speech_decoding.models import BrainEncoder, Classifier       # noqa: E402  (train.py:22 import path)
from speech_decoding.utils.loss import CLIPLoss                    # noqa: E402  (train.py:24)
from speech_decoding_amd import load_config                        # noqa: E402
from speech_decoding_amd.distributed import allreduce_gradients, broadcast_parameters, shard_range  # noqa: E402


class SyntheticSegments:
    """Seeded stand-in for the segmented datasets: X (N, C, T) ~ N(0,1) clamped to ±clamp_lim with the first
    baseline samples' mean removed (preproc_utils.py:69-90,128-142), Y = P·X + 0.5·eps (so retrieval is
    learnable), subject indices uniform.  Split by `split_ratio` like the shallow split."""

    def __init__(self, args, n_segments: int, device, seed: int = 0):
        C = int(args.get("num_channels", 208 if args.dataset == "Gwilliams2022" else 60))
        T = int(args.preprocs["seq_len_sec"] * args.preprocs["brain_resample_rate"])
        F = 1024 if args.preprocs["last4layers"] else int(args.F)
        g = torch.Generator().manual_seed(seed)
        X = torch.randn(n_segments, C, T, generator=g)
        nb = int(args.preprocs["baseline_len_sec"] * args.preprocs["brain_resample_rate"])
        X = X - X[:, :, :nb].mean(dim=-1, keepdim=True)
        if args.preprocs["clamp"]:
            X = X.clamp(-float(args.preprocs["clamp_lim"]), float(args.preprocs["clamp_lim"]))
        P = torch.randn(F, C, generator=g) / np.sqrt(C)
        Y = torch.einsum("fc,nct->nft", P, X) + 0.5 * torch.randn(n_segments, F, T, generator=g)
        self.X, self.Y = X.to(device), Y.to(device)
        self.subj = torch.randint(0, int(args.num_subjects), (n_segments,), generator=g, dtype=torch.int32)
        n_train = int(n_segments * float(args.split_ratio))
        self.train_idx = torch.arange(n_train)
        self.test_idx = torch.arange(n_train, n_segments)
        self.gen = torch.Generator().manual_seed(seed + 1)

    def train_batches(self, batch_size: int, updates: int) -> Iterable[Tuple[torch.Tensor, torch.Tensor, torch.Tensor]]:
        """`updates` batches; like RandomSampler(replacement=True) over the train split (get_dataloaders.py:48-87)
        but without duplicates inside a batch (train.py:181-183 aborts on duplicate segments)."""
        for _ in range(updates):
            pick = self.train_idx[torch.randperm(len(self.train_idx), generator=self.gen)[:batch_size]]
            yield self.X[pick], self.Y[pick], self.subj[pick]

    def test_batch(self):
        i = self.test_idx
        return self.X[i], self.Y[i], self.subj[i]


def run(args, train_batches: Optional[Callable] = None, test_batch: Optional[Callable] = None, log=print):
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    from speech_decoding_amd.streams import use_training_stream
    caller_stream = torch.cuda.current_stream(device)
    use_training_stream(device)          # the loop's chain on a high-priority stream; the engine's side streams stay normal
    owns_group = world > 1 and not dist.is_initialized()
    if owns_group:
        from speech_decoding_amd.distributed import init_process_group as sda_init_pg
        sda_init_pg(os.environ.get("SDA_DIST_BACKEND", "nccl"))
    if args.get("reproducible", False):
        np.random.seed(0)
        torch.manual_seed(0)
    elif world > 1:
        # SpatialDropout's centre comes from NumPy's global generator (models.py:81) and must be the same on every rank
        from speech_decoding_amd.distributed import seed_numpy_all_ranks
        seed_numpy_all_ranks()

    # data=pool (default): a resident pool of ready-made segments, every rank slicing its shard out of the global batch;
    # data=resident: the reference's own input path on the GPU — recordings resident in HBM, RandomSampler(replacement=True)
    # (get_dataloaders.py:48-87) cut into rank shards so that no rank materialises another rank's samples, window gather +
    # baseline correction + robust scaling + clamp (gwilliams2022.py:129-142,640-661) as one kernel per batch
    feeds_local_shards = False
    if train_batches is None and str(args.get("data", "pool")) == "resident":
        from speech_decoding_amd.data import ShardedRandomSampler, synthetic_resident_dataset
        n_seg = int(args.get("synthetic_segments", 4 * int(args.batch_size)))
        feed, train_idx, test_idx = synthetic_resident_dataset(args, device, n_segments=n_seg, seed=1234)
        updates = int(args.get("updates_per_epoch", max(1, len(train_idx) // int(args.batch_size))))
        epoch_no = [0]
        pack_resident_embeddings = feed          # (packed once the encoder's compute dtype is known, below)

        def train_batches():
            sampler = ShardedRandomSampler(len(train_idx), int(args.batch_size), updates, rank, world, seed=4321 + epoch_no[0])
            epoch_no[0] += 1
            # (under data parallelism the recordings of the whole global batch are drawn on every rank alike and sliced)
            yield from feed.batches(sampler, index_map=train_idx)

        def test_batch():
            lo, hi = shard_range(len(test_idx), rank, world)
            return feed.batch(test_idx[lo:hi])
        feeds_local_shards = True
    elif train_batches is None:
        n_seg = int(args.get("synthetic_segments", 4 * int(args.batch_size)))
        data = SyntheticSegments(args, n_seg, device, seed=1234)
        updates = int(args.get("updates_per_epoch", max(1, len(data.train_idx) // int(args.batch_size))))
        train_batches = lambda: data.train_batches(int(args.batch_size), updates)      # noqa: E731
        test_batch = data.test_batch

    brain_encoder = BrainEncoder(args).to(device)
    if feeds_local_shards and "pack_resident_embeddings" in locals():
        pack_resident_embeddings.pack_embeddings(brain_encoder.compute_dtype)      # Y batches arrive as the loss's packed operand
    classifier = Classifier(args)
    loss_func = CLIPLoss(args).to(device)
    loss_func.train()
    broadcast_parameters(brain_encoder)
    params = list(brain_encoder.parameters()) + list(loss_func.parameters())
    optimizer = torch.optim.Adam(params, lr=float(args.lr))
    wandb = None
    if args.get("use_wandb", False) and rank == 0:
        import wandb as _wandb
        wandb = _wandb
        wandb.init(project=args.wandb.project, entity=args.wandb.entity, config=dict(args), save_code=True)
        wandb.run.name = args.wandb.run_name + "_" + str(args.split_mode)

    from speech_decoding_amd.amp import LossScaler
    seq_T = int(args.preprocs["seq_len_sec"] * args.preprocs["brain_resample_rate"])
    scaler = LossScaler.for_dtype(brain_encoder.compute_dtype, float(args.get("fp16_loss_scale", 1024.0)),
                                  global_batch=int(args.batch_size), T=seq_T)                      # no-op unless fp16

    fp16 = scaler.enabled            # (not `scale != 1`: a scale halved down to 1, or fp16_loss_scale=1, keeps the overflow guard)
    one = torch.ones((), dtype=torch.float32, device=device)       # d loss / d loss, resident (autograd would fill one per step)

    def backward_and_step(loss):
        optimizer.zero_grad()
        scaler.scale(loss).backward(gradient=one)
        if world > 1:        # (SUM of the still-scaled gradients: every rank then sees the same overflow, or none)
            allreduce_gradients(list(loss_func.parameters()) if brain_encoder.grads_are_reduced else params)
        # fp16 only: an activation gradient that overflowed to inf / NaN must not reach Adam — the finiteness check is a
        # host read-back per step of ONE device flag written by one fused launch (fp16 is configs[4]'s dtype; bf16 / fp32
        # never take it), the step is skipped and the scale halved (amp.LossScaler.update, GradScaler's rule).  A skipped
        # step is not a no-op for BatchNorm: its forward has already moved the running statistics and num_batches_tracked
        # (as it would under torch.cuda.amp with nn.BatchNorm1d) — harmless for training, visible in a bit-for-bit replay
        ok = scaler.unscale_(params, check=fp16)
        scaler.update(ok)
        if ok:
            optimizer.step()

    history = []
    for epoch in range(int(args.epochs)):
        if epoch == 1:
            # everything allocated so far (model, optimizer state, workspaces) is permanent: keep CPython's full
            # collections from walking it again and again (one costs ~90 ms of host time against an 8 ms step)
            import gc
            gc.collect()
            gc.freeze()
        tr_loss, tr_ranks = [], []
        brain_encoder.train()
        loss = None
        def local_shard(batch):
            X, Y, subject_idxs = batch
            if world > 1 and not feeds_local_shards:
                lo, hi = shard_range(X.shape[0], rank, world)
                X, Y, subject_idxs = X[lo:hi], Y[lo:hi], subject_idxs[lo:hi]
            return X, Y, subject_idxs

        # The speech side of the loss does not depend on the encoder: it is started ONE BATCH AHEAD, like a data loader —
        # batch n + 1's rows are packed (and, under data parallelism, all-gathered: 197 MB per rank at config 3) while batch
        # n's backward runs, five milliseconds of cover instead of the forward's two and a half.  CLIPLoss keeps two packed
        # buffers in rotation for exactly this; only the first batch of an epoch is prefetched at its own start.
        batches = (local_shard(b) for b in train_batches())       # (sharded ONCE per batch: the prefetch is matched by tensor identity)
        ahead = next(batches, None)
        first = True
        while ahead is not None:
            X, Y, subject_idxs = ahead
            ahead = next(batches, None)
            if first:
                loss_func.prefetch(Y, brain_encoder.compute_dtype)
                first = False
            Z = brain_encoder(X, subject_idxs)
            loss = loss_func(Y, Z)
            # The reference reads loss.item() and the two accuracies back on the host here, every batch (train.py:194-198):
            # a host<->GPU synchronisation in the middle of the step.  The same numbers are kept on the device and read
            # once per epoch, below: the host keeps enqueueing while the GPU works (8 vs 16 ms per step at batch 256).
            with torch.no_grad():
                ranks = classifier.ranks(Z, Y)                   # Classifier.forward's ranks, kept on the device
            tr_loss.append(loss.detach())
            tr_ranks.append(ranks)
            if ahead is not None:
                loss_func.prefetch(ahead[1], brain_encoder.compute_dtype)
            if args.dataset == "Gwilliams2022":
                backward_and_step(loss)
        if args.dataset == "Brennan2018" and loss is not None:          # once per epoch, last batch only
            backward_and_step(loss)
        tr_loss = [float(l) for l in tr_loss]
        tr_top1 = [float((r == 0).float().mean()) for r in tr_ranks]
        tr_top10 = [float((r < 10).float().mean()) for r in tr_ranks]

        brain_encoder.eval()
        te_loss, te_top1, te_top10 = [], [], []
        with torch.no_grad():
            X, Y, subject_idxs = test_batch()
            if world > 1 and not feeds_local_shards:
                lo, hi = shard_range(X.shape[0], rank, world)
                X, Y, subject_idxs = X[lo:hi], Y[lo:hi], subject_idxs[lo:hi]
            Z = brain_encoder(X, subject_idxs)
            te_loss.append(loss_func(Y, Z).item())
            t1, t10 = classifier(Z, Y, test=True)
            te_top1.append(t1)
            te_top10.append(t10)

        row = {"epoch": epoch, "train_loss": np.mean(tr_loss), "test_loss": np.mean(te_loss),
               "trainTop1acc": np.mean(tr_top1), "trainTop10acc": np.mean(tr_top10),
               "testTop1acc": np.mean(te_top1), "testTop10acc": np.mean(te_top10),
               "lrate": optimizer.param_groups[0]["lr"], "temp": loss_func.temp.item()}
        history.append(row)
        if rank == 0:
            log(f"Ep {epoch}/{args.epochs} | ", f"train l: {row['train_loss']:.3f} | ", f"test l: {row['test_loss']:.3f} | ",
                f"trainTop10acc: {row['trainTop10acc']:.3f} | ", f"testTop10acc: {row['testTop10acc']:.3f} | ",
                f"lr: {row['lrate']:.5f}", f"temp: {row['temp']:.3f}")
            if wandb is not None:
                wandb.log(row)
            torch.save(brain_encoder.state_dict(), "model_last.pt")
    if owns_group:                       # a group this function created is also ended here, in order (distributed.shutdown)
        from speech_decoding_amd.distributed import shutdown
        loss_func.drain()
        shutdown()
    # hand the thread back on the stream it came with (everything queued on the training stream is waited for first)
    caller_stream.wait_stream(torch.cuda.current_stream(device))
    torch.cuda.set_stream(caller_stream)
    return history, brain_encoder, loss_func


def main(argv):
    overrides = [a for a in argv if "=" in a and not a.startswith("--")]
    path = next((a.split("=", 1)[1] for a in argv if a.startswith("--config=")), None)
    args = load_config(path, overrides)
    args.setdefault("root_dir", os.getcwd())                       # injected by the reference via open_dict
    args.setdefault("num_subjects", 27 if args.dataset == "Gwilliams2022" else 33)
    t0 = time.time()
    run(args)
    print(f"done in {time.time() - t0:.1f} s")


if __name__ == "__main__":
    main(sys.argv[1:])
