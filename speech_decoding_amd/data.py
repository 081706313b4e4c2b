"""Per-rank data feed for the training step: the reference's sampler cut into rank shards, and the segment gather +
batch collate on the GPU in the loop.

What it mirrors (SeanNobel/speech-decoding):
  * utils/get_dataloaders.py:48-87 (`get_samplers`): `RandomSampler(train_set, replacement=True, num_samples=updates *
    batch_size, generator=g)` consumed in batches of `batch_size` — here the SAME index stream is generated on every rank
    from the same seeded generator and rank r keeps positions [r * B_local, (r + 1) * B_local) of each global batch
    (`ShardedRandomSampler`): no rank ever materialises another rank's samples, and the union over ranks is exactly the
    single-process batch;
  * dataclass/gwilliams2022.py:129-142 (`__getitem__`: a speech segment i -> a random recording of that task, the MEG
    window at the segment's onset, the subject index) and :640-661 (`Gwilliams2022Collator`: baseline correction, robust
    scaling, clamp) — here `ResidentSegmentFeed`: the recordings stay in HBM, a batch is (recording, onset) pairs turned
    into X (B, C, T) by ONE kernel (collate.ResidentSegments), Y rows are gathered from the resident embedding table.
"""
from __future__ import annotations

from typing import Iterator, List, Sequence, Tuple

import numpy as np
import torch

from .collate import ResidentSegments


class ShardedRandomSampler:
    """Indices of `updates` global batches of `batch_size` drawn with replacement from range(n) (get_dataloaders.py:57-62),
    rank `rank` of `world` keeping its contiguous slice of every global batch.  Every rank runs the same generator, so the
    global batch is well defined without any communication."""

    def __init__(self, n: int, batch_size: int, updates: int, rank: int = 0, world: int = 1, seed: int = 0, replacement: bool = True):
        if batch_size % world:
            raise ValueError("the global batch must divide evenly over the ranks")
        self.n, self.batch_size, self.updates, self.rank, self.world = int(n), int(batch_size), int(updates), int(rank), int(world)
        self.replacement = bool(replacement)
        self.gen = torch.Generator().manual_seed(int(seed))

    def global_batches(self) -> Iterator[torch.Tensor]:
        for _ in range(self.updates):
            if self.replacement:
                yield torch.randint(0, self.n, (self.batch_size,), generator=self.gen)
            else:
                yield torch.randperm(self.n, generator=self.gen)[: self.batch_size]

    def __iter__(self) -> Iterator[torch.Tensor]:
        per = self.batch_size // self.world
        for idx in self.global_batches():
            yield idx[self.rank * per: (self.rank + 1) * per]

    def __len__(self) -> int:
        return self.updates


class ResidentSegmentFeed:
    """recordings: list of (C, L_r) device tensors, one per (subject, session) recording, each with the subject index
    `rec_subject[r]` and the task `rec_task[r]` it recorded; onsets[r][j] = first MEG sample of the j-th speech segment of
    that task; seg_task[i], seg_in_task[i] = (task, position) of global speech segment i (gwilliams2022.py: `segment_to_task`);
    Y (N, F, T) speech embeddings, resident.  batch(idx) -> (X, Y, subject_idxs) with X collated on the device."""

    def __init__(self, recordings: Sequence[torch.Tensor], rec_subject: Sequence[int], rec_task: Sequence[int],
                 onsets: Sequence[np.ndarray], seg_task: np.ndarray, seg_in_task: np.ndarray, Y: torch.Tensor, *,
                 seq_len_samp: int, baseline_len_samp: int, clamp_lim: float, clamp: bool = True, seed: int = 0):
        self.rs = ResidentSegments(list(recordings), seq_len_samp, baseline_len_samp, clamp_lim, clamp)
        self.rec_subject = np.asarray(rec_subject, dtype=np.int32)
        self.rec_task = np.asarray(rec_task)
        self.onsets = [np.asarray(o, dtype=np.int64) for o in onsets]
        self.seg_task, self.seg_in_task = np.asarray(seg_task), np.asarray(seg_in_task)
        self.Y = Y
        self._Yt = None           # row-layout table in the compute dtype (pack_embeddings)
        self.by_task = {int(t): np.nonzero(self.rec_task == t)[0] for t in np.unique(self.rec_task)}
        # (gwilliams2022.py:133 draws the recording from NumPy's GLOBAL generator inside DataLoader workers; the global
        # generator stays reserved for SpatialDropout's centre here — every rank must draw that one in lockstep)
        self.rng = np.random.RandomState(seed)
        # dense tables for the vectorised draw / onset lookup (None where the structure is ragged: the loops remain)
        self._task_keys = np.array(sorted(self.by_task), dtype=self.rec_task.dtype)
        counts = {len(v) for v in self.by_task.values()}
        self._rec_table = np.stack([self.by_task[int(t)] for t in self._task_keys]) if len(counts) == 1 else None
        lens = {len(o) for o in self.onsets}
        self._onset_table = np.stack(self.onsets) if len(lens) == 1 else None

    def __len__(self) -> int:
        return int(self.Y.shape[0])

    def pack_embeddings(self, dtype: torch.dtype, chunk: int = 128):
        """Keep the speech embeddings resident as a row-layout table in `dtype` (the encoder's compute dtype): batch() then
        hands out Y as a zero-copy (B, F, T) view of ONE gather kernel's output, which CLIPLoss consumes as its packed operand —
        per step 0.4 GB of HBM traffic instead of the 1.3 GB of index_select + sda_pack_rows (batch 256, bf16).  Values: each
        embedding is rounded to `dtype` once here instead of once per step in the loss's pack — the same numbers."""
        from . import lib as L
        from . import ops
        N, F, T = self.Y.shape
        Tp, Fp = L.rows_tp(T), L.pad_channels(F)
        table = torch.zeros((N * Tp + L.rows_alloc(1, T) - Tp, Fp), dtype=dtype, device=self.Y.device)
        for k in range(0, N, chunk):
            n = min(chunk, N - k)
            ops.pack_rows(self.Y[k: k + n], table[k * Tp:])
        self._Yt = table
        return self

    def draw_recordings(self, idx) -> np.ndarray:
        """gwilliams2022.py:133: one random recording of each segment's task, drawn item by item from the feed's generator
        (`rng.choice(recordings of the task)`).  When every task has the same number of recordings the whole batch is ONE
        vectorised draw — the same stream of values as the item-by-item calls (checked in tests/test_host_cpu.py), without
        256 Python-level generator calls in front of a 7 ms step."""
        ii = np.asarray(idx, dtype=np.int64)
        tasks = self.seg_task[ii]
        if self._rec_table is not None:
            u = self.rng.randint(0, self._rec_table.shape[1], size=len(ii))
            return self._rec_table[np.searchsorted(self._task_keys, tasks), u].astype(np.int64)
        return np.array([self.rng.choice(self.by_task[int(t)]) for t in tasks], dtype=np.int64)

    def batch(self, idx, rec=None) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
        """idx: this rank's global segment indices; rec: their recordings when the caller drew them (data parallelism: the
        draws are made for the GLOBAL batch on every rank alike and sliced, see batches())."""
        idx = torch.as_tensor(idx, dtype=torch.int64)
        ii = idx.numpy()
        rec = self.draw_recordings(ii) if rec is None else np.asarray(rec, dtype=np.int64)
        on = self._onset_of(rec, self.seg_in_task[ii])
        X = self.rs.batch(rec, on)
        if self._Yt is not None:
            # embeddings resident in ROW LAYOUT in the compute dtype (pack_embeddings): the gathered batch IS the loss's packed
            # operand — CLIPLoss recognises the view and neither copies nor re-packs it
            from . import ops
            B, (F, T) = len(ii), self.Y.shape[1:]
            with torch.cuda.device(self._Yt.device):
                Y = ops.rows_view(ops.gather_samples(self._Yt, ops.upload_small(ii, self._Yt.device), B, T), B, F, T)
        elif self.Y.is_cuda:
            from . import ops
            with torch.cuda.device(self.Y.device):
                Y = self.Y.index_select(0, ops.upload_small(ii, self.Y.device))    # (index table in kernel arguments: no host wait)
        else:
            Y = self.Y.index_select(0, idx)
        return X, Y, torch.from_numpy(self.rec_subject[rec].astype(np.int32))

    def _onset_of(self, rec: np.ndarray, j: np.ndarray) -> np.ndarray:
        if self._onset_table is not None:
            return self._onset_table[rec, j]
        return np.array([self.onsets[r][k] for r, k in zip(rec, j)], dtype=np.int64)

    def batches(self, sampler, index_map=None) -> Iterator[Tuple[torch.Tensor, torch.Tensor, torch.Tensor]]:
        """One (X, Y, subject_idxs) per batch of `sampler`.  With a ShardedRandomSampler the recording of EVERY segment of the
        global batch is drawn on every rank from the same generator state and the rank keeps its slice — the union over the
        ranks is exactly the batch a single process would have drawn, and no two ranks share a draw.  `index_map`: optional
        array mapping sampler indices to segment indices (a train split)."""
        if hasattr(sampler, "global_batches") and getattr(sampler, "world", 1) > 1:
            per = sampler.batch_size // sampler.world
            lo = sampler.rank * per
            for gidx in sampler.global_batches():
                g = gidx.numpy() if index_map is None else np.asarray(index_map)[gidx.numpy()]
                rec = self.draw_recordings(g)
                yield self.batch(g[lo: lo + per], rec=rec[lo: lo + per])
            return
        for idx in sampler:
            i = idx.numpy() if isinstance(idx, torch.Tensor) else np.asarray(idx)
            yield self.batch(i if index_map is None else np.asarray(index_map)[i])


def synthetic_resident_dataset(args, device, *, n_segments: int, n_tasks: int = 4, seed: int = 1234,
                               recs_per_task: int = 2) -> Tuple[ResidentSegmentFeed, np.ndarray, np.ndarray]:
    """Seeded stand-in with the STRUCTURE of Gwilliams2022 (27 subjects x up to 2 sessions x 4 tasks of ~100 k samples are
    not needed to exercise the path): `n_tasks` tasks, `recs_per_task` recordings per task from different subjects (the real
    dataset has every subject hear every task: pass ceil(S / n_tasks) to see all S subjects in a batch), segments laid out
    back to back with an overlap; the speech embedding of a segment is a fixed linear read-out of the raw window of the
    task's FIRST recording plus noise, so retrieval is learnable.  Returns (feed, train indices, test indices)."""
    C = int(args.get("num_channels", 208 if args.dataset == "Gwilliams2022" else 60))
    rate = args.preprocs["brain_resample_rate"]
    T = int(args.preprocs["seq_len_sec"] * rate)
    nb = int(args.preprocs["baseline_len_sec"] * rate)
    F = 1024 if args.preprocs["last4layers"] else int(args.F)
    S = int(args.num_subjects)
    g = torch.Generator().manual_seed(seed)
    per_task = (n_segments + n_tasks - 1) // n_tasks
    hop = max(1, T // 2)
    Lr = per_task * hop + T + 64
    recordings, rec_subject, rec_task, onsets = [], [], [], []
    for t in range(n_tasks):
        base = torch.randn(C, Lr, generator=g)
        for k in range(recs_per_task):
            drift = torch.linspace(0, float(k + 1), Lr)[None, :] * torch.randn(C, 1, generator=g)     # what the collate removes
            recordings.append(((base + 0.3 * torch.randn(C, Lr, generator=g)) * (2.0 + k) + drift).to(device))
            rec_subject.append((recs_per_task * t + k) % S)
            rec_task.append(t)
            onsets.append(np.arange(per_task, dtype=np.int64) * hop + 7 * k)
    seg_task = np.repeat(np.arange(n_tasks), per_task)[:n_segments]
    seg_in_task = np.concatenate([np.arange(per_task)] * n_tasks)[:n_segments]
    P = torch.randn(F, C, generator=g) / np.sqrt(C)
    Y = torch.empty((n_segments, F, T))
    for i in range(n_segments):
        r0 = recs_per_task * int(seg_task[i])
        o = int(onsets[r0][seg_in_task[i]])
        w = recordings[r0][:, o: o + T].cpu()
        w = (w - w[:, :nb].mean(dim=1, keepdim=True)) / (2.0 * 1.35)
        Y[i] = P @ w + 0.5 * torch.randn(F, T, generator=g)
    feed = ResidentSegmentFeed(recordings, rec_subject, rec_task, onsets, seg_task, seg_in_task, Y.to(device), seq_len_samp=T,
                               baseline_len_samp=nb, clamp_lim=float(args.preprocs["clamp_lim"]), clamp=bool(args.preprocs["clamp"]),
                               seed=seed + 17)
    n_train = int(n_segments * float(args.split_ratio))
    perm = np.random.RandomState(seed).permutation(n_segments)
    return feed, perm[:n_train], perm[n_train:]
