"""GPU batch collate — drop-in for the reference's `Gwilliams2022Collator` (gwilliams2022.py:640-661):
baseline correction + per-(sample, channel) RobustScaler + clamp, run as ONE kernel on the device instead
of a Python loop over the batch with sklearn on DataLoader workers."""
from __future__ import annotations

from typing import List

import torch
import torch.nn as nn

from . import lib as L
from . import ops


def robust_scale_clamp(X: torch.Tensor, baseline_len_samp: int, clamp_lim: float, clamp: bool = True) -> torch.Tensor:
    """X: (B, C, T) float on the GPU -> same shape, fp32."""
    if not X.is_cuda:
        raise L.SdaError("collate needs device tensors (no CPU fallback)")
    B, C, T = X.shape
    src = X.contiguous().float()
    out = torch.empty_like(src)
    L.check(L.load().sda_collate_rows(src.data_ptr(), out.data_ptr(), B * C, T, int(baseline_len_samp), float(clamp_lim),
                                      int(bool(clamp)), torch.cuda.current_stream().cuda_stream), "collate_rows")
    return out


class Gwilliams2022Collator(nn.Module):
    def __init__(self, args, device="cuda"):
        super().__init__()
        self.brain_resample_rate = args.preprocs["brain_resample_rate"]
        self.baseline_len_samp = int(self.brain_resample_rate * args.preprocs["baseline_len_sec"])
        self.clamp = args.preprocs["clamp"]
        self.clamp_lim = args.preprocs["clamp_lim"]
        self.device = device

    def forward(self, batch: List[tuple]):
        X = torch.stack([item[0] for item in batch]).to(self.device, non_blocking=True)
        Y = torch.stack([item[1] for item in batch])
        subject_idx = torch.IntTensor([item[2] for item in batch])
        X = robust_scale_clamp(X, self.baseline_len_samp, self.clamp_lim, self.clamp)
        return X, Y, subject_idx


class ResidentSegments:
    """Segment gather on the GPU (gwilliams2022.py:129-142): per-session recordings (C, L_s) stay resident in HBM;
    a batch is described by (session index, onset sample) per segment and materialised — window extraction,
    baseline correction, robust scaling, clamp — by one kernel, replacing `__getitem__` + the collator."""

    def __init__(self, sessions, seq_len_samp: int, baseline_len_samp: int, clamp_lim: float, clamp: bool = True):
        self.sessions = [s.contiguous().float() for s in sessions]
        if not all(s.is_cuda and s.dim() == 2 for s in self.sessions):
            raise L.SdaError("ResidentSegments needs (C, L) device tensors")
        self.C = self.sessions[0].shape[0]
        self.T, self.nb, self.lim, self.clamp = int(seq_len_samp), int(baseline_len_samp), float(clamp_lim), bool(clamp)
        self._base = torch.tensor([s.data_ptr() for s in self.sessions], dtype=torch.int64)
        self._len = torch.tensor([s.shape[1] for s in self.sessions], dtype=torch.int64)

    def batch(self, session_idx, onsets) -> torch.Tensor:
        """session_idx, onsets: integer sequences of length B (host) -> X (B, C, T) fp32 on the device."""
        sidx = torch.as_tensor(session_idx, dtype=torch.int64)
        on = torch.as_tensor(onsets, dtype=torch.int64)
        if bool(((on < 0) | (on + self.T > self._len[sidx])).any()):
            raise IndexError("segment window leaves its session recording")
        dev = self.sessions[0].device
        # the two index tables travel in kernel arguments (ops.upload_small): a copy from pageable host memory makes the host
        # wait for the stream it is queued on — with the step's loop two milliseconds ahead of the GPU that wait is the step's
        with torch.cuda.device(dev):
            ptrs = ops.upload_small((self._base[sidx] + 4 * on).numpy(), dev)
            cstr = ops.upload_small(self._len[sidx].numpy(), dev)
        out = torch.empty((len(sidx), self.C, self.T), dtype=torch.float32, device=dev)
        L.check(L.load().sda_collate_windows(ptrs.data_ptr(), cstr.data_ptr(), out.data_ptr(), len(sidx), self.C, self.T, self.nb,
                                             self.lim, int(self.clamp), torch.cuda.current_stream().cuda_stream), "collate_windows")
        return out
