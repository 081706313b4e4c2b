"""Static loss scaling for the fp16 compute path.

fp16 has 5 exponent bits: the activation gradients of this model (~1e-5 at the training shapes) sit in its subnormal
range, so — as with any fp16 training — the loss is multiplied by a power of two before `backward()` and the fp32
parameter gradients are divided by it afterwards.  The whole backward pass is linear in dL/dZ, and a power of two
changes no mantissa, so the result equals the unscaled one wherever nothing under- or overflows.  bf16 and fp32 need
none of this (`LossScaler(1)` is a no-op).

    scaler = LossScaler.for_dtype(brain_encoder.compute_dtype)
    scaler.scale(loss).backward()
    ok = scaler.unscale_(params, check=True)      # before the all-reduce / optimiser step; check=True is a host sync
    scaler.update(ok)                             # overflow: halve the scale ...
    if ok: optimizer.step()                       # ... and skip the step
"""
from __future__ import annotations

from typing import Iterable

import torch

DEFAULT_FP16_SCALE = 1024.0


class LossScaler:
    def __init__(self, scale: float = 1.0, growth_interval: int = 2000):
        self.scale_value = float(scale)
        self.growth_interval = int(growth_interval)
        self._good_steps = 0
        self.skipped_steps = 0

    @classmethod
    def for_dtype(cls, dtype: torch.dtype, scale: float = DEFAULT_FP16_SCALE) -> "LossScaler":
        return cls(scale if dtype == torch.float16 else 1.0)

    def scale(self, loss: torch.Tensor) -> torch.Tensor:
        return loss if self.scale_value == 1.0 else loss * self.scale_value

    def unscale_(self, params: Iterable[torch.nn.Parameter], check: bool = False) -> bool:
        """Divide every gradient by the scale (one fused launch).  With check=True (a host synchronisation) returns
        False when a gradient is not finite: skip the step and lower the scale."""
        if self.scale_value == 1.0:
            return True
        grads = [torch.view_as_real(p.grad) if p.grad.is_complex() else p.grad for p in params if p.grad is not None]
        if not grads:
            return True
        torch._foreach_mul_(grads, 1.0 / self.scale_value)
        if not check:
            return True
        return bool(torch.isfinite(torch.stack([g.abs().max() for g in grads])).all())

    def update(self, ok: bool) -> None:
        """Dynamic scaling, for callers that pass check=True to unscale_() (torch.cuda.amp.GradScaler's rule): a step whose
        gradients were not finite is skipped by the caller and halves the scale; `growth_interval` good steps in a row
        double it.  A scale of 1 (bf16 / fp32) never changes."""
        if self.scale_value == 1.0:
            return
        if not ok:
            self.scale_value = max(1.0, self.scale_value * 0.5)
            self._good_steps = 0
            self.skipped_steps += 1
            return
        self._good_steps += 1
        if self._good_steps >= self.growth_interval:
            self.scale_value = min(65536.0, self.scale_value * 2.0)
            self._good_steps = 0
