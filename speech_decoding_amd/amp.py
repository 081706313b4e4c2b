"""Static loss scaling for the fp16 compute path.

fp16 has 5 exponent bits: the activation gradients of this model (~1e-5 at the training shapes) sit in its subnormal
range, so — as with any fp16 training — the loss is multiplied by a power of two before `backward()` and the fp32
parameter gradients are divided by it afterwards.  The whole backward pass is linear in dL/dZ, and a power of two
changes no mantissa, so the result equals the unscaled one wherever nothing under- or overflows.  bf16 and fp32 need
none of this (`LossScaler(1)` is a no-op).

    scaler = LossScaler.for_dtype(brain_encoder.compute_dtype)
    scaler.scale(loss).backward()
    scaler.unscale_(params)           # before the all-reduce / optimiser step
"""
from __future__ import annotations

from typing import Iterable

import torch

DEFAULT_FP16_SCALE = 1024.0


class LossScaler:
    def __init__(self, scale: float = 1.0):
        self.scale_value = float(scale)

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
