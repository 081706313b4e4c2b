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


def fp16_scale_for(global_batch: int, T: int, base: float = DEFAULT_FP16_SCALE, ref_batch: int = 256, ref_T: int = 360) -> float:
    """Static fp16 loss scale for a run with `global_batch` segments of `T` samples.  The gradient that enters the encoder is
    ~ exp(temp) / (2 B_global |Y| |Z|) with norms ~ sqrt(F T), i.e. it shrinks like 1 / (B_global T): 9e-7 at batch 256 x 360
    samples (where `base` = 1024 puts it at 9e-4), 2e-8 at BASELINE configs[4]'s 4096 x 1000 — below fp16's smallest
    subnormal (6e-8) unscaled and still subnormal at 1024.  Scaled by the same power of two the gradients stay where the
    calibration left them; parameter gradients are fp32 and are un-scaled before the optimiser sees them."""
    import math
    f = max(1.0, global_batch / ref_batch) * max(1.0, T / ref_T)
    return min(float(base) * 2.0 ** math.ceil(math.log2(f)), 2.0 ** 24)


class LossScaler:
    """enabled: the fp16 path (set by for_dtype; independent of the scale's VALUE — a scale that repeated overflows have
    halved down to 1, or fp16_loss_scale=1 in the config, still gets its gradients checked for inf / NaN before Adam sees
    them).  max_scale: growth never passes max(initial scale, 2^24): fp16_scale_for() may start above GradScaler's usual
    65536 cap (131072 at 8192 x 1000 samples), and capping growth below the calibrated start would push the activation
    gradients back into fp16's subnormal range."""

    def __init__(self, scale: float = 1.0, growth_interval: int = 2000, enabled: bool | None = None):
        self.scale_value = float(scale)
        self.enabled = bool(scale != 1.0) if enabled is None else bool(enabled)
        self.max_scale = max(float(scale), 2.0 ** 24)
        self.growth_interval = int(growth_interval)
        self._good_steps = 0
        self.skipped_steps = 0
        self._found_inf = None          # device flag of the fused finiteness check (one per device)

    @classmethod
    def for_dtype(cls, dtype: torch.dtype, scale: float = DEFAULT_FP16_SCALE, global_batch: int = 0, T: int = 0) -> "LossScaler":
        """fp16: `scale`, or with (global_batch, T) given the scale fp16_scale_for() derives from them; other dtypes: a no-op."""
        if dtype != torch.float16:
            return cls(1.0, enabled=False)
        return cls(fp16_scale_for(global_batch, T, scale) if global_batch and T else scale, enabled=True)

    def scale(self, loss: torch.Tensor) -> torch.Tensor:
        return loss if self.scale_value == 1.0 else loss * self.scale_value

    def unscale_(self, params: Iterable[torch.nn.Parameter], check: bool = False) -> bool:
        """Divide every gradient by the scale and, with check=True, test them for inf / NaN — ONE fused launch
        (torch._amp_foreach_non_finite_check_and_unscale_, what GradScaler uses) writing one device flag, read back once
        (the host synchronisation of check=True).  Returns False when a gradient is not finite: skip the step and lower
        the scale.  The check runs whenever the scaler is enabled (fp16), whatever the scale's value."""
        if not self.enabled and self.scale_value == 1.0:
            return True
        grads = [torch.view_as_real(p.grad) if p.grad.is_complex() else p.grad for p in params if p.grad is not None]
        if not grads:
            return True
        if not check:
            if self.scale_value != 1.0:
                torch._foreach_mul_(grads, 1.0 / self.scale_value)
            return True
        dev = grads[0].device
        if self._found_inf is None or self._found_inf.device != dev:
            self._found_inf = torch.zeros(1, dtype=torch.float32, device=dev)
            self._inv = torch.ones(1, dtype=torch.float32, device=dev)
        self._found_inf.zero_()
        self._inv.fill_(1.0 / self.scale_value)
        torch._amp_foreach_non_finite_check_and_unscale_(grads, self._found_inf, self._inv)
        return not bool(self._found_inf.item())

    def update(self, ok: bool) -> None:
        """Dynamic scaling, for callers that pass check=True to unscale_() (torch.cuda.amp.GradScaler's rule): a step whose
        gradients were not finite is skipped by the caller and halves the scale; `growth_interval` good steps in a row
        double it, never past max_scale.  A disabled scaler (bf16 / fp32) never changes."""
        if not self.enabled:
            return
        if not ok:
            self.scale_value = max(1.0, self.scale_value * 0.5)
            self._good_steps = 0
            self.skipped_steps += 1
            return
        self._good_steps += 1
        if self._good_steps >= self.growth_interval:
            self.scale_value = min(self.max_scale, self.scale_value * 2.0)
            self._good_steps = 0
