"""Adam over all parameter tensors in ONE kernel launch (same update rule and defaults as the
`torch.optim.Adam(params, lr=...)` of train.py:161-163; complex parameters are stepped through their real
view, exactly as PyTorch does).  Optional: the stock optimiser works unchanged with these modules."""
from __future__ import annotations

import ctypes as C

import torch

from . import lib as L


class FusedAdam(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8):
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps))
        L.load()

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        for group in self.param_groups:
            items = []
            for p in group["params"]:
                if p.grad is None:
                    continue
                if not p.is_cuda:
                    raise L.SdaError("FusedAdam needs device parameters (no CPU path)")
                st = self.state[p]
                if not st:
                    st["step"] = 0
                    st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                st["step"] += 1
                real = (lambda t: torch.view_as_real(t) if t.is_complex() else t)
                g = p.grad if p.grad.is_contiguous() else p.grad.contiguous()
                items.append((real(p.data), real(g), real(st["exp_avg"]), real(st["exp_avg_sq"]), st["step"]))
            if not items:
                continue
            steps = {it[4] for it in items}
            for stp in sorted(steps):                       # parameters that joined later have their own step count
                sub = [it for it in items if it[4] == stp]
                arr = (L.AdamDesc * len(sub))()
                for i, (pp, gg, m, v, _) in enumerate(sub):
                    arr[i].param, arr[i].grad, arr[i].exp_avg, arr[i].exp_avg_sq = pp.data_ptr(), gg.data_ptr(), m.data_ptr(), v.data_ptr()
                    arr[i].n = pp.numel()
                    arr[i].aligned = int(all(t.data_ptr() % 16 == 0 for t in (pp, gg, m, v)))
                import numpy as np
                from .ops import UPLOADER
                # gradient tensors recur at the same addresses in steady state: the table is then a cache hit
                table = UPLOADER.upload(("adam", id(self)), np.frombuffer(bytes(arr), dtype=np.uint8), sub[0][0].device)
                b1, b2 = group["betas"]
                L.check(L.load().sda_adam_multi(table.data_ptr(), len(sub), max(int(a.n) for a in arr), float(group["lr"]),
                                                float(b1), float(b2), float(group["eps"]), int(stp),
                                                torch.cuda.current_stream().cuda_stream), "adam_multi")
                self._keep = (table, [s[1] for s in sub])     # keep the table / contiguous grads alive until the next step
        return loss
