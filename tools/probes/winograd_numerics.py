"""Winograd F(2,3) along time on the dilated lattice — the NUMERICS half of the go / no-go (CPU, no GPU needed).

A kernel-3 dilated conv  y[t] = sum_tap W_tap x[t + (tap - 1) d]  computed for output pairs (t, t + d) from the four inputs
x[t - d], x[t], x[t + d], x[t + 2d] with 4 channel-contractions instead of 6 (Lavin & Gray's F(2,3), applied along the dilated
time lattice).  On the 16-bit MFMA path the TRANSFORMED inputs (x0 - x2, x1 + x2, x2 - x1, x1 - x3) and the transformed
weights (g0, (g0 + g1 + g2) / 2, (g0 - g1 + g2) / 2, g2) have to be 16-bit MFMA operands, i.e. they are rounded once more.
This script measures what that costs: relative L2 error against fp64 of (a) the direct conv on 16-bit-rounded operands with
fp32 accumulation (what conv3_flat computes) and (b) the Winograd form with its operands rounded to the same type.
    python tools/probes/winograd_numerics.py"""
import torch

torch.manual_seed(0)
B, C, T, d = 4, 320, 360, 4


def rnd(x, dt):
    return x.to(dt).to(torch.float64)


for dt, name in ((torch.bfloat16, "bf16"), (torch.float16, "fp16")):
    x = torch.randn(B, C, T + 3 * d, dtype=torch.float64)              # activations ~ N(0, 1) (post-GELU ones are similar in scale)
    w = torch.randn(3, C, C, dtype=torch.float64) / (3 * C) ** 0.5      # [tap][co][ci]
    xr, wr = rnd(x, dt), rnd(w, dt)
    t = torch.arange(d, T + d)                                         # outputs whose four inputs exist
    ref = sum(torch.einsum("oc,bct->bot", w[k], x[:, :, t + (k - 1) * d]) for k in range(3))
    # (a) direct form on rounded operands, products exact, accumulation in fp32 (emulated: fp64 sums rounded once — the MFMA's
    # fp32 accumulation error is far below the operand rounding)
    direct = sum(torch.einsum("oc,bct->bot", wr[k], xr[:, :, t + (k - 1) * d]) for k in range(3))
    # (b) Winograd: pairs (t, t + d) over blocks of 2d outputs
    tt = t[: (len(t) // (2 * d)) * 2 * d].reshape(-1, 2, d)             # [block][which of the pair][offset]
    t0, t1 = tt[:, 0].reshape(-1), tt[:, 1].reshape(-1)
    x0, x1, x2, x3 = (xr[:, :, t0 + k * d - d] for k in range(4))
    g = [wr[0], (wr[0] + wr[1] + wr[2]) / 2, (wr[0] - wr[1] + wr[2]) / 2, wr[2]]
    A = [rnd(x0 - x2, dt), rnd(x1 + x2, dt), rnd(x2 - x1, dt), rnd(x1 - x3, dt)]      # transformed inputs: MFMA operands
    G = [rnd(gi, dt) for gi in g]                                                      # transformed weights: MFMA operands
    m = [torch.einsum("oc,bct->bot", G[i], A[i]) for i in range(4)]
    y0, y1 = m[0] + m[1] + m[2], m[1] - m[2] - m[3]
    wino = torch.empty_like(ref)
    pos = {int(v): i for i, v in enumerate(t)}
    i0 = torch.tensor([pos[int(v)] for v in t0]); i1 = torch.tensor([pos[int(v)] for v in t1])
    wino[:, :, i0], wino[:, :, i1] = y0, y1
    covered = torch.cat([i0, i1])
    rel = lambda a: float((a[:, :, covered] - ref[:, :, covered]).norm() / ref[:, :, covered].norm())
    print(f"{name}: direct conv on rounded operands {rel(direct):.2e}   Winograd F(2,3), transformed operands rounded {rel(wino):.2e}   "
          f"ratio {rel(wino) / rel(direct):.2f}")
