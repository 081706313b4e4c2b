#!/usr/bin/env python3
"""Per-tensor relative L2 errors of the 16-bit paths against the CPU oracle fed the same rounded operands — the numbers
the bounds in tests/test_e2e_gpu.py / tests/test_fullsize_gpu.py are calibrated on.  Runs on the GPU box:
    python tests/precision_survey.py [bf16 fp16]"""
import os
import sys
import warnings

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import brain_oracle as O          # noqa: E402   (test infrastructure)
from tests.parity import operands_as_device_sees_them, rel_l2, round_to, temp_grad_terms_norm      # noqa: E402
from tests.test_e2e_gpu import build, grads_by_state_key, make_args, null_grad   # noqa: E402

DEV = "cuda:0"


def survey(name, C, S, T, B, dtype, D1=270, D2=320, F=1024, K=32, last4=True, seed=4):
    loc = O.synthetic_positions(C, seed=2)
    P = O.seeded_params(C, S, D1, D2, F, K, seed=seed, loc=loc)
    args = make_args(C, S, D1, D2, 512 if last4 else F, K, last4, loc.numpy(), dtype)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        enc, lossf, clf = build(args, P, [5.1])
    X, Y, subj = O.synthetic_batch(B, C, T, F, S, seed=11)
    enc.train()
    enc.set_drop_centre(7)
    Z = enc(X.to(DEV), subj)
    loss = lossf(Y.to(DEV), Z)
    from speech_decoding_amd.amp import LossScaler
    scaler = LossScaler.for_dtype(Z.dtype)
    scaler.scale(loss).backward()
    scaler.unscale_(list(enc.parameters()) + list(lossf.parameters()))
    Pr = operands_as_device_sees_them(P, dtype)
    taps = {}
    lo, Zo, lgo, go = O.train_step(Pr, torch.tensor([5.1]), round_to(X, dtype), round_to(Y, dtype), subj, loc=loc, drop_centre=7,
                                 taps=taps)
    # the shared 1x1 conv's bias gradient is an almost exactly cancelling sum: its error against the scale of its TERMS
    k = "subject_block.conv.bias"
    terms = taps["subject_block.conv.out"].grad
    err = (grads_by_state_key(enc)[k].float().cpu() - go[k]).norm()
    print(f"   {k}: |err| / |sqrt(sum terms^2)| = {float(err / terms.pow(2).sum(dim=(0, 2)).sqrt().norm()):.3e}   "
          f"|err| / |sum |terms|| = {float(err / terms.abs().sum(dim=(0, 2)).norm()):.3e}   |err| / |grad| = {float(err / go[k].norm()):.3e}")
    print(f"== {name} {dtype}: Z rel_l2 {rel_l2(Z.detach().float(), Zo):.3e}  max/max {float((Z.detach().float().cpu() - Zo).abs().max() / Zo.abs().max()):.3e}"
          f"  loss {float(loss):.5f} vs {float(lo):.5f} (rel {abs(float(loss) - float(lo)) / float(lo):.2e})"
          f"  temp.grad {float(lossf.temp.grad):.5f} vs {float(go['temp']):.5f}"
          f"  |err| / |terms| = {abs(float(lossf.temp.grad) - float(go['temp'])) / temp_grad_terms_norm(lgo):.3e}")
    worst = []
    for k, g in grads_by_state_key(enc).items():
        ref = go[k]
        if ref is None or g is None:
            continue
        e = rel_l2(g, ref)
        worst.append((e, k, null_grad(k)))
    for e, k, ng in sorted(worst, reverse=True)[:14]:
        print(f"   {e:.3e}  {k}{'  (null-gradient bias)' if ng else ''}")


if __name__ == "__main__":
    dts = [a for a in sys.argv[1:] if not a.startswith("--")] or ["bf16"]
    for dt in dts:
        survey("small", 12, 3, 40, 6, dt, D1=16, D2=24, F=32, K=4, last4=False)
        survey("spot208-B8", 208, 27, 360, 8, dt)
        survey("spot60-B8", 60, 1, 360, 8, dt)
        survey("config5-shape-B12", 306, 100, 1000, 12, dt)
        survey("config2-B64", 208, 27, 360, 64, dt)
        if "--full" in sys.argv:
            survey("config2-B256", 208, 27, 360, 256, dt)
