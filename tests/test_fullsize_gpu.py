"""Parity at the size the bench measures: BASELINE.json configs[1] — 208 sensors x 360 samples, 27 subjects,
F = 1024, batch 256 — where the launch geometry (sample segments of the weight gradients, XCD-ordered grids, paired
tiles, workspace sizes) differs from the small fixtures.

  * fp32 training step at B = 256 against the LIVE oracle on this box's host cores (one oracle step: tens of seconds):
    embeddings, loss, every gradient, BatchNorm running statistics — the 1e-4 gate of north_star;
  * eval-mode forward at B = 256 in every compute dtype, checked on a random 8-sample subset (eval mode is
    per-sample independent, so the oracle runs at B = 8)."""
import warnings

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import brain_oracle as O                                  # noqa: E402
from tests.parity import operands_as_device_sees_them, rel_l2, round_to   # noqa: E402
from tests.test_e2e_gpu import DTYPES16, build, check_lowprec_step, grads_by_state_key, make_args, null_grad   # noqa: E402

DEV = "cuda:0"
C, S, D1, D2, F, K, T, B = 208, 27, 270, 320, 1024, 32, 360, 256


def _setup(dtype):
    loc = O.synthetic_positions(C, seed=0)
    P = O.seeded_params(C, S, D1, D2, F, K, seed=0, loc=loc)
    args = make_args(C, S, D1, D2, 512, K, True, loc.numpy(), dtype)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        enc, lossf, clf = build(args, P, [5.1])
    X, Y, subj = O.synthetic_batch(B, C, T, F, S, seed=4321)
    return loc, P, enc, lossf, clf, X, Y, subj


def test_config2_full_batch_fp32_train_step_vs_live_oracle():
    loc, P, enc, lossf, clf, X, Y, subj = _setup("fp32")
    enc.train()
    enc.set_drop_centre(5)
    Yd = Y.to(DEV)
    Z = enc(X.to(DEV), subj)
    logits, loss = lossf(Yd, Z, return_logits=True)
    top1, top10 = clf(Z, Yd)
    loss.backward()
    stats = {k: v.clone() for k, v in P.items() if "running" in k or "num_batches" in k}
    lo, Zo, logits_o, go = O.train_step(P, torch.tensor([5.1]), X, Y, subj, loc=loc, drop_centre=5, stats=stats)
    Zc = Z.detach().cpu()
    assert float((Zc - Zo).abs().max()) <= 1e-4 * float(Zo.abs().max()), "embeddings"
    assert rel_l2(Zc, Zo) < 2e-5
    assert abs(float(loss.detach()) - float(lo)) < 1e-4
    np.testing.assert_allclose(logits.cpu().numpy(), logits_o.numpy(), rtol=1e-3, atol=5e-3)
    assert (top1, top10) == pytest.approx(O.topk_accuracy(Zo, Y))
    for k, g in grads_by_state_key(enc).items():
        ref = go[k]
        if ref is None:                                      # a subject absent from the batch
            assert g is None or float(g.abs().max()) == 0.0, k
            continue
        gf = (torch.view_as_real(g) if g.is_complex() else g).float().cpu().reshape(-1)
        rf = (torch.view_as_real(ref) if ref.is_complex() else ref).reshape(-1)
        if null_grad(k):                                     # mathematically zero (feeds a training-mode BatchNorm)
            assert float(gf.abs().max()) < 1e-4, k
            continue
        assert float((gf - rf).abs().max()) <= 2e-3 * float(rf.abs().max()) + 1e-9, (k, float((gf - rf).abs().max()), float(rf.abs().max()))
        assert rel_l2(gf, rf) < 1e-3, (k, rel_l2(gf, rf))
    assert abs(float(lossf.temp.grad) - float(go["temp"])) < 1e-3 * max(1.0, abs(float(go["temp"])))
    sd = enc.state_dict()
    for k, v in stats.items():
        if "running" in k:
            np.testing.assert_allclose(sd[k].cpu().numpy(), v.numpy(), rtol=1e-4, atol=1e-5, err_msg=k)


@pytest.mark.parametrize("dtype", ["fp32"] + DTYPES16)
def test_config2_full_batch_eval_forward_sampled_against_oracle(dtype):
    loc, P, enc, lossf, clf, X, Y, subj = _setup(dtype)
    enc.eval()
    with torch.no_grad():
        Z = enc(X.to(DEV), subj)
        Z_again = enc(X.to(DEV), subj)
    assert Z.data_ptr() != Z_again.data_ptr()                 # a fresh tensor per forward, as in the reference
    assert torch.equal(Z, Z_again)                            # bitwise run-to-run reproducible
    pick = torch.from_numpy(np.random.RandomState(3).choice(B, 8, replace=False)).sort().values
    Pr = operands_as_device_sees_them(P, dtype)
    Zo = O.brain_encoder_forward(Pr, round_to(X[pick], dtype), subj[pick], training=False)
    got = Z[pick.to(DEV)].float().cpu()
    if dtype == "fp32":
        assert float((got - Zo).abs().max()) <= 1e-4 * float(Zo.abs().max())
    else:
        assert rel_l2(got, Zo) < REL_EVAL[dtype], rel_l2(got, Zo)
        assert float((got - Zo).abs().max()) <= MAX_EVAL[dtype] * float(Zo.abs().max())


# 16-bit storage between the ~20 kernels of a forward: calibrated on tests/precision_survey.py (x ~2 margin)
REL_EVAL = {"bf16": 4e-2, "fp16": 5e-3}
MAX_EVAL = {"bf16": 8e-2, "fp16": 1e-2}


def test_config2_full_batch_bf16_train_step_vs_oracle_on_rounded_operands():
    """THE benchmarked configuration in THE benchmarked dtype (BASELINE.json configs[1]: 208 ch, batch 256, bf16): one
    training step — embeddings, loss, temperature gradient and EVERY parameter gradient — against the oracle fed the same
    rounded operands, with the same per-tensor bounds as the small shapes (tests/test_e2e_gpu.py)."""
    loc, P, enc, lossf, clf, X, Y, subj = _setup("bf16")
    report = check_lowprec_step(enc, lossf, P, [5.1], X, Y, subj, loc, 5, "bf16")
    assert len([k for k in report if k.startswith("grad ")]) >= 40


@pytest.mark.parametrize("name,Cc,Ss,Tt,dtype", [("configs[3] per-rank batch: 60 ch, 1 subject, batch 512", 60, 1, 360, "bf16"),
                                                 ("configs[4] per-rank batch: 306 ch, T=1000, 100 subjects, batch 512", 306, 100, 1000, "fp16")])
def test_other_configs_per_rank_batch_eval_forward_sampled_against_oracle(name, Cc, Ss, Tt, dtype):
    """BASELINE.json configs[3] / configs[4] at the batch ONE rank of the 8-GPU run holds (512 segments): eval-mode forward in
    the dtype the config names, checked on a random 8-sample subset (eval mode is per-sample independent)."""
    Bb = 512
    loc = O.synthetic_positions(Cc, seed=1)
    P = O.seeded_params(Cc, Ss, D1, D2, F, K, seed=1, loc=loc)
    args = make_args(Cc, Ss, D1, D2, 512, K, True, loc.numpy(), dtype)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        enc, lossf, clf = build(args, P, [5.1])
    X, _, subj = O.synthetic_batch(Bb, Cc, Tt, 1, Ss, seed=99)          # (the speech side is not needed here)
    enc.eval()
    with torch.no_grad():
        Z = enc(X.to(DEV), subj)
    assert tuple(Z.shape) == (Bb, F, Tt)
    pick = torch.from_numpy(np.random.RandomState(5).choice(Bb, 8, replace=False)).sort().values
    Zo = O.brain_encoder_forward(operands_as_device_sees_them(P, dtype), round_to(X[pick], dtype), subj[pick], training=False)
    got = Z[pick.to(DEV)].float().cpu()
    assert rel_l2(got, Zo) < REL_EVAL[dtype], rel_l2(got, Zo)
    assert float((got - Zo).abs().max()) <= MAX_EVAL[dtype] * float(Zo.abs().max())
    enc.engine.release_workspace()


def _free_host_gib():
    try:
        import psutil
        return psutil.virtual_memory().available / 2 ** 30
    except Exception:                                                 # noqa: BLE001
        return 0.0


@pytest.mark.parametrize("name,Cc,Ss,Tt,dtype", [("configs[3] per-rank batch: 60 ch, 1 subject, batch 512, bf16", 60, 1, 360, "bf16"),
                                                 ("configs[4] per-rank batch: 306 ch, T=1000, 100 subjects, batch 512, fp16", 306, 100, 1000, "fp16")])
def test_other_configs_per_rank_batch_train_step_vs_oracle(name, Cc, Ss, Tt, dtype, record_property):
    """BASELINE.json configs[3] / configs[4] at the batch ONE rank of the 8-GPU run holds (512 segments), TRAINING step in the
    dtype the config names: embeddings, loss, temperature gradient and every parameter gradient against the oracle fed the same
    rounded operands (reference: models.py:191-196 + utils/loss.py:58-79 through autograd).  This is the backward geometry the
    small fixtures never reach: weight-gradient sample segments at B = 512 / T = 1000, slab sums, one subject cut into r > 1
    K-slices (S = 1), ~36 GiB of workspace at configs[4].  The oracle keeps every activation for autograd in fp32 on the host;
    where the host cannot hold batch 512 the test runs the largest batch that fits (a multiple of 64, at least 128) and says so."""
    per_sample = Tt * (45 * D2 + 6 * F) * 4 * 1.3 / 2 ** 30            # GiB the oracle's autograd graph needs per sample
    free = _free_host_gib()
    Bb = 512
    while Bb > 128 and Bb * per_sample + 6 > free:
        Bb -= 64
    if Bb * per_sample + 6 > free:
        pytest.skip(f"{name}: the CPU oracle needs ~{128 * per_sample + 6:.0f} GiB of host memory even at batch 128 ({free:.0f} free)")
    record_property("batch", Bb)
    if Bb != 512:
        warnings.warn(f"{name}: host memory ({free:.0f} GiB free) holds the oracle at batch {Bb}, not 512")
    loc = O.synthetic_positions(Cc, seed=1)
    P = O.seeded_params(Cc, Ss, D1, D2, F, K, seed=1, loc=loc)
    args = make_args(Cc, Ss, D1, D2, 512, K, True, loc.numpy(), dtype)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        enc, lossf, clf = build(args, P, [5.1])
    X, Y, subj = O.synthetic_batch(Bb, Cc, Tt, F, Ss, seed=77)
    report = check_lowprec_step(enc, lossf, P, [5.1], X, Y, subj, loc, 3, dtype, scale_for=(Bb, Tt))
    assert len([k for k in report if k.startswith("grad ")]) >= 40
    enc.engine.release_workspace()
