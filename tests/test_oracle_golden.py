"""Pin the CPU oracle (oracle/brain_oracle.py) against outputs of the reference itself.

The fixtures were written by tests/golden/make_golden.py, which ran the reference's own
BrainEncoder / CLIPLoss / Classifier in the build container.  CPU only."""
import numpy as np
import pytest
import torch

from oracle import brain_oracle as O
from tests import golden_io as G

TOL = dict(rtol=2e-5, atol=2e-6)


def null_grad(key):
    """conv0/conv1 biases feed a training-mode BatchNorm, which removes any per-channel constant:
    their gradient is mathematically zero and what the reference reports is rounding noise."""
    return key.startswith("conv_blocks.") and key.endswith((".conv0.bias", ".conv1.bias"))


@pytest.fixture(scope="module")
def small():
    return G.load("e2e_small.npz")


def _run_small(small):
    P = G.state_from(small, "init/")
    P.pop("temp")
    temp = torch.from_numpy(small["init/temp"])
    loc = torch.from_numpy(small["loc"])
    X, Y = torch.from_numpy(small["X"]), torch.from_numpy(small["Y"])
    subj = torch.from_numpy(small["subject_idxs"])
    return P, temp, loc, X, Y, subj


def test_fourier_tables_match_reference_buffers(small):
    P, _, loc, *_ = _run_small(small)
    cos, sin = O.fourier_tables(loc, int(small["dims"][5]))
    np.testing.assert_allclose(cos.numpy(), P["subject_block.spatial_attention.cos"].numpy(), **TOL)
    np.testing.assert_allclose(sin.numpy(), P["subject_block.spatial_attention.sin"].numpy(), **TOL)


def test_stage_activations(small):
    P, temp, loc, X, Y, subj = _run_small(small)
    centre = int(small["drop_centres"][0])
    mask = O.dropout_mask(loc, centre, 0.1)
    sa = O.spatial_attention(P, X, mask)
    np.testing.assert_allclose(sa.numpy(), small["act/spatial_attention"], **TOL)
    h = O.subject_block(P, X, subj, mask)
    np.testing.assert_allclose(h.numpy(), small["act/subject_block"], **TOL)
    for k in range(5):
        h = O.conv_block(P, h, k, True, None)
        np.testing.assert_allclose(h.numpy(), small[f"act/conv_block{k}"], rtol=1e-4, atol=1e-5)


def test_train_step_forward_backward(small):
    P, temp, loc, X, Y, subj = _run_small(small)
    stats = {k: v.clone() for k, v in P.items() if "running" in k or "num_batches" in k}
    loss, Z, logits, grads = O.train_step(P, temp, X, Y, subj, loc=loc,
                                          drop_centre=int(small["drop_centres"][0]), stats=stats)
    np.testing.assert_allclose(Z.numpy(), small["step0/Z"], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(logits.numpy(), small["step0/logits"], rtol=1e-4, atol=2e-4)
    np.testing.assert_allclose(loss.numpy(), small["step0/loss"], rtol=1e-5)
    for k, g in grads.items():
        if k == "temp":
            ref = small["grad/temp"]
            got = g.numpy()
        elif g.is_complex():
            ref = small[f"grad/{k}@re"] + 1j * small[f"grad/{k}@im"]
            got = g.numpy()
        else:
            ref = small[f"grad/{k}"]
            got = g.numpy()
        if null_grad(k):
            assert np.abs(got).max() < 1e-4 and np.abs(ref).max() < 1e-4, k
            continue
        scale = max(np.abs(ref).max(), 1e-8)
        assert np.abs(got - ref).max() <= 2e-4 * scale + 1e-7, k
    for k, v in stats.items():
        np.testing.assert_allclose(v.numpy(), small["after1fwd/" + k], rtol=1e-5, atol=1e-6)


def test_two_adam_steps_and_eval(small):
    """train.py:161-163,200-203 — Adam(lr=3e-4) over encoder params + temp, then eval-mode forward."""
    P, temp, loc, X, Y, subj = _run_small(small)
    names = [k for k, v in P.items()
             if (v.is_floating_point() or v.is_complex()) and "running" not in k
             and not k.endswith(".cos") and not k.endswith(".sin")]
    params = [P[k].clone().requires_grad_(True) for k in names]
    t = temp.clone().requires_grad_(True)
    stats = {k: v.clone() for k, v in P.items() if "running" in k or "num_batches" in k}
    opt = torch.optim.Adam(params + [t], lr=3e-4)
    losses = []
    for step in range(2):
        Q = dict(P)
        Q.update(dict(zip(names, params)))
        Z = O.brain_encoder_forward(Q, X, subj, training=True, loc=loc,
                                    drop_centre=int(small["drop_centres"][step]), stats=stats)
        loss, _ = O.clip_loss(Y, Z, t)
        opt.zero_grad()
        loss.backward()
        opt.step()
        losses.append(loss.item())
    assert abs(losses[0] - float(small["step0/loss"])) < 1e-4
    assert abs(losses[1] - float(small["step1/loss"])) < 2e-3
    after = G.state_from(small, "after2/")
    for k, p in zip(names, params):
        ref = after[k]
        got = p.detach()
        if ref.is_complex():
            ref, got = torch.view_as_real(ref), torch.view_as_real(got)
        # Adam's first steps move every weight by ≈lr whatever the gradient's size, so entries whose
        # gradient is rounding noise (null_grad keys; z[:, m=0], whose Fourier row is constant over
        # sensors and cancels in the softmax) may differ by up to 2 steps × 2·lr between two runs.
        diff = (got - ref).abs()
        assert diff.max().item() < 4 * 3e-4 + 1e-6, k
        if null_grad(k):
            continue
        if k.endswith(".z"):
            g0 = np.stack([small[f"grad/{k}@re"], small[f"grad/{k}@im"]], axis=-1)
        else:
            g0 = small[f"grad/{k}"]
        solid = torch.from_numpy(np.abs(g0) > 1e-3 * np.abs(g0).max())
        assert diff[solid].max().item() < 2e-5, k
    assert abs(t.item() - float(small["after2/temp"][0])) < 1e-5
    Q = dict(after)
    Q.pop("temp")
    for k, v in stats.items():   # second update saw biases moved ±lr by noise-gradient Adam steps
        np.testing.assert_allclose(v.numpy(), after[k].numpy(), rtol=1e-3, atol=2e-4)
    Ze = O.brain_encoder_forward(Q, X, subj, training=False)
    le, _ = O.clip_loss(Y, Ze, after["temp"])
    np.testing.assert_allclose(Ze.numpy(), small["eval/Z"], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(le.numpy(), small["eval/loss"], rtol=1e-4)


def test_classifier_matches_reference_loop():
    c = G.load("classifier.npz")
    Z, Y = torch.from_numpy(c["Z"]), torch.from_numpy(c["Y"])
    assert O.topk_accuracy(Z, Y) == pytest.approx((float(c["top1"]), float(c["top10"])))
    assert O.classifier_loop(Z, Y) == pytest.approx((float(c["top1"]), float(c["top10"])))


def test_clip_loss_asserts_batch_gt_one():
    with pytest.raises(AssertionError):
        O.clip_loss(torch.randn(1, 4, 3), torch.randn(1, 4, 3), torch.tensor([5.1]))


@pytest.mark.parametrize("tag,C,S", [("208", 208, 27), ("60", 60, 1)])
def test_full_dimension_spot_checks(tag, C, S):
    s = G.load(f"spot_{tag}.npz")
    _, _, D1, D2, F, K, T, B = [int(v) for v in s["dims"]]
    loc = O.synthetic_positions(C, seed=0)
    P = O.seeded_params(C, S, D1, D2, F, K, seed=0, loc=loc)
    X, Y, subj = O.synthetic_batch(B, C, T, F, S, seed=1234)
    temp = torch.tensor([5.1])
    stats = {k: v.clone() for k, v in P.items() if "running" in k or "num_batches" in k}
    loss, Z, logits, grads = O.train_step(P, temp, X, Y, subj, loc=loc,
                                          drop_centre=int(s["drop_centre"]), stats=stats)
    np.testing.assert_allclose(loss.numpy(), s["loss"], rtol=2e-5)
    np.testing.assert_allclose(logits.numpy(), s["logits"], rtol=1e-3, atol=2e-3)
    zv = Z.reshape(-1).numpy()[s["Z@idx"]]
    np.testing.assert_allclose(zv, s["Z@val"], rtol=1e-3, atol=2e-5)
    assert (Z.double() ** 2).sum().item() == pytest.approx(float(s["Z@sumsq"]), rel=1e-4)
    for k, g in grads.items():
        if k == "temp":
            np.testing.assert_allclose(g.numpy(), s["grad/temp"], rtol=1e-3, atol=1e-6)
            continue
        if f"grad/{k}@none" in s:
            assert g is None or float(g.abs().max()) == 0.0
            continue
        flat = torch.view_as_real(g).reshape(-1) if g.is_complex() else g.reshape(-1)
        if null_grad(k):
            assert float(flat.abs().max()) < 1e-4, k
            continue
        ref = s[f"grad/{k}@val"]
        got = flat.numpy()[s[f"grad/{k}@idx"]]
        rms = np.sqrt(float(s[f"grad/{k}@sumsq"]) / flat.numel())
        assert np.abs(got - ref).max() <= 2e-3 * rms + 1e-9, k
    for k in stats:
        if "running" in k:
            np.testing.assert_allclose(stats[k].numpy(), s["after1fwd/" + k], rtol=1e-4, atol=1e-5)


def test_blockwise_loss_oracle_equals_the_pinned_one():
    """oracle.clip_loss_blockwise (the chunked form the full-size loss-block tests need) against clip_loss + autograd, which the
    fixtures above pin to the reference: loss, logits, temperature gradient and embedding gradients, with distinct columns and
    with a repeated column block."""
    g = torch.Generator().manual_seed(3)
    B, N = 12, 700
    Y = torch.randn(B, N, generator=g)
    temp = torch.tensor([2.3])
    for col_src, Zsrc in (([*range(B)], 0.3 * Y + torch.randn(B, N, generator=g)),
                          ([0, 1, 2, 3] * 3, 0.3 * Y[:4] + torch.randn(4, N, generator=g))):
        Z = Zsrc[col_src].clone().requires_grad_(True)
        t = temp.clone().requires_grad_(True)
        loss, logits = O.clip_loss(Y.view(B, 7, 100), Z.view(B, 7, 100), t)
        loss.backward()
        got = O.clip_loss_blockwise(Y, Zsrc, col_src, temp, grad_blocks=[(0, 4), (4, 12)], chunk=256)
        assert abs(float(got["loss"]) - float(loss)) < 1e-6
        np.testing.assert_allclose(got["logits"].numpy(), logits.detach().numpy(), rtol=1e-5, atol=1e-5)
        assert abs(float(got["dtemp"]) - float(t.grad)) < 1e-5
        for (j0, j1), dz in got["dZ"].items():
            np.testing.assert_allclose(dz.numpy(), Z.grad[j0:j1].numpy(), rtol=1e-4, atol=1e-7)
