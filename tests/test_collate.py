"""Batch collate: oracle vs the reference's own preproc_utils output (CPU), HIP kernel vs both (GPU)."""
import numpy as np
import pytest
import torch

from oracle import brain_oracle as O
from tests import golden_io as G


def test_oracle_collate_matches_reference_fixture():
    c = G.load("collate.npz")
    X = torch.from_numpy(c["X"])
    got = O.collate_batch(X, int(c["baseline_len"]), float(c["clamp_lim"]), True)
    np.testing.assert_allclose(got.numpy(), c["out"], rtol=1e-5, atol=1e-5)
    got = O.collate_batch(X, int(c["baseline_len"]), float(c["clamp_lim"]), False)
    np.testing.assert_allclose(got.numpy(), c["out_noclamp"], rtol=1e-5, atol=1e-4)
    assert float(np.abs(c["out"]).max()) == 20.0 and float(np.abs(c["out_noclamp"]).max()) > 20.0
    assert float(np.abs(c["out"][1, 2]).max()) == 0.0            # constant row: scale falls back to 1


@pytest.mark.gpu
def test_hip_collate_matches_reference_fixture_and_oracle():
    from speech_decoding_amd.collate import Gwilliams2022Collator, robust_scale_clamp
    c = G.load("collate.npz")
    X = torch.from_numpy(c["X"]).to("cuda:0")
    got = robust_scale_clamp(X, int(c["baseline_len"]), float(c["clamp_lim"]), True).cpu()
    np.testing.assert_allclose(got.numpy(), c["out"], rtol=1e-5, atol=1e-5)
    got = robust_scale_clamp(X, int(c["baseline_len"]), float(c["clamp_lim"]), False).cpu()
    np.testing.assert_allclose(got.numpy(), c["out_noclamp"], rtol=1e-5, atol=1e-4)
    # config-2 sized batch and the T = 1000 path (config 5), against the oracle
    for (B, C, T, nb) in [(64, 208, 360, 60), (3, 306, 1000, 100), (2, 5, 61, 7)]:
        g = torch.Generator().manual_seed(T)
        Xb = torch.randn(B, C, T, generator=g) * 5 + torch.randn(B, C, 1, generator=g)
        want = O.collate_batch(Xb, nb, 20.0, True)
        got = robust_scale_clamp(Xb.to("cuda:0"), nb, 20.0, True).cpu()
        np.testing.assert_allclose(got.numpy(), want.numpy(), rtol=1e-5, atol=2e-5)
    # the Collator interface of gwilliams2022.py:640-661
    from speech_decoding_amd import Config
    args = Config.wrap({"preprocs": {"brain_resample_rate": 120, "baseline_len_sec": 0.5, "clamp": True, "clamp_lim": 20}})
    col = Gwilliams2022Collator(args, device="cuda:0")
    batch = [(torch.from_numpy(c["X"][i]), torch.zeros(4, 360), i % 3) for i in range(5)]
    Xc, Yc, sidx = col(batch)
    np.testing.assert_allclose(Xc.cpu().numpy(), c["out"], rtol=1e-5, atol=1e-5)
    assert Yc.shape == (5, 4, 360) and sidx.dtype == torch.int32 and sidx.tolist() == [0, 1, 2, 0, 1]


@pytest.mark.gpu
def test_resident_segment_gather_fused_with_collate():
    """gwilliams2022.py:129-142 window extraction + collate in one kernel vs slicing on the host + oracle."""
    from speech_decoding_amd.collate import ResidentSegments
    g = torch.Generator().manual_seed(9)
    sessions = [torch.randn(13, L, generator=g) * 2 + 0.5 for L in (5000, 7321, 4100)]
    T, nb = 360, 60
    rs = ResidentSegments([s.to("cuda:0") for s in sessions], T, nb, 20.0, True)
    sidx = [2, 0, 1, 1, 0, 2, 2]
    onsets = [0, 4640, 17, 6961, 1000, 3740, 123]
    got = rs.batch(sidx, onsets).cpu()
    want = O.collate_batch(torch.stack([sessions[s][:, o:o + T] for s, o in zip(sidx, onsets)]), nb, 20.0, True)
    np.testing.assert_allclose(got.numpy(), want.numpy(), rtol=1e-5, atol=2e-5)
    with pytest.raises(IndexError):
        rs.batch([0], [4700])


# --------------------------------------------------------------------------------------------------------------------------
# Randomised property test of the sorting network (csrc/collate.hip: all-ascending bitonic network over registers x lanes,
# v_med3 against a per-lane infinity, DPP mirrors, ds_bpermute) against np.percentile (the oracle's RobustScaler,
# preproc_utils.py:69-90): every template boundary (T <= 512 sorts 8 values per lane, above 16), both access widths (16-byte
# when T % 4 == 0 and the source is aligned, 4-byte otherwise), ties, constant rows, baseline windows from 0 to T.
# --------------------------------------------------------------------------------------------------------------------------
PROPERTY_T = [2, 3, 63, 64, 65, 127, 359, 360, 511, 512, 513, 516, 1000, 1023, 1024]


def _rows_for(T, kind, g):
    """(R, T) fp32 rows of one flavour."""
    R = 37
    if kind == "normal":
        return torch.randn(R, T, generator=g) * 4 + torch.randn(R, 1, generator=g) * 3
    if kind == "ties":                 # values from a set of five: long runs of equal values around every quantile position
        return torch.randint(-2, 3, (R, T), generator=g).float()
    if kind == "two":                  # two values only: the quartiles fall inside a run or on its edge
        return (torch.rand(R, T, generator=g) < torch.rand(R, 1, generator=g)).float() * 7 - 3
    if kind == "constant":
        return torch.randn(R, 1, generator=g).expand(R, T).contiguous()
    if kind == "sorted":               # already ascending / descending rows
        x = torch.sort(torch.randn(R, T, generator=g), dim=1).values
        x[::2] = x[::2].flip(1)
        return x
    if kind == "outliers":             # a few huge values: the clamp, and quartiles that ignore them
        x = torch.randn(R, T, generator=g)
        x[:, :: max(1, T // 7)] *= 1e4
        return x
    raise ValueError(kind)


@pytest.mark.gpu
@pytest.mark.parametrize("T", PROPERTY_T)
def test_collate_rows_property_against_numpy_percentile(T):
    from speech_decoding_amd.collate import robust_scale_clamp
    g = torch.Generator().manual_seed(1000 + T)
    for kind in ("normal", "ties", "two", "constant", "sorted", "outliers"):
        X = _rows_for(T, kind, g)
        for nb in sorted({1, max(1, T // 6), T}):      # (nb = 0: the reference's mean of an empty window is NaN)
            for clamp in (True, False):
                want = O.collate_batch(X[None], nb, 20.0, clamp)[0].numpy()
                got = robust_scale_clamp(X[None].to("cuda:0"), nb, 20.0, clamp)[0].cpu().numpy()
                # fp32 on both sides: a row whose baseline-corrected values are large against its inter-quartile range loses
                # |x - base| * 2^-23 to cancellation in (x - median) before the division (outliers inside the baseline window)
                v = X.double() - X[:, :nb].double().mean(dim=1, keepdim=True)
                q = np.percentile(v.numpy(), [25.0, 75.0], axis=1)
                iqr = np.where(q[1] - q[0] == 0, 1.0, q[1] - q[0])
                k = (1e-6 * v.abs().max(dim=1).values.numpy() / iqr)[:, None]       # rounding of median and IQR, relative to the IQR
                tol = 3e-5 + 2e-5 * np.abs(want) + k * (1.0 + np.abs(want))
                err = np.abs(got - want)
                assert (err <= tol).all(), f"T={T} {kind} nb={nb} clamp={clamp}: worst {float((err - tol).max()):.3e} over its bound"


@pytest.mark.gpu
@pytest.mark.parametrize("T", [2, 61, 64, 360, 511, 512, 513, 1000, 1024])
def test_collate_windows_property_unaligned_sources(T):
    """The fused window gather: windows that start at ANY sample of a resident recording (4-byte accesses unless the window
    happens to be 16-byte aligned and T % 4 == 0), recordings of different lengths, against slicing on the host + oracle."""
    from speech_decoding_amd.collate import ResidentSegments
    g = torch.Generator().manual_seed(77 + T)
    C = 7
    sessions = [torch.randn(C, L, generator=g) * 3 + torch.randn(C, 1, generator=g) for L in (2 * T + 5, 3 * T + 64, T)]
    sessions[1][2] = torch.randint(-1, 2, (sessions[1].shape[1],), generator=g).float()       # a channel of ties
    sessions[0][4] = 1.25                                                                       # a constant channel
    nb = max(1, T // 6)
    rs = ResidentSegments([s.to("cuda:0") for s in sessions], T, nb, 20.0, True)
    sidx = [0, 0, 0, 0, 1, 1, 1, 1, 1, 2]
    onsets = [0, 1, 2, T + 5, 3, 4, 64, 2 * T + 63, 2 * T + 64, 0]
    got = rs.batch(sidx, onsets).cpu()
    want = O.collate_batch(torch.stack([sessions[s][:, o:o + T] for s, o in zip(sidx, onsets)]), nb, 20.0, True)
    np.testing.assert_allclose(got.numpy(), want.numpy(), rtol=2e-5, atol=3e-5)


@pytest.mark.gpu
def test_collate_rows_with_infinite_samples():
    """Defined behaviour for non-finite input (the reference's sklearn / numpy path yields NaN for a whole row once a
    non-finite value reaches its baseline mean or a quartile): an isolated +-inf OUTSIDE the baseline window and beyond the
    quartile positions is an outlier like any other — clamped, equal to the reference; a row whose baseline window holds an
    inf has unspecified values in THAT row only.  No other row is ever affected (one wavefront sorts one row; the sort's
    padding value is +inf too)."""
    from speech_decoding_amd.collate import robust_scale_clamp
    for T in (360, 1000):
        g = torch.Generator().manual_seed(T)
        X = torch.randn(12, T, generator=g) * 2
        clean = O.collate_batch(X[None], 60, 20.0, True)[0].numpy()
        Xi = X.clone()
        Xi[3, 200] = float("inf")
        Xi[5, 100] = float("-inf")
        Xi[7, 70], Xi[7, 300] = float("inf"), float("-inf")
        Xi[9, 10] = float("inf")                     # inside the baseline window: row 9 is unspecified
        got = robust_scale_clamp(Xi[None].to("cuda:0"), 60, 20.0, True)[0].cpu().numpy()
        want = O.collate_batch(Xi[None], 60, 20.0, True)[0].numpy()
        for r in (3, 5, 7):
            assert np.isfinite(want[r]).all()
            np.testing.assert_allclose(got[r], want[r], rtol=2e-5, atol=3e-5)
        assert got[3, 200] == 20.0 and got[5, 100] == -20.0 and got[7, 70] == 20.0 and got[7, 300] == -20.0
        others = [r for r in range(12) if r not in (3, 5, 7, 9)]
        np.testing.assert_allclose(got[others], clean[others], rtol=2e-5, atol=3e-5)
