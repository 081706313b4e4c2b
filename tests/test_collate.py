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
