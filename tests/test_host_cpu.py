"""Host-side logic that needs no GPU: config surface, parameter containers / state_dict compatibility,
initialisation order, sensor layout normalisation, segment construction."""
import os
import warnings

import numpy as np
import pytest
import torch

from oracle import brain_oracle as O
from tests import golden_io as G

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def small_cfg(**kw):
    from speech_decoding_amd import load_config
    ov = ["num_subjects=3", "D1=16", "D2=24", "F=32", "K=4", "preprocs.last4layers=False", "num_channels=12"]
    ov += [f"{k}={v}" for k, v in kw.items()]
    return load_config(overrides=ov)


def test_config_surface_matches_reference_keys():
    from speech_decoding_amd import load_config
    cfg = load_config()
    for key in ["dataset", "rebuild_dataset", "use_wandb", "wandb", "use_sampler", "reproducible", "split_ratio",
                "split_mode", "num_workers", "batch_size", "updates", "lr", "epochs", "reduction", "D1", "D2", "F",
                "K", "d_drop", "init_temperature", "wav2vec_model", "preprocs", "memory_efficient", "hydra"]:
        assert key in cfg, key
    assert (cfg.D1, cfg.D2, cfg.F, cfg.K, cfg.d_drop, cfg.init_temperature) == (270, 320, 512, 32, 0.1, 5.1)
    assert cfg.preprocs["last4layers"] is True and cfg.preprocs.clamp_lim == 20       # item and attribute access
    assert float(cfg.lr) == 3e-4 and cfg.hydra.job.chdir is True
    cfg2 = load_config(overrides=["dataset=Brennan2018", "preprocs.clamp_lim=10", "+num_subjects=5"])
    assert cfg2.dataset == "Brennan2018" and cfg2.preprocs.clamp_lim == 10 and cfg2.num_subjects == 5
    with pytest.raises(ValueError):
        load_config(overrides=["nonsense"])


def test_state_dict_keys_shapes_and_roundtrip():
    from speech_decoding.models import BrainEncoder
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        enc = BrainEncoder(small_cfg())
    shapes = O.param_shapes(12, 3, 16, 24, 32, 4)
    sd = enc.state_dict()
    assert list(sd.keys()) == list(shapes.keys())
    for k, (shape, kind) in shapes.items():
        assert tuple(sd[k].shape) == tuple(shape), k
        assert sd[k].is_complex() == (kind == "complex")
    small = G.load("e2e_small.npz")
    ref_state = G.state_from(small, "init/")
    ref_state.pop("temp")
    missing, unexpected = enc.load_state_dict(ref_state, strict=True)
    assert not missing and not unexpected
    back = enc.state_dict()
    for k, v in ref_state.items():
        assert torch.equal(back[k], v), k
    bad = dict(ref_state)
    bad["subject_block.subject_layer.7.weight"] = bad["subject_block.subject_layer.0.weight"]
    with pytest.raises(RuntimeError):
        enc.load_state_dict(bad, strict=True)
    n_real = sum(p.numel() * (2 if p.is_complex() else 1) for p in enc.parameters())
    assert n_real == sum(int(np.prod(s)) * (2 if kd == "complex" else 1) for s, kd in shapes.values() if kd != "buffer")


def test_full_size_parameter_count_matches_survey():
    """9,565,054 real parameters at C=208 (any C), S=27."""
    shapes = O.param_shapes(208, 27, 270, 320, 1024, 32)
    n = sum(int(np.prod(s)) * (2 if kd == "complex" else 1) for s, kd in shapes.values() if kd != "buffer")
    assert n == 9_565_054


def test_spatial_dropout_mask_and_tables_follow_reference_formulas():
    from speech_decoding_amd.models import SpatialAttention
    cfg = small_cfg()
    loc = O.synthetic_positions(12, seed=7)
    cfg["sensor_positions"] = loc.numpy()
    sa = SpatialAttention(cfg)
    cos, sin = O.fourier_tables(sa.loc, 4)
    assert torch.allclose(sa.cos, cos, atol=1e-6) and torch.allclose(sa.sin, sin, atol=1e-6)
    for centre in range(12):
        assert torch.equal(sa.mask_for(centre), O.dropout_mask(sa.loc, centre, 0.1))
    np.random.seed(5)
    expect = int(np.random.randint(12))
    np.random.seed(5)
    assert torch.equal(sa.draw_mask(), sa.mask_for(expect))          # NumPy global RNG, one draw (models.py:81)
    assert sa.z.dtype == torch.complex64 and float(sa.z.real.min()) >= 0 and float(sa.z.imag.max()) < 1


def test_layout_normalisation():
    from speech_decoding_amd.layout import normalise
    raw = np.random.RandomState(0).rand(30, 2) * 7 - 3
    out = normalise(raw)
    assert out.dtype == torch.float32
    assert float(out.min()) == pytest.approx(0.1) and float(out.max()) == pytest.approx(0.9)
    assert torch.allclose(out, O.normalise_positions(raw))


def test_uniform_and_subject_segments():
    from speech_decoding_amd.engine import EncoderDims, EncoderEngine, block_dilations
    assert [block_dilations(k) for k in range(5)] == [(1, 2, 2), (4, 8, 2), (16, 1, 2), (2, 4, 2), (8, 16, 2)]
    eng = EncoderEngine(EncoderDims(208, 27, 270, 320, 1024, 32))
    for B, ntiles in [(256, 10), (7, 1), (256, 80), (3, 1000)]:
        perm, seg, nseg = eng._uniform_segments(B, ntiles, "cpu")
        seg = seg.numpy()
        assert seg[0] == 0 and seg[-1] == B and len(seg) == nseg + 1 and (np.diff(seg) >= 0).all()
        assert perm is None                                   # consecutive samples: no permutation table
    d = eng.d
    assert (d.Cp, d.D1p, d.D2p, d.F1p, d.Fp) == (256, 320, 320, 640, 1024)


@pytest.mark.parametrize("S,r,B", [(27, 1, 256), (1, 26, 512), (3, 2, 8), (4, 5, 13), (2, 3, 2)])
def test_subject_segments_partition_the_batch_slice_major(S, r, B):
    """Per-subject weight-gradient segments: every sample appears once, segment j*S + s holds only subject s,
    the r slices of a subject cover it in order and differ in size by at most one."""
    from speech_decoding_amd.engine import subject_segments
    rng = np.random.default_rng(S * 100 + r)
    sidx = rng.integers(0, S, size=B).astype(np.int64)
    if S > 2:
        sidx[sidx == 1] = 0                                  # a subject that is absent from the batch
    perm, seg = subject_segments(sidx, S, r)
    assert perm.dtype == np.int32 and seg.dtype == np.int32 and seg.shape == (r * S + 1,)
    assert sorted(perm.tolist()) == list(range(B)) and seg[0] == 0 and seg[-1] == B and (np.diff(seg) >= 0).all()
    for s in range(S):
        members = [perm[seg[j * S + s]: seg[j * S + s + 1]] for j in range(r)]
        for m in members:
            assert (sidx[m] == s).all()
        sizes = [len(m) for m in members]
        assert sum(sizes) == int((sidx == s).sum()) and max(sizes) - min(sizes) <= 1
        joined = np.concatenate(members) if members else np.empty(0, dtype=np.int32)
        assert (np.diff(joined) > 0).all()                   # stable: ascending sample index within a subject


def test_norm_cache_entries_die_with_their_producer_and_on_inplace_edits():
    """ops.ROW_NORMS: norms left by the encoder are served only to the same live buffer at the same version; a freed
    buffer whose address is reused by another tensor must not inherit them (an intermittent wrong-loss bug once)."""
    from speech_decoding_amd.ops import _NormCache
    cache = _NormCache()
    buf = torch.zeros(40, 8)
    norms = torch.arange(4.0)
    cache.put(buf, norms)
    view = buf.detach().as_strided((40, 8), (8, 1), 0)              # what loss.as_rows hands to clip_forward
    assert cache.get(view, 4) is norms
    assert cache.get(view, 5) is None                               # different batch size
    buf[3].add_(1.0)                                                # in-place edit through any view bumps the version
    assert cache.get(view, 4) is None
    cache.put(buf, norms)
    assert cache.get(view, 4) is norms
    ptr = buf.data_ptr()
    del buf, view
    for _ in range(64):                                             # try to get the freed address back
        other = torch.zeros(40, 8)
        if other.data_ptr() == ptr:
            assert cache.get(other, 4) is None
            break


def test_fp16_loss_scale_grows_with_global_batch_and_sequence_length():
    """The gradient entering the encoder shrinks like 1 / (B_global * T): the static fp16 loss scale follows (powers of two), and a
    step whose gradients overflowed is skipped with the scale halved (GradScaler's rule)."""
    from speech_decoding_amd.amp import DEFAULT_FP16_SCALE, LossScaler, fp16_scale_for
    assert fp16_scale_for(256, 360) == DEFAULT_FP16_SCALE == 1024.0
    assert fp16_scale_for(64, 360) == 1024.0                              # never below the calibration point
    assert fp16_scale_for(2048, 360) == 8192.0 and fp16_scale_for(4096, 1000) == 65536.0
    for b, t in [(300, 360), (4096, 360), (512, 1000)]:
        s = fp16_scale_for(b, t)
        assert s >= 1024.0 * (b / 256) * (t / 360) > s / 2 and np.log2(s) == int(np.log2(s))
    assert LossScaler.for_dtype(torch.bfloat16, global_batch=4096, T=1000).scale_value == 1.0
    assert LossScaler.for_dtype(torch.float16).scale_value == 1024.0
    sc = LossScaler.for_dtype(torch.float16, global_batch=4096, T=1000)
    assert sc.scale_value == 65536.0
    loss = torch.tensor(2.0, requires_grad=True)
    assert float(sc.scale(loss)) == 2.0 * 65536.0
    p = torch.nn.Parameter(torch.zeros(3))
    p.grad = torch.tensor([65536.0, float("inf"), 0.0])
    assert sc.unscale_([p], check=True) is False
    sc.update(False)
    assert sc.scale_value == 32768.0 and sc.skipped_steps == 1
    p.grad = torch.tensor([32768.0, 0.0, -32768.0])
    assert sc.unscale_([p], check=True) is True and torch.equal(p.grad, torch.tensor([1.0, 0.0, -1.0]))
    sc.growth_interval = 2
    sc.update(True); sc.update(True)
    assert sc.scale_value == 65536.0
    none = LossScaler(1.0)
    none.update(False)
    assert none.scale_value == 1.0 and none.unscale_([p]) is True


def test_row_statistics_merge_equals_a_global_log_sum_exp():
    """distributed.combine_row_stats — the arithmetic behind the one small all-gather of the data-parallel loss: per-rank (max,
    sum exp(l - max), positives' logits) over column blocks merge into the row lse over ALL columns and the full diagonal."""
    from speech_decoding_amd.distributed import combine_row_stats
    g = torch.Generator().manual_seed(0)
    B, world = 24, 4
    logits = torch.randn(B, B, generator=g, dtype=torch.float64) * 5
    per = B // world
    stats = []
    for r in range(world):
        blk = logits[:, r * per: (r + 1) * per]
        mx = blk.max(dim=1).values
        diag = torch.zeros(B, dtype=torch.float64)
        diag[r * per: (r + 1) * per] = logits.diagonal()[r * per: (r + 1) * per]
        stats.append(torch.stack([mx, torch.exp(blk - mx[:, None]).sum(dim=1), diag]))
    lse, diag = combine_row_stats(torch.stack(stats))
    assert torch.allclose(lse, torch.logsumexp(logits, dim=1), rtol=0, atol=1e-12)
    assert torch.equal(diag, logits.diagonal())
    assert torch.allclose(combine_row_stats(torch.stack(stats)[:, :2]), lse)          # without the diagonal: the lse alone


def test_loss_scaler_guards_fp16_whatever_the_scale_and_never_grows_below_its_start():
    """amp.LossScaler: the overflow guard belongs to the DTYPE (fp16), not to the scale's value — a scale halved down to 1 (or
    fp16_loss_scale=1) still checks for inf / NaN; growth never caps below the calibrated start (131072 at 8192 x 1000)."""
    from speech_decoding_amd.amp import LossScaler, fp16_scale_for
    s = LossScaler.for_dtype(torch.float16, global_batch=8192, T=1000)
    assert s.enabled and s.scale_value == fp16_scale_for(8192, 1000) == 131072.0
    s.growth_interval = 1
    s.update(True)
    assert s.scale_value == 262144.0                      # grows past 65536, never below the start
    for _ in range(40):
        s.update(True)
    assert s.scale_value == 2.0 ** 24
    p = torch.nn.Parameter(torch.ones(4))
    one = LossScaler(1.0, enabled=True)                   # fp16 with the scale at 1: the check still runs
    p.grad = torch.tensor([1.0, float("nan"), 3.0, 4.0])
    assert one.unscale_([p], check=True) is False
    one.update(False)
    assert one.scale_value == 1.0 and one.skipped_steps == 1
    p.grad = torch.tensor([2.0, 4.0, float("inf"), 8.0])
    s4 = LossScaler(4.0)
    assert s4.unscale_([p], check=True) is False
    p.grad = torch.tensor([2.0, 4.0, 6.0, 8.0])
    z = torch.nn.Parameter(torch.ones(2, dtype=torch.complex64))
    z.grad = torch.full((2,), 8 + 4j, dtype=torch.complex64)
    assert s4.unscale_([p, z], check=True) is True
    assert p.grad.tolist() == [0.5, 1.0, 1.5, 2.0] and z.grad.tolist() == [2 + 1j, 2 + 1j]
    off = LossScaler.for_dtype(torch.bfloat16)
    assert not off.enabled and off.scale(p.grad) is p.grad and off.unscale_([p], check=True) is True
    off.update(False)
    assert off.scale_value == 1.0


def test_sharded_random_sampler_is_the_global_sampler_cut_into_rank_slices():
    """get_dataloaders.py:57-62 (RandomSampler(replacement=True, num_samples=updates * batch_size)) consumed in batches: the
    concatenation of the ranks' slices is the single-process batch, for every update, whatever the world size."""
    from speech_decoding_amd.data import ShardedRandomSampler
    whole = [b.tolist() for b in ShardedRandomSampler(97, 24, 5, 0, 1, seed=11)]
    assert len(whole) == 5 and all(len(b) == 24 and max(b) < 97 for b in whole)
    for world in (2, 3, 4):
        parts = [[b.tolist() for b in ShardedRandomSampler(97, 24, 5, r, world, seed=11)] for r in range(world)]
        assert [sum((parts[r][u] for r in range(world)), []) for u in range(5)] == whole
    assert any(len(set(b)) < len(b) for b in [x.tolist() for x in ShardedRandomSampler(10, 8, 20, seed=1)])     # with replacement
    norep = [b.tolist() for b in ShardedRandomSampler(30, 8, 4, seed=1, replacement=False)]
    assert all(len(set(b)) == 8 for b in norep)
    import pytest
    with pytest.raises(ValueError):
        ShardedRandomSampler(10, 9, 1, 0, 2)


def _toy_feed(monkeypatch, seed=5):
    """ResidentSegmentFeed with the device kernel replaced by a recorder (host logic only)."""
    from speech_decoding_amd import data as D

    class FakeSegments:
        def __init__(self, sessions, *a, **k):
            self.sessions = sessions
            self.calls = []

        def batch(self, rec, on):
            self.calls.append((np.asarray(rec).copy(), np.asarray(on).copy()))
            return torch.zeros(len(rec), 2, 3)
    monkeypatch.setattr(D, "ResidentSegments", FakeSegments)
    n_task, per = 4, 10
    onsets = [np.arange(per, dtype=np.int64) * 5 + 7 * (r % 2) for r in range(2 * n_task)]
    return D.ResidentSegmentFeed([None] * (2 * n_task), list(range(2 * n_task)), np.repeat(np.arange(n_task), 2), onsets,
                                 np.repeat(np.arange(n_task), per), np.tile(np.arange(per), n_task), torch.zeros(n_task * per, 2, 3),
                                 seq_len_samp=3, baseline_len_samp=1, clamp_lim=20.0, seed=seed)


def test_feed_vectorised_draw_equals_item_by_item_choice(monkeypatch):
    """gwilliams2022.py:133 draws one recording per item; the feed draws a batch at once — the same stream of values."""
    feed = _toy_feed(monkeypatch, seed=5)
    twin = np.random.RandomState(5)
    for idx in (np.array([3, 17, 17, 39, 0, 21]), np.arange(40)):
        rec = feed.draw_recordings(idx)
        want = [int(twin.choice(feed.by_task[int(feed.seg_task[i])])) for i in idx]
        assert rec.tolist() == want
        on = feed._onset_of(rec, feed.seg_in_task[idx])
        assert on.tolist() == [int(feed.onsets[r][feed.seg_in_task[i]]) for r, i in zip(rec, idx)]


def test_feed_rank_shards_union_to_the_single_process_batch(monkeypatch):
    """Data parallelism: every rank draws the recordings of the GLOBAL batch from the same generator state and keeps its slice —
    the union over ranks is the batch one process would have made, and no two ranks share a draw."""
    from speech_decoding_amd.data import ShardedRandomSampler
    single = _toy_feed(monkeypatch, seed=9)
    s1 = ShardedRandomSampler(40, 8, 5, rank=0, world=1, seed=3)
    list(single.batches(s1))
    ref = single.rs.calls
    parts = []
    for rank in range(2):
        f = _toy_feed(monkeypatch, seed=9)
        list(f.batches(ShardedRandomSampler(40, 8, 5, rank=rank, world=2, seed=3)))
        parts.append(f.rs.calls)
    for b in range(5):
        rec = np.concatenate([parts[0][b][0], parts[1][b][0]])
        on = np.concatenate([parts[0][b][1], parts[1][b][1]])
        assert rec.tolist() == ref[b][0].tolist() and on.tolist() == ref[b][1].tolist()
