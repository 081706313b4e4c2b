"""Kernel-level parity on a real MI355X: every C-ABI stage against the CPU oracle / plain torch fp32
on the same seeded inputs.  fp32 mode runs on the exact-f32 MFMA (tolerance 1e-5..1e-4 relative);
bf16 mode is checked against the same fp32 expectation at bf16 resolution (2^-8 relative per operand)."""
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as TF

pytestmark = pytest.mark.gpu

from oracle import brain_oracle as O   # noqa: E402

DEV = "cuda:0"
DTYPES = [torch.float32, torch.bfloat16, torch.float16]


@pytest.fixture(scope="module")
def ops():
    from speech_decoding_amd import ops as _ops
    from speech_decoding_amd import lib
    lib.load()
    assert torch.cuda.is_available(), "these tests need the MI355X"
    return _ops


def tol(dtype, k_terms=1):
    if dtype == torch.float32:
        return dict(rtol=2e-5, atol=2e-5)
    if dtype == torch.float16:         # 2^-11 per stored value (bf16: 2^-8)
        return dict(rtol=2.5e-3, atol=2.5e-3 * math.sqrt(max(1, k_terms)) / 8)
    return dict(rtol=2e-2, atol=2e-2 * math.sqrt(max(1, k_terms)) / 8)


def to_rows(ops, x, dtype, Cp=None):
    from speech_decoding_amd import lib as L
    B, C, T = x.shape
    buf = ops.new_rows(B, T, Cp or L.pad_channels(C), dtype, DEV)
    ops.pack_rows(x.to(DEV), buf)
    return buf


def from_rows(ops, buf, B, C, T):
    return ops.unpack_rows(buf, B, C, T).cpu()


def rel_err(got, ref):
    return float((got - ref).abs().max() / (ref.abs().max() + 1e-12))


def q(x, dtype):
    """quantise an fp32 tensor to the compute dtype (what the kernels will actually read)"""
    return x.to(dtype).float()


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("C,T,misalign", [(70, 150, False), (70, 152, False), (208, 360, False), (70, 152, True), (1024, 68, False)])
def test_pack_unpack_roundtrip_and_padding(ops, dtype, C, T, misalign):
    """T % 4 == 0 with an aligned source takes the 16-byte kernel, everything else the element-wise one: same results."""
    from speech_decoding_amd import lib as L
    g = torch.Generator().manual_seed(0)
    B = 3
    x = torch.randn(B, C, T, generator=g)
    if misalign:                        # a source that starts 4 bytes into an allocation
        flat = torch.empty(B * C * T + 1, device=DEV)
        flat[1:].copy_(x.reshape(-1))
        src = flat[1:].view(B, C, T)
        assert src.data_ptr() % 16 == 4
        buf = ops.new_rows(B, T, L.pad_channels(C), dtype, DEV)
        ops.pack_rows(src, buf)
    else:
        buf = to_rows(ops, x, dtype)
    back = from_rows(ops, buf, B, C, T)
    assert torch.equal(back, q(x, dtype))
    full = buf.float().cpu()
    Tp = L.rows_tp(T)
    valid = torch.zeros(full.shape[0], dtype=torch.bool)
    for b in range(B):
        valid[b * Tp + L.ROW_PAD: b * Tp + L.ROW_PAD + T] = True
    assert float(full[~valid].abs().max()) == 0.0          # pad rows untouched
    if full.shape[1] > C:
        assert float(full[:, C:].abs().max()) == 0.0       # pad channels zero
    view = ops.rows_view(buf, B, C, T).float().cpu()
    assert torch.equal(view, q(x, dtype))


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("tiling", [4096, 8192, 16384])   # one tile / two tiles per workgroup / 256-row flat tiles
@pytest.mark.parametrize("cin,cout,dil,T,B", [(40, 48, 1, 40, 3), (270, 320, 2, 300, 3), (320, 320, 16, 360, 3),
                                              (96, 128, 8, 130, 2), (64, 640, 4, 129, 3), (320, 320, 4, 360, 37),
                                              (64, 160, 16, 1000, 5)])
def test_conv3_forward_bias_residual_stats(ops, dtype, cin, cout, dil, T, tiling, B):
    from speech_decoding_amd import lib as L
    g = torch.Generator().manual_seed(cin + cout + dil)
    x = q(torch.randn(B, cin, T, generator=g), dtype)
    w = q(torch.randn(cout, cin, 3, generator=g) / math.sqrt(3 * cin), dtype)
    bias = torch.randn(cout, generator=g)
    use_res = cin == cout
    ref = TF.conv1d(x, w, bias, padding=dil, dilation=dil)
    if use_res:
        ref = ref + x
    xb = to_rows(ops, x, dtype)
    Cin_p, Cout_p = L.pad_channels(cin), L.pad_channels(cout)
    wp = ops.pack_conv_weight(w.to(DEV), Cout_p, Cin_p, dtype)
    yb = ops.new_rows(B, T, Cout_p, dtype, DEV)
    stats = torch.full((ops.conv_stats_rows(B, T, 3, Cout_p, tiling), 2, Cout_p), float("nan"), device=DEV)   # every row must be written
    ops.conv_gemm(xb, wp, yb, B=B, T=T, KS=3, dil=dil, bias=ops.pack_vector(bias.to(DEV), Cout_p),
                  res=xb if use_res else None, stats=stats, flags=tiling)
    got = from_rows(ops, yb, B, cout, T)
    np.testing.assert_allclose(got.numpy(), ref.numpy(), **tol(dtype, 3 * cin))
    # pad rows / channels of the output stay zero
    full = yb.float().cpu()
    if cout < Cout_p:
        assert float(full[:, cout:].abs().max()) == 0.0
    assert float(full[: L.ROW_PAD].abs().max()) == 0.0
    # BatchNorm partial statistics: sums over valid rows of the stored values
    s = stats.sum(dim=0).cpu()
    np.testing.assert_allclose(s[0, :cout].numpy(), got.sum(dim=(0, 2)).numpy(), rtol=1e-4, atol=1e-3)
    np.testing.assert_allclose(s[1, :cout].numpy(), (got ** 2).sum(dim=(0, 2)).numpy(), rtol=1e-4, atol=1e-3)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("cin,half,dil,T,B", [(320, 320, 4, 360, 5), (64, 300, 2, 129, 3), (96, 600, 16, 200, 2)])
def test_conv3_glu_epilogue_equals_conv_then_glu(ops, dtype, cin, half, dil, T, B):
    """F.glu in the conv's epilogue (EPI_GLU, weights packed 80 values + 80 gates per tile) is BIT-equal to the conv
    followed by glu_forward; the gate it keeps is the conv's gate half; the (out, gate) backward matches the [a | g] one."""
    from speech_decoding_amd import lib as L
    g = torch.Generator().manual_seed(cin + half + dil)
    x = q(torch.randn(B, cin, T, generator=g), dtype)
    w = q(torch.randn(2 * half, cin, 3, generator=g) / math.sqrt(3 * cin), dtype)
    bias = torch.randn(2 * half, generator=g)
    Cin_p, Hp = L.pad_channels(cin), L.pad_channels(half)
    assert Hp % 80 == 0
    xb = to_rows(ops, x, dtype)
    w_d, b_d = w.to(DEV), bias.to(DEV)
    packs = {}
    for name, tile in (("halves", 0), ("tiles", 80)):
        plan = ops.PackPlan(dtype, DEV)
        plan.add_weight("w", w_d, 2 * Hp, Cin_p, glu_half=half, glu_half_p=Hp, glu_tile=tile)
        plan.add_vector("b", b_d, 2 * Hp, glu_half=half, glu_half_p=Hp, glu_tile=tile)
        packs[name] = dict(plan.run({"w": w_d, "b": b_d}))
    flat = L.CONV_FLAT_TILES
    c2 = ops.conv_gemm(xb, packs["halves"]["w"], ops.new_rows(B, T, 2 * Hp, dtype, DEV), B=B, T=T, KS=3, dil=dil,
                       bias=packs["halves"]["b"], flags=flat)
    out_ref = ops.glu_forward(c2, ops.new_rows(B, T, Hp, dtype, DEV), B, T)
    out, gate = ops.new_rows(B, T, Hp, dtype, DEV), ops.new_rows(B, T, Hp, dtype, DEV)
    ops.conv_gemm(xb, packs["tiles"]["w"], out, B=B, T=T, KS=3, dil=dil, bias=packs["tiles"]["b"], y_pre=gate,
                  flags=flat | L.EPI_GLU)
    assert torch.equal(out, out_ref)
    assert torch.equal(gate, c2[:, Hp:].contiguous())
    ref = TF.glu(TF.conv1d(x, w, bias, padding=dil, dilation=dil), dim=1)
    np.testing.assert_allclose(from_rows(ops, out, B, half, T).numpy(), ref.numpy(), **tol(dtype, 3 * cin))
    # without a gate buffer (inference)
    out2 = ops.new_rows(B, T, Hp, dtype, DEV)
    ops.conv_gemm(xb, packs["tiles"]["w"], out2, B=B, T=T, KS=3, dil=dil, bias=packs["tiles"]["b"], flags=flat | L.EPI_GLU)
    assert torch.equal(out2, out)
    # backward from (out, gate) against the [value | gate] form
    dy = to_rows(ops, q(torch.randn(B, half, T, generator=g), dtype), dtype)
    scratch = ops.reduce_scratch(2 * Hp, DEV)
    dx_ref, dx = ops.new_rows(B, T, 2 * Hp, dtype, DEV), ops.new_rows(B, T, 2 * Hp, dtype, DEV)
    cs_ref = ops.glu_backward_colsum(c2, dy, dx_ref, B, T, scratch)
    cs = ops.glu_backward_colsum_og(out, gate, dy, dx, B, T, scratch)
    assert torch.equal(dx[:, :Hp], dx_ref[:, :Hp])                      # d value = dy * sigmoid(gate): same expression
    t = tol(dtype, 4)
    np.testing.assert_allclose(dx[:, Hp:].float().cpu().numpy(), dx_ref[:, Hp:].float().cpu().numpy(), **t)
    ts = tol(dtype, B * T)
    np.testing.assert_allclose(cs.cpu().numpy(), cs_ref.cpu().numpy(), rtol=ts["rtol"], atol=ts["atol"])
    # EPI_GLU is refused where the flat-tile kernel does not apply
    with pytest.raises(L.SdaError):
        ops.conv_gemm(xb, packs["tiles"]["w"], out2, B=B, T=T, KS=3, dil=dil, bias=packs["tiles"]["b"], flags=L.EPI_GLU)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("KS,cin,H,dil,T,B", [(3, 320, 320, 4, 200, 3), (1, 640, 320, 0, 129, 2), (3, 96, 64, 2, 77, 3)])
def test_conv_glu_backward_epilogue(ops, dtype, KS, cin, H, dil, T, B):
    """EPI_GLU_BWD: a (data-gradient) conv whose output is the gradient entering an F.glu writes the GLU backward
    [dy * sig(g) | dy * out * (1 - sig(g))] and the per-tile column sums of both halves, against conv -> glu_backward_colsum_og."""
    from speech_decoding_amd import lib as L
    g = torch.Generator().manual_seed(KS + cin + H)
    x = q(torch.randn(B, cin, T, generator=g), dtype)
    w = q(torch.randn(H, cin, KS, generator=g) / math.sqrt(KS * cin), dtype)
    res = q(torch.randn(B, H, T, generator=g), dtype)
    out = q(torch.randn(B, H, T, generator=g), dtype)
    gate = q(torch.randn(B, H, T, generator=g), dtype)
    Cin_p, Hp = L.pad_channels(cin), L.pad_channels(H)
    xb, resb, outb, gateb = (to_rows(ops, t, dtype) for t in (x, res, out, gate))
    wp = ops.pack_conv_weight(w.to(DEV), Hp, Cin_p, dtype)
    # reference path: the conv stores dy, a separate pass applies the GLU backward and sums the columns
    dy = ops.conv_gemm(xb, wp, ops.new_rows(B, T, Hp, dtype, DEV), B=B, T=T, KS=KS, dil=dil, res=resb)
    scratch = ops.reduce_scratch(2 * Hp, DEV)
    want = ops.new_rows(B, T, 2 * Hp, dtype, DEV)
    cs_want = ops.glu_backward_colsum_og(outb, gateb, dy, want, B, T, scratch)
    got = ops.new_rows(B, T, 2 * Hp, dtype, DEV)
    stats = torch.full((ops.conv_stats_rows(B, T, KS, Hp, 0), 2, Hp), float("nan"), device=DEV)
    ops.conv_gemm(xb, wp, got, B=B, T=T, KS=KS, dil=dil, res=resb, stats=stats, glu_bwd=(outb, gateb))
    t = tol(dtype, KS * cin)       # (the fused form multiplies the unrounded dy: one rounding less than the reference path)
    np.testing.assert_allclose(got.float().cpu().numpy(), want.float().cpu().numpy(), **t)
    cs = ops.reduce_stats(stats)
    ts = tol(dtype, B * T)
    np.testing.assert_allclose(cs.cpu().numpy(), cs_want.cpu().numpy(), rtol=ts["rtol"], atol=ts["atol"] * 4)
    assert float(got[: L.ROW_PAD].float().abs().max()) == 0.0                 # pad rows untouched
    with pytest.raises(L.SdaError):                                           # not with the flat-tile kernel
        ops.conv_gemm(xb, wp, got, B=B, T=T, KS=KS, dil=dil, stats=stats, glu_bwd=(outb, gateb), flags=L.CONV_FLAT_TILES)


@pytest.mark.parametrize("dtype", DTYPES)
def test_conv1_per_sample_weights_and_gelu(ops, dtype):
    from speech_decoding_amd import lib as L
    g = torch.Generator().manual_seed(3)
    B, C, T, S = 5, 72, 200, 4
    x = q(torch.randn(B, C, T, generator=g), dtype)
    w = q(torch.randn(S, C, C, 1, generator=g) / math.sqrt(C), dtype)
    idx = torch.tensor([2, 0, 3, 3, 1], dtype=torch.int32)
    ref = torch.bmm(w[idx.long(), :, :, 0], x)
    Cp = L.pad_channels(C)
    xb = to_rows(ops, x, dtype)
    wp = ops.pack_conv_weight(w.to(DEV), Cp, Cp, dtype)
    yb = ops.new_rows(B, T, Cp, dtype, DEV)
    ops.conv_gemm(xb, wp, yb, B=B, T=T, KS=1, dil=0, widx=idx.to(DEV))
    np.testing.assert_allclose(from_rows(ops, yb, B, C, T).numpy(), ref.numpy(), **tol(dtype, C))
    # GELU epilogue with the pre-activation saved
    bias = torch.randn(C, generator=g)
    w1 = w[0]
    ref_pre = TF.conv1d(x, w1, bias)
    pre, post = ops.new_rows(B, T, Cp, dtype, DEV), ops.new_rows(B, T, Cp, dtype, DEV)
    ops.conv_gemm(xb, ops.pack_conv_weight(w1.to(DEV), Cp, Cp, dtype), post, B=B, T=T, KS=1, dil=0,
                  bias=ops.pack_vector(bias.to(DEV), Cp), y_pre=pre, gelu=True)
    np.testing.assert_allclose(from_rows(ops, pre, B, C, T).numpy(), ref_pre.numpy(), **tol(dtype, C))
    np.testing.assert_allclose(from_rows(ops, post, B, C, T).numpy(), TF.gelu(ref_pre).numpy(), **tol(dtype, C))


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("cin,cout,T,B", [(320, 640, 360, 5), (640, 1024, 129, 3), (64, 160, 40, 7), (100, 128, 77, 2),
                                          (1024, 640, 200, 3), (192, 384, 1, 9)])
def test_conv1_flat_tiles_equal_tile_kernel(ops, dtype, cin, cout, T, B):
    """conv1_flat (SDA_CONV_FLAT_TILES, kernel size 1) against conv_gemm's tile-per-workgroup kernel: same contraction order,
    so bit-equal; plain, and with bias + saved pre-activation + GELU (conv_final1/2, models.py:194-195)."""
    from speech_decoding_amd import lib as L
    g = torch.Generator().manual_seed(11)
    Cip, Cop = L.pad_channels(cin), L.pad_channels(cout)
    x = q(torch.randn(B, cin, T, generator=g), dtype)
    w = q(torch.randn(cout, cin, 1, generator=g) / math.sqrt(cin), dtype)
    bias = ops.pack_vector(torch.randn(cout, generator=g).to(DEV), Cop)
    xb = to_rows(ops, x, dtype)
    wp = ops.pack_conv_weight(w.to(DEV), Cop, Cip, dtype)
    if Cop % 160 and Cop % 128:
        pytest.skip("no flat tiling for this width")
    for kw in (dict(), dict(bias=bias, gelu=True, with_pre=True)):
        outs = []
        for flags in (0, L.CONV_FLAT_TILES):
            kw2 = dict(kw)
            y, pre = ops.new_rows(B, T, Cop, dtype, DEV), ops.new_rows(B, T, Cop, dtype, DEV)
            if kw2.pop("with_pre", False):
                kw2["y_pre"] = pre
            ops.conv_gemm(xb, wp, y, B=B, T=T, KS=1, dil=0, flags=flags, **kw2)
            outs.append((y, pre))
        assert torch.equal(outs[0][0], outs[1][0])
        assert torch.equal(outs[0][1], outs[1][1])
    ref = TF.conv1d(x, w)
    np.testing.assert_allclose(from_rows(ops, ops.conv_gemm(xb, wp, ops.new_rows(B, T, Cop, dtype, DEV), B=B, T=T, KS=1, dil=0,
                                                            flags=L.CONV_FLAT_TILES), B, cout, T).numpy(), ref.numpy(), **tol(dtype, cin))


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("cin,cout,T,B", [(1024, 640, 360, 4), (128, 256, 77, 3), (64, 320, 130, 6)])
def test_conv1_flat_gelu_backward_epilogue(ops, dtype, cin, cout, T, B):
    """SDA_EPI_GELU_BWD: the data-gradient conv's epilogue multiplies by GELU'(u) and sums the columns = conv, store, then
    sda_gelu_backward_colsum (bit-equal gradient; the column sums agree to summation order)."""
    from speech_decoding_amd import lib as L
    g = torch.Generator().manual_seed(12)
    Cip, Cop = L.pad_channels(cin), L.pad_channels(cout)
    xb = to_rows(ops, q(torch.randn(B, cin, T, generator=g), dtype), dtype)
    wp = ops.pack_conv_weight(q(torch.randn(cout, cin, 1, generator=g) / math.sqrt(cin), dtype).to(DEV), Cop, Cip, dtype)
    ub = to_rows(ops, q(torch.randn(B, cout, T, generator=g), dtype), dtype)
    dg = ops.conv_gemm(xb, wp, ops.new_rows(B, T, Cop, dtype, DEV), B=B, T=T, KS=1, dil=0)
    du_ref = ops.new_rows(B, T, Cop, dtype, DEV)
    cs_ref = ops.gelu_backward_colsum(ub, dg, du_ref, B, T, ops.reduce_scratch(Cop, DEV))
    du = ops.new_rows(B, T, Cop, dtype, DEV)
    st = torch.full((ops.conv_stats_rows(B, T, 1, Cop, L.CONV_FLAT_TILES | L.EPI_GELU_BWD), 2, Cop), float("nan"), device=DEV)
    ops.conv_gemm(xb, wp, du, B=B, T=T, KS=1, dil=0, flags=L.CONV_FLAT_TILES, gelu_bwd_u=ub, stats=st)
    assert torch.equal(du, du_ref)
    assert bool(torch.isfinite(st).all()) and float(st[:, 1].abs().max()) == 0.0
    t = tol(dtype, B * T)
    np.testing.assert_allclose(st[:, 0].double().sum(0).cpu().numpy(), cs_ref.double().cpu().numpy(), rtol=1e-4, atol=t["atol"] * 1e-2 + 1e-4)
    # the same epilogue in the tile-per-workgroup kernel (statistics rows per (sample, 128-row tile))
    du2 = ops.new_rows(B, T, Cop, dtype, DEV)
    st2 = torch.full((ops.conv_stats_rows(B, T, 1, Cop, 0), 2, Cop), float("nan"), device=DEV)
    ops.conv_gemm(xb, wp, du2, B=B, T=T, KS=1, dil=0, gelu_bwd_u=ub, stats=st2)
    assert torch.equal(du2, du_ref)
    assert bool(torch.isfinite(st2).all()) and float(st2[:, 1].abs().max()) == 0.0
    np.testing.assert_allclose(st2[:, 0].double().sum(0).cpu().numpy(), cs_ref.double().cpu().numpy(), rtol=1e-4, atol=t["atol"] * 1e-2 + 1e-4)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("cin,cout,T,B", [(640, 1024, 360, 3), (64, 128, 50, 5), (128, 512, 129, 2)])
def test_conv1_flat_row_sumsq_epilogue(ops, dtype, cin, cout, T, B):
    """SDA_EPI_ROW_SUMSQ + sda_rows_sumsq_from_row_parts: per-sample ||y_b||^2 of the stored output without a pass over it
    (the brain-embedding norms of CLIPLoss, loss.py:65)."""
    from speech_decoding_amd import lib as L
    g = torch.Generator().manual_seed(13)
    Cip, Cop = L.pad_channels(cin), L.pad_channels(cout)
    xb = to_rows(ops, q(torch.randn(B, cin, T, generator=g), dtype), dtype)
    wp = ops.pack_conv_weight(q(torch.randn(cout, cin, 1, generator=g) / math.sqrt(cin), dtype).to(DEV), Cop, Cip, dtype)
    bias = ops.pack_vector(torch.randn(cout, generator=g).to(DEV), Cop)
    y0 = ops.conv_gemm(xb, wp, ops.new_rows(B, T, Cop, dtype, DEV), B=B, T=T, KS=1, dil=0, bias=bias, gelu=True)
    parts = torch.full((L.rows_alloc(B, T), Cop // 128), float("nan"), device=DEV)
    y = ops.conv_gemm(xb, wp, ops.new_rows(B, T, Cop, dtype, DEV), B=B, T=T, KS=1, dil=0, bias=bias, gelu=True,
                      flags=L.CONV_FLAT_TILES, row_sumsq=parts)
    assert torch.equal(y, y0)
    norms = ops.rows_sumsq_from_row_parts(parts, B, T)
    ref = from_rows(ops, y, B, cout, T).double().pow(2).sum(dim=(1, 2))
    np.testing.assert_allclose(norms.cpu().double().numpy(), ref.numpy(), rtol=2e-6)


# conv1_wide.hip (round 5, SDA_CONV_WIDE_TILES): the 1 x 1 convs on 256-row x 256 / 320-channel tiles, eight waves, persistent
# workgroups — against the tile-per-workgroup kernel (same products, same K order: bit-equal) and torch
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("cin,cout,T,B", [(320, 640, 360, 5), (640, 1024, 129, 3), (1024, 640, 200, 3), (640, 320, 77, 4),
                                          (64, 256, 50, 9), (96, 512, 1, 9), (320, 640, 360, 64)])
def test_conv1_wide_tiles_equal_tile_kernel(ops, dtype, cin, cout, T, B):
    from speech_decoding_amd import lib as L
    g = torch.Generator().manual_seed(21)
    Cip, Cop = L.pad_channels(cin), L.pad_channels(cout)
    x = q(torch.randn(B, cin, T, generator=g), dtype)
    w = q(torch.randn(cout, cin, 1, generator=g) / math.sqrt(cin), dtype)
    bias = ops.pack_vector(torch.randn(cout, generator=g).to(DEV), Cop)
    xb = to_rows(ops, x, dtype)
    wp = ops.pack_conv_weight(w.to(DEV), Cop, Cip, dtype)
    for kw in (dict(), dict(bias=bias, gelu=True, with_pre=True), dict(bias=bias)):
        outs = []
        for flags in (0, L.CONV_WIDE_TILES):
            kw2 = dict(kw)
            y, pre = ops.new_rows(B, T, Cop, dtype, DEV), ops.new_rows(B, T, Cop, dtype, DEV)
            if kw2.pop("with_pre", False):
                kw2["y_pre"] = pre
            ops.conv_gemm(xb, wp, y, B=B, T=T, KS=1, dil=0, flags=flags, **kw2)
            outs.append((y, pre))
        assert torch.equal(outs[0][0], outs[1][0])        # (pad rows and the slack behind the last sample included: never written)
        assert torch.equal(outs[0][1], outs[1][1])
    ref = TF.conv1d(x, w)
    np.testing.assert_allclose(from_rows(ops, ops.conv_gemm(xb, wp, ops.new_rows(B, T, Cop, dtype, DEV), B=B, T=T, KS=1, dil=0,
                                                            flags=L.CONV_WIDE_TILES), B, cout, T).numpy(), ref.numpy(), **tol(dtype, cin))
    # a width that is neither a multiple of 256 nor of 320 is refused, not silently served by another kernel
    if cout == 640:
        bad = ops.pack_conv_weight(q(torch.randn(128, cin, 1, generator=g), dtype).to(DEV), 128, Cip, dtype)
        with pytest.raises(L.SdaError):
            ops.conv_gemm(xb, bad, ops.new_rows(B, T, 128, dtype, DEV), B=B, T=T, KS=1, dil=0, flags=L.CONV_WIDE_TILES)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("cin,cout,T,B", [(1024, 640, 360, 4), (128, 256, 77, 3), (64, 320, 130, 6), (256, 512, 360, 20)])
def test_conv1_wide_gelu_backward_epilogue(ops, dtype, cin, cout, T, B):
    from speech_decoding_amd import lib as L
    g = torch.Generator().manual_seed(22)
    Cip, Cop = L.pad_channels(cin), L.pad_channels(cout)
    xb = to_rows(ops, q(torch.randn(B, cin, T, generator=g), dtype), dtype)
    wp = ops.pack_conv_weight(q(torch.randn(cout, cin, 1, generator=g) / math.sqrt(cin), dtype).to(DEV), Cop, Cip, dtype)
    ub = to_rows(ops, q(torch.randn(B, cout, T, generator=g), dtype), dtype)
    dg = ops.conv_gemm(xb, wp, ops.new_rows(B, T, Cop, dtype, DEV), B=B, T=T, KS=1, dil=0)
    du_ref = ops.new_rows(B, T, Cop, dtype, DEV)
    cs_ref = ops.gelu_backward_colsum(ub, dg, du_ref, B, T, ops.reduce_scratch(Cop, DEV))
    du = ops.new_rows(B, T, Cop, dtype, DEV)
    nrows = ops.conv_stats_rows(B, T, 1, Cop, L.CONV_WIDE_TILES | L.EPI_GELU_BWD)
    assert nrows == (B * L.rows_tp(T) + 255) // 256
    st = torch.full((nrows, 2, Cop), float("nan"), device=DEV)
    ops.conv_gemm(xb, wp, du, B=B, T=T, KS=1, dil=0, flags=L.CONV_WIDE_TILES, gelu_bwd_u=ub, stats=st)
    assert torch.equal(du, du_ref)
    assert bool(torch.isfinite(st).all()) and float(st[:, 1].abs().max()) == 0.0
    t = tol(dtype, B * T)
    np.testing.assert_allclose(st[:, 0].double().sum(0).cpu().numpy(), cs_ref.double().cpu().numpy(), rtol=1e-4, atol=t["atol"] * 1e-2 + 1e-4)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("cin,cout,T,B", [(640, 1024, 360, 3), (64, 256, 50, 5), (128, 512, 129, 2)])
def test_conv1_wide_row_sumsq_epilogue(ops, dtype, cin, cout, T, B):
    from speech_decoding_amd import lib as L
    g = torch.Generator().manual_seed(23)
    Cip, Cop = L.pad_channels(cin), L.pad_channels(cout)
    xb = to_rows(ops, q(torch.randn(B, cin, T, generator=g), dtype), dtype)
    wp = ops.pack_conv_weight(q(torch.randn(cout, cin, 1, generator=g) / math.sqrt(cin), dtype).to(DEV), Cop, Cip, dtype)
    bias = ops.pack_vector(torch.randn(cout, generator=g).to(DEV), Cop)
    y0, pre0 = ops.new_rows(B, T, Cop, dtype, DEV), ops.new_rows(B, T, Cop, dtype, DEV)
    ops.conv_gemm(xb, wp, y0, B=B, T=T, KS=1, dil=0, bias=bias, gelu=True, y_pre=pre0)
    parts = torch.full((L.rows_alloc(B, T), Cop // 128), float("nan"), device=DEV)
    y, pre = ops.new_rows(B, T, Cop, dtype, DEV), ops.new_rows(B, T, Cop, dtype, DEV)
    ops.conv_gemm(xb, wp, y, B=B, T=T, KS=1, dil=0, bias=bias, gelu=True, y_pre=pre, flags=L.CONV_WIDE_TILES, row_sumsq=parts)
    assert torch.equal(y, y0) and torch.equal(pre, pre0)
    norms = ops.rows_sumsq_from_row_parts(parts, B, T)
    ref = from_rows(ops, y, B, cout, T).double().pow(2).sum(dim=(1, 2))
    np.testing.assert_allclose(norms.cpu().double().numpy(), ref.numpy(), rtol=2e-6)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("cin,cout,dil,T,glu", [(40, 48, 2, 70, False), (320, 320, 4, 200, False),
                                                (64, 48, 2, 90, True), (320, 640, 2, 140, True)])
def test_conv3_dgrad_and_wgrad(ops, dtype, cin, cout, dil, T, glu):
    """dx via conv_gemm on mode-1 weights, dW via wgrad_gemm + reduce + unpack, vs torch autograd."""
    from speech_decoding_amd import lib as L
    g = torch.Generator().manual_seed(cin * 3 + cout + dil)
    B = 3
    x = q(torch.randn(B, cin, T, generator=g), dtype).requires_grad_(True)
    w = q(torch.randn(cout, cin, 3, generator=g) / math.sqrt(3 * cin), dtype).requires_grad_(True)
    dy = q(torch.randn(B, cout, T, generator=g), dtype)
    TF.conv1d(x, w, None, padding=dil, dilation=dil).backward(dy)
    half = cout // 2 if glu else 0
    half_p = L.pad_channels(half) if glu else 0
    Cin_p = L.pad_channels(cin)
    Cout_p = 2 * half_p if glu else L.pad_channels(cout)
    # dy in the packed channel order (GLU layout splits the two halves at half_p)
    dyb = ops.new_rows(B, T, Cout_p, dtype, DEV)
    if glu:
        tmp = torch.zeros(B, Cout_p, T)
        tmp[:, :half] = dy[:, :half]
        tmp[:, half_p: half_p + cout - half] = dy[:, half:]
        ops.pack_rows(tmp.to(DEV), dyb)
    else:
        ops.pack_rows(dy.to(DEV), dyb)
    xb = to_rows(ops, x.detach(), dtype)
    wt = ops.pack_conv_weight(w.detach().to(DEV), Cout_p, Cin_p, dtype, mode=1, glu_half=half, glu_half_p=half_p)
    dxb = ops.new_rows(B, T, Cin_p, dtype, DEV)
    ops.conv_gemm(dyb, wt, dxb, B=B, T=T, KS=3, dil=dil)
    np.testing.assert_allclose(from_rows(ops, dxb, B, cin, T).numpy(), x.grad.numpy(), **tol(dtype, 3 * cout))
    perm = torch.tensor([2, 0, 1], dtype=torch.int32, device=DEV)
    seg = torch.tensor([0, 1, 3], dtype=torch.int32, device=DEV)
    slabs = ops.wgrad_gemm(dyb, xb, B=B, T=T, KS=3, dil=dil, perm=perm, seg_start=seg, nseg=2)
    gw = ops.unpack_conv_wgrad(ops.reduce_slabs(slabs), 1, cout, cin, 3, Cout_p, Cin_p, half, half_p)[0].cpu()
    np.testing.assert_allclose(gw.numpy(), w.grad.numpy(), **tol(dtype, B * T))


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("tiling", [4096, 8192, 16384])
@pytest.mark.parametrize("cin,cout,dil,T", [(48, 40, 2, 70), (640, 320, 2, 300), (320, 320, 8, 360)])
def test_conv3_bn_backward_statistics_epilogue(ops, dtype, cin, cout, dil, T, tiling):
    """conv_gemm(bn_x=...) writes its output AND the BatchNorm+GELU backward sums of the layer that output is
    the gradient of; they must equal the stand-alone reduction pass over the stored output."""
    from speech_decoding_amd import lib as L
    g = torch.Generator().manual_seed(cin + cout + dil)
    B = 3
    Cin_p, Cout_p = L.pad_channels(cin), L.pad_channels(cout)
    dyb = to_rows(ops, q(torch.randn(B, cin, T, generator=g), dtype), dtype)
    w = q(torch.randn(cout, cin, 3, generator=g) / math.sqrt(3 * cin), dtype)
    wp = ops.pack_conv_weight(w.to(DEV), Cout_p, Cin_p, dtype)
    # the residual of every data-gradient conv of the model is the conv's own input (y = conv(x) + x): the flat-tile
    # kernel supports exactly that form; the tile-per-workgroup kernels take any residual tensor
    res = dyb if (tiling == 16384 and cin == cout) else (None if tiling == 16384 else to_rows(ops, q(torch.randn(B, cout, T, generator=g), dtype), dtype))
    h = to_rows(ops, q(torch.randn(B, cout, T, generator=g) * 1.3 + 0.2, dtype), dtype)      # the BN input
    gamma = (torch.rand(cout, generator=g) + 0.5).to(DEV)
    beta = (torch.rand(cout, generator=g) - 0.5).to(DEV)
    hs = ops.unpack_rows(h, B, cout, T)
    part = torch.zeros((1, 2, Cout_p), device=DEV)
    part[0, 0, :cout] = hs.sum(dim=(0, 2))
    part[0, 1, :cout] = (hs ** 2).sum(dim=(0, 2))
    mean, rstd, _, _, coef = ops.bn_finalize(part, 1, B * T, gamma, beta, torch.zeros(cout, device=DEV),
                                             torch.ones(cout, device=DEV), Cout_p, True, want_bwd_coef=True)
    np.testing.assert_array_equal(coef[2].cpu().numpy(), mean.cpu().numpy())
    np.testing.assert_array_equal(coef[0, :cout].cpu().numpy(), gamma.cpu().numpy())
    assert float(coef[:, cout:].abs().max()) == 0.0 if Cout_p > cout else True
    out = ops.new_rows(B, T, Cout_p, dtype, DEV)
    st = torch.full((ops.conv_stats_rows(B, T, 3, Cout_p, tiling), 2, Cout_p), float("nan"), device=DEV)   # every row must be written
    ops.conv_gemm(dyb, wp, out, B=B, T=T, KS=3, dil=dil, res=res, stats=st, bn_x=h, bn_coef=coef, flags=tiling)
    plain = ops.new_rows(B, T, Cout_p, dtype, DEV)
    ops.conv_gemm(dyb, wp, plain, B=B, T=T, KS=3, dil=dil, res=res, flags=tiling)
    assert torch.equal(out, plain)                                   # the output itself is unchanged by the mode
    dx1, dx2 = ops.new_rows(B, T, Cout_p, dtype, DEV), ops.new_rows(B, T, Cout_p, dtype, DEV)
    dg_ref, db_ref = ops.bn_gelu_backward(out, h, mean, rstd, gamma, beta, dx1, B, T, ops.reduce_scratch(Cout_p, DEV))
    dg, db = ops.bn_gelu_backward(out, h, mean, rstd, gamma, beta, dx2, B, T, ops.reduce_scratch(Cout_p, DEV), tile_stats=st)
    scale = float(dg_ref.abs().max()) + 1e-6
    # same fp32 terms, different summation order (per tile vs per row block): 1e-5 of the largest sum
    assert float((dg - dg_ref).abs().max()) <= 1e-5 * scale * (1 if dtype == torch.float32 else 4)
    assert float((db - db_ref).abs().max()) <= 1e-5 * (float(db_ref.abs().max()) + 1e-6) * (1 if dtype == torch.float32 else 4)
    np.testing.assert_allclose(from_rows(ops, dx2, B, cout, T).numpy(), from_rows(ops, dx1, B, cout, T).numpy(),
                               rtol=1e-3, atol=1e-4 if dtype == torch.float32 else 2e-2)
    # SDA_EPI_BN_STORE_DG: the conv stores dg = dy * GELU'(u) (rounded) instead of dy, statistics of dg as stored; the
    # BatchNorm backward is finished without a second GELU' (fp32: the same arithmetic; 16-bit: one more rounding of dg)
    out_dg = ops.new_rows(B, T, Cout_p, dtype, DEV)
    st_dg = torch.full_like(st, float("nan"))
    ops.conv_gemm(dyb, wp, out_dg, B=B, T=T, KS=3, dil=dil, res=res, stats=st_dg, bn_x=h, bn_coef=coef, flags=tiling | L.EPI_BN_STORE_DG)
    u = ops.unpack_rows(h, B, cout, T)
    u = (u - mean[:cout, None]) * rstd[:cout, None] * gamma[:, None] + beta[:, None]
    uu = u.double()
    gprime = 0.5 * (1 + torch.erf(uu / math.sqrt(2))) + uu * torch.exp(-0.5 * uu * uu) / math.sqrt(2 * math.pi)
    dg_expect = (ops.unpack_rows(out, B, cout, T).double() * gprime).float().cpu()
    np.testing.assert_allclose(from_rows(ops, out_dg, B, cout, T).numpy(), dg_expect.numpy(), **tol(dtype))
    dx3 = ops.new_rows(B, T, Cout_p, dtype, DEV)
    dg3, db3 = ops.bn_gelu_backward(out_dg, h, mean, rstd, gamma, beta, dx3, B, T, ops.reduce_scratch(Cout_p, DEV), tile_stats=st_dg,
                                    dy_is_dg=True)
    loose = 1e-5 if dtype == torch.float32 else (2e-3 if dtype == torch.float16 else 1e-2)
    assert float((dg3 - dg_ref).abs().max()) <= loose * scale * 4
    assert float((db3 - db_ref).abs().max()) <= loose * (float(db_ref.abs().max()) + 1e-6) * 4
    # (16-bit: the extra rounding moves some outputs by one unit in the last place, 2^-7 / 2^-10 relative)
    np.testing.assert_allclose(from_rows(ops, dx3, B, cout, T).numpy(), from_rows(ops, dx1, B, cout, T).numpy(),
                               rtol={torch.float32: 1e-3, torch.float16: 2e-3, torch.bfloat16: 1e-2}[dtype],
                               atol=1e-4 if dtype == torch.float32 else 2e-2)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("cin,cout,KS,dil,T,B,nseg", [(320, 320, 3, 16, 360, 7, 3), (64, 160, 3, 1, 50, 5, 5), (320, 640, 3, 2, 200, 9, 2),
                                                      (640, 1024, 1, 0, 131, 4, 1), (48, 40, 3, 8, 64, 3, 1)])
def test_wgrad_flat_rows(ops, dtype, cin, cout, KS, dil, T, B, nseg):
    """SDA_WGRAD_FLAT_ROWS: a segment of consecutive samples contracted as ONE run of rows (pad rows of dy are zero) in whole
    K-chunks — same weight gradient as torch autograd (models.py:128-150 through autograd), and as the per-sample form."""
    from speech_decoding_amd import lib as L
    g = torch.Generator().manual_seed(cin + 7 * cout + dil + T)
    x = q(torch.randn(B, cin, T, generator=g), dtype).requires_grad_(True)
    w = q(torch.randn(cout, cin, KS, generator=g) / math.sqrt(KS * cin), dtype).requires_grad_(True)
    dy = q(torch.randn(B, cout, T, generator=g), dtype)
    TF.conv1d(x, w, None, padding=dil if KS == 3 else 0, dilation=max(dil, 1)).backward(dy)
    Cin_p, Cout_p = L.pad_channels(cin), L.pad_channels(cout)
    dyb, xb = to_rows(ops, dy, dtype), to_rows(ops, x.detach(), dtype)      # (to_rows fills valid rows of a zeroed buffer)
    edges = np.floor(np.linspace(0, B, nseg + 1)).astype(np.int32)
    seg = torch.from_numpy(edges).to(DEV)
    flat = ops.wgrad_gemm(dyb, xb, B=B, T=T, KS=KS, dil=dil, seg_start=seg, nseg=nseg, flat_rows=True)
    per_sample = ops.wgrad_gemm(dyb, xb, B=B, T=T, KS=KS, dil=dil, seg_start=seg, nseg=nseg)
    gw = ops.unpack_conv_wgrad(ops.reduce_slabs(flat), 1, cout, cin, KS, Cout_p, Cin_p)[0].cpu()
    # operands are exact in every storage type (q), products accumulate in fp32: accumulation noise only, relative to the
    # size of the sums (up to 1e2 here)
    assert float((gw - w.grad).abs().max()) <= 1e-5 * float(w.grad.abs().max())
    # the same products in a different summation order (chunks cut differently)
    scale = float(per_sample.abs().max()) + 1e-12
    assert float((flat - per_sample).abs().max()) <= 2e-5 * scale * math.sqrt(B * T / 64)


@pytest.mark.parametrize("dtype", DTYPES)
def test_wgrad_per_subject_segments(ops, dtype):
    from speech_decoding_amd import lib as L
    g = torch.Generator().manual_seed(8)
    B, C, T, S = 6, 80, 100, 4
    x = q(torch.randn(B, C, T, generator=g), dtype)
    dy = q(torch.randn(B, C, T, generator=g), dtype)
    sidx = np.array([1, 3, 1, 0, 3, 3])
    ref = torch.zeros(S, C, C)
    for b in range(B):
        ref[sidx[b]] += dy[b] @ x[b].T
    order = np.argsort(sidx, kind="stable").astype(np.int32)
    seg = np.searchsorted(sidx[order], np.arange(S + 1)).astype(np.int32)
    Cp = L.pad_channels(C)
    slabs = ops.wgrad_gemm(to_rows(ops, dy, dtype), to_rows(ops, x, dtype), B=B, T=T, KS=1, dil=0,
                           perm=torch.from_numpy(order).to(DEV), seg_start=torch.from_numpy(seg).to(DEV), nseg=S)
    got = ops.unpack_conv_wgrad(slabs, S, C, C, 1, Cp, Cp)[..., 0].cpu()
    np.testing.assert_allclose(got.numpy(), ref.numpy(), **tol(dtype, 3 * T))
    assert float(got[2].abs().max()) == 0.0          # subject absent from the batch


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("T,KS,dil", [(360, 3, 1), (360, 3, 16), (100, 3, 4), (104, 1, 0), (130, 3, 2)])
def test_wgrad_per_subject_segments_on_padded_samples(ops, dtype, T, KS, dil):
    """SDA_WGRAD_FLAT_ROWS with a sample permutation (round 5): every sample contracted as whole K-chunks across the zero rows
    around it (T = 360: 16 + 360 + 8; T = 130: needs 62 extra rows, more than the padding holds -> the plain per-sample form).
    Equal to the plain form up to the fp32 accumulation order (the rows fall into different K-steps): the extra rows are exact
    zeros in dy."""
    from speech_decoding_amd import lib as L
    g = torch.Generator().manual_seed(T + dil)
    B, Cin, Cout, S = 7, 70, 160, 3
    x = q(torch.randn(B, Cin, T, generator=g), dtype)
    dy = q(torch.randn(B, Cout, T, generator=g), dtype)
    sidx = np.array([2, 0, 2, 2, 0, 2, 0])
    order = np.argsort(sidx, kind="stable").astype(np.int32)
    seg = np.searchsorted(sidx[order], np.arange(S + 1)).astype(np.int32)
    dyt, xt = to_rows(ops, dy, dtype), to_rows(ops, x, dtype)
    kw = dict(B=B, T=T, KS=KS, dil=dil, perm=torch.from_numpy(order).to(DEV), seg_start=torch.from_numpy(seg).to(DEV), nseg=S)
    plain = ops.wgrad_gemm(dyt, xt, **kw)
    padded = ops.wgrad_gemm(dyt, xt, flat_rows=True, **kw)
    assert float((plain - padded).abs().max()) <= 2e-5 * float(plain.abs().max())
    ref = torch.zeros(S, KS, Cout, Cin)
    xp = TF.pad(x, (dil, dil))
    for b in range(B):
        for tap in range(KS):
            off = tap * dil if KS == 3 else 0
            ref[sidx[b], tap] += dy[b] @ (xp[b][:, off: off + T] if KS == 3 else x[b]).T
    got = padded[:, :, :Cout, :Cin].cpu()
    t_ = tol(dtype, 3 * T) if dtype != torch.float32 else dict(rtol=1e-4, atol=2e-4)      # (fp32: ~2 500-term sums of O(1) products)
    np.testing.assert_allclose(got.numpy(), ref.numpy(), **t_)
    assert float(got[1].abs().max()) == 0.0          # subject absent from the batch


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("C,T", [(24, 40), (320, 200)])
def test_batchnorm_gelu_forward_backward(ops, dtype, C, T):
    from speech_decoding_amd import lib as L
    g = torch.Generator().manual_seed(C)
    B = 4
    x = q(torch.randn(B, C, T, generator=g) * 1.5 + 0.3, dtype).requires_grad_(True)
    gamma = (torch.rand(C, generator=g) + 0.5).requires_grad_(True)
    beta = (torch.rand(C, generator=g) - 0.5).requires_grad_(True)
    rm, rv = torch.zeros(C), torch.ones(C)
    rm0, rv0 = rm.clone(), rv.clone()                      # torch updates rm/rv in place below
    dy = q(torch.randn(B, C, T, generator=g), dtype)
    y = TF.gelu(TF.batch_norm(x, rm, rv, gamma, beta, training=True, momentum=0.1, eps=1e-5))
    y.backward(dy)
    Cp = L.pad_channels(C)
    xb = to_rows(ops, x.detach(), dtype)
    # statistics through a 1x1 identity conv epilogue would be circular; build the partials directly
    xs = ops.unpack_rows(xb, B, C, T)                      # stored (rounded) values, on device
    part = torch.zeros((1, 2, Cp), device=DEV)
    part[0, 0, :C] = xs.sum(dim=(0, 2))
    part[0, 1, :C] = (xs ** 2).sum(dim=(0, 2))
    rm_d, rv_d = rm0.to(DEV), rv0.to(DEV)
    mean, rstd, scale, shift = ops.bn_finalize(part, 1, B * T, gamma.detach().to(DEV), beta.detach().to(DEV), rm_d, rv_d,
                                               Cp, True)
    np.testing.assert_allclose(rm_d.cpu().numpy(), rm.numpy(), rtol=1e-4, atol=1e-5)     # torch updated rm/rv in place
    np.testing.assert_allclose(rv_d.cpu().numpy(), rv.numpy(), rtol=1e-4, atol=1e-5)
    yb = ops.bn_gelu_forward(xb, ops.new_rows(B, T, Cp, dtype, DEV), scale, shift, B, T)
    np.testing.assert_allclose(from_rows(ops, yb, B, C, T).numpy(), y.detach().numpy(), **tol(dtype))
    dxb = ops.new_rows(B, T, Cp, dtype, DEV)
    dgam, dbet = ops.bn_gelu_backward(to_rows(ops, dy, dtype), xb, mean, rstd, gamma.detach().to(DEV), beta.detach().to(DEV),
                                      dxb, B, T, ops.reduce_scratch(Cp, DEV))
    t = tol(dtype, B * T)
    np.testing.assert_allclose(dgam[:C].cpu().numpy(), gamma.grad.numpy(), rtol=t["rtol"], atol=t["atol"])
    np.testing.assert_allclose(dbet[:C].cpu().numpy(), beta.grad.numpy(), rtol=t["rtol"], atol=t["atol"])
    np.testing.assert_allclose(from_rows(ops, dxb, B, C, T).numpy(), x.grad.numpy(), **tol(dtype, 4))
    # eval mode: scale/shift from the running statistics
    _, _, sc_e, sh_e = ops.bn_finalize(None, 0, B * T, gamma.detach().to(DEV), beta.detach().to(DEV), rm_d, rv_d, Cp, False)
    ye = TF.gelu(TF.batch_norm(x.detach(), rm, rv, gamma.detach(), beta.detach(), training=False, eps=1e-5))
    yb = ops.bn_gelu_forward(xb, ops.new_rows(B, T, Cp, dtype, DEV), sc_e, sh_e, B, T)
    np.testing.assert_allclose(from_rows(ops, yb, B, C, T).numpy(), ye.numpy(), **tol(dtype))


@pytest.mark.parametrize("dtype", DTYPES)
def test_glu_gelu_colsum(ops, dtype):
    from speech_decoding_amd import lib as L
    g = torch.Generator().manual_seed(5)
    B, Ch, T = 3, 24, 77
    Chp = L.pad_channels(Ch)
    a = q(torch.randn(B, Ch, T, generator=g), dtype).requires_grad_(True)
    gate = q(torch.randn(B, Ch, T, generator=g), dtype).requires_grad_(True)
    dy = q(torch.randn(B, Ch, T, generator=g), dtype)
    y = a * torch.sigmoid(gate)
    y.backward(dy)
    packed = torch.zeros(B, 2 * Chp, T)
    packed[:, :Ch], packed[:, Chp: Chp + Ch] = a.detach(), gate.detach()
    xb = to_rows(ops, packed, dtype, Cp=2 * Chp)
    yb = ops.glu_forward(xb, ops.new_rows(B, T, Chp, dtype, DEV), B, T)
    np.testing.assert_allclose(from_rows(ops, yb, B, Ch, T).numpy(), y.detach().numpy(), **tol(dtype))
    dxb = ops.glu_backward(xb, to_rows(ops, dy, dtype), ops.new_rows(B, T, 2 * Chp, dtype, DEV), B, T)
    dx = from_rows(ops, dxb, B, 2 * Chp, T)
    np.testing.assert_allclose(dx[:, :Ch].numpy(), a.grad.numpy(), **tol(dtype))
    np.testing.assert_allclose(dx[:, Chp: Chp + Ch].numpy(), gate.grad.numpy(), **tol(dtype))
    # GELU backward + column sums
    u = q(torch.randn(B, Ch, T, generator=g), dtype).requires_grad_(True)
    TF.gelu(u).backward(dy)
    du = ops.gelu_backward(to_rows(ops, u.detach(), dtype), to_rows(ops, dy, dtype), ops.new_rows(B, T, Chp, dtype, DEV), B, T)
    np.testing.assert_allclose(from_rows(ops, du, B, Ch, T).numpy(), u.grad.numpy(), **tol(dtype))
    cs = ops.colsum(to_rows(ops, dy, dtype), B, T, ops.reduce_scratch(Chp, DEV))
    np.testing.assert_allclose(cs[:Ch].cpu().numpy(), dy.sum(dim=(0, 2)).numpy(), rtol=1e-4, atol=1e-3)
    # fused variants: same outputs + column sums of the outputs (bias gradients)
    dxb2 = ops.new_rows(B, T, 2 * Chp, dtype, DEV)
    cs2 = ops.glu_backward_colsum(xb, to_rows(ops, dy, dtype), dxb2, B, T, ops.reduce_scratch(2 * Chp, DEV))
    assert torch.equal(dxb2, dxb)
    dx_stored = from_rows(ops, dxb2, B, 2 * Chp, T)
    t = tol(dtype, B * T)
    np.testing.assert_allclose(cs2.cpu().numpy(), dx_stored.sum(dim=(0, 2)).numpy(), rtol=t["rtol"], atol=t["atol"])
    du2 = ops.new_rows(B, T, Chp, dtype, DEV)
    cs3 = ops.gelu_backward_colsum(to_rows(ops, u.detach(), dtype), to_rows(ops, dy, dtype), du2, B, T, ops.reduce_scratch(Chp, DEV))
    assert torch.equal(du2, du)
    np.testing.assert_allclose(cs3[:Ch].cpu().numpy(), u.grad.sum(dim=(0, 2)).numpy(), rtol=t["rtol"], atol=t["atol"])


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("D1,K,C,gemm", [(20, 6, 70, False), (40, 8, 70, True), (270, 32, 60, True)])
def test_spatial_attention_weights_forward_backward(ops, dtype, D1, K, C, gemm):
    """gemm: the two contractions on the matrix-core path (needs 2*K*K % 64 == 0) vs the stand-alone kernels."""
    from speech_decoding_amd import lib as L
    g = torch.Generator().manual_seed(2)
    loc = O.synthetic_positions(C, seed=1)
    cos, sin = O.fourier_tables(loc, K)
    z = torch.complex(torch.rand(D1, K * K, generator=g), torch.rand(D1, K * K, generator=g)).requires_grad_(True)
    mask = O.dropout_mask(loc, 5, 0.15)
    P = {"subject_block.spatial_attention.z": z, "subject_block.spatial_attention.cos": cos,
         "subject_block.spatial_attention.sin": sin}
    W = O.sa_weights(P)
    dWd = torch.randn(D1, C, generator=g)
    (W * mask[None, :] * dWd).sum().backward()
    D1p, Cp = L.pad_channels(D1), L.pad_channels(C)
    tab_f, tab_b = ops.sa_gemm_tables(cos.to(DEV), sin.to(DEV)) if gemm else (None, None)
    Wg, Wp = ops.sa_weights_forward(z.detach().to(DEV), cos.to(DEV), sin.to(DEV), mask.to(DEV), D1p, Cp, dtype, fwd_table=tab_f)
    np.testing.assert_allclose(Wg.cpu().numpy(), W.detach().numpy(), rtol=2e-4, atol=1e-7)
    wp = Wp.float().cpu()[0, 0]
    np.testing.assert_allclose(wp[:D1, :C].numpy(), q(W.detach() * mask[None, :], dtype).numpy(), rtol=1e-2 if dtype != torch.float32 else 2e-4, atol=1e-6)
    assert float(wp[D1:].abs().max()) == 0.0 and float(wp[:, C:].abs().max()) == 0.0
    dpad = torch.zeros(D1p, Cp)
    dpad[:D1, :C] = dWd
    dz = ops.sa_weights_backward(dpad.to(DEV), Wg, mask.to(DEV), cos.t().contiguous().to(DEV), sin.t().contiguous().to(DEV), K * K,
                                 bwd_table=tab_b)
    ref = z.grad
    np.testing.assert_allclose(torch.view_as_real(dz.cpu()).numpy(), torch.view_as_real(ref).numpy(), rtol=2e-3,
                               atol=2e-6 * max(1.0, float(ref.abs().max())))


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("B,F,T", [(6, 32, 40), (24, 64, 100), (130, 64, 20)])
def test_clip_loss_forward_backward_and_ranks(ops, dtype, B, F, T):
    from speech_decoding_amd import engine as E, lib as L
    g = torch.Generator().manual_seed(B)
    Y = q(torch.randn(B, F, T, generator=g), dtype)
    Z = q(0.3 * Y + torch.randn(B, F, T, generator=g), dtype).requires_grad_(True)
    temp = torch.tensor([2.0], requires_grad=True)
    loss, logits = O.clip_loss(Y, Z, temp)
    loss.backward()
    Yt, Zt = to_rows(ops, Y, dtype), to_rows(ops, Z.detach(), dtype)
    lg, lgts, cnt, ctx = E.clip_forward(Yt, Zt, temp.detach().to(DEV), Bm=B, Bn=B, T=T)
    lt = dict(rtol=1e-4, atol=1e-4) if dtype == torch.float32 else dict(rtol=3e-2, atol=3e-2)
    np.testing.assert_allclose(lgts.cpu().numpy(), logits.detach().numpy(), **lt)
    assert abs(float(lg) - float(loss)) < (1e-4 if dtype == torch.float32 else 3e-2)
    assert abs(float(ctx.dtemp) - float(temp.grad)) < (1e-4 if dtype == torch.float32 else 3e-2) * max(1.0, abs(float(temp.grad)))
    dZt = ops.new_rows(B, T, L.pad_channels(F), dtype, DEV)
    E.clip_backward(ctx, dZt)
    dz = from_rows(ops, dZt, B, F, T)
    scale = float(Z.grad.abs().max())
    assert float((dz - Z.grad).abs().max()) <= (2e-4 if dtype == torch.float32 else 3e-2) * scale
    # retrieval ranks against a direct count on the oracle logits
    sim = logits.detach()
    want = (sim > sim.diag()[:, None]).sum(dim=1)
    if dtype == torch.float32:
        assert torch.equal(cnt.cpu().long(), want)
    else:
        assert (cnt.cpu().long() - want).abs().max() <= 1


def test_fused_adam_matches_torch_adam(ops):
    """speech_decoding_amd.optim.FusedAdam vs torch.optim.Adam (train.py:161-163) over mixed shapes, a complex
    parameter, an odd-sized tensor and a parameter without gradient; 5 steps."""
    from speech_decoding_amd.optim import FusedAdam
    g = torch.Generator().manual_seed(0)
    shapes = [(320, 270, 3), (33,), (1,), (7, 5)]
    mk = lambda: [torch.nn.Parameter(torch.randn(s, generator=torch.Generator().manual_seed(i)).to(DEV)) for i, s in enumerate(shapes)] + \
                 [torch.nn.Parameter(torch.randn(6, 9, dtype=torch.cfloat, generator=torch.Generator().manual_seed(9)).to(DEV)),
                  torch.nn.Parameter(torch.zeros(4, device=DEV))]
    pa, pb = mk(), mk()
    oa, ob = torch.optim.Adam(pa, lr=3e-4), FusedAdam(pb, lr=3e-4)
    for step in range(5):
        for a, b in zip(pa[:-1], pb[:-1]):                   # the last parameter never receives a gradient
            gr = torch.randn(a.shape, dtype=a.dtype, generator=g).to(DEV) * (10.0 ** (step - 2))
            a.grad, b.grad = gr.clone(), gr.clone()
        oa.step(); ob.step()
    for a, b in zip(pa, pb):
        ra, rb = (torch.view_as_real(a), torch.view_as_real(b)) if a.is_complex() else (a, b)
        assert float((ra - rb).abs().max()) < 2e-7 + 1e-6 * float(ra.abs().max())


# ---------------------------------------------------------------------------------------------------------------
# parameter-space products (csrc/param_gemm.hip): the composed SubjectBlock's small fp32 matrix products on strided views
# ---------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("case", ["plain", "a_transposed", "b_transposed", "batched_a", "batched_b", "batched_both_sliced",
                                  "tiny", "wide_tiles", "typed_out_into_view"])
def test_param_gemm_on_strided_views_against_fp64(ops, case):
    g = torch.Generator().manual_seed(len(case))

    def rnd(*shape):
        return torch.randn(*shape, generator=g).to(DEV)

    out = None
    if case == "plain":
        A, B = rnd(270, 270), rnd(270, 208)
    elif case == "a_transposed":                     # K runs along A's slow index
        A, B = rnd(270, 270).t(), rnd(270, 208)
    elif case == "b_transposed":
        A, B = rnd(270, 208), rnd(270, 208).t()
    elif case == "batched_a":                        # 3-d A, shared 2-d B (the composition W_subj[s] . T1aug)
        A, B = rnd(27, 270, 270), rnd(270, 209)
    elif case == "batched_b":                        # shared 2-d A, 3-d B
        A, B = rnd(270, 960), rnd(5, 960, 256)[:, :, :209]
    elif case == "batched_both_sliced":              # sliced, transposed batch operands (chain rule: W_subj[s]^T . G[s])
        A, B = rnd(7, 300, 270)[:, :270].transpose(1, 2), rnd(7, 320, 256)[:, :270, :209]
    elif case == "tiny":
        A, B = rnd(3, 5), rnd(5, 2)
    elif case == "wide_tiles":                       # enough tiles for the 128 x 128 variant
        A, B = rnd(12, 700, 77), rnd(12, 77, 650)
    else:                                            # bf16 result written into a slice of a padded, zero-initialised buffer
        A, B = rnd(4, 270, 270), rnd(270, 209)
        buf = torch.zeros((4, 1, 320, 256), dtype=torch.bfloat16, device=DEV)
        out = buf[:, 0, :270, :209]
    got = ops.param_gemm(A, B, out=out)
    ref = torch.matmul(A.double().cpu(), B.double().cpu())
    scale = float(ref.abs().max())
    if out is None:
        assert got.dtype == torch.float32 and got.is_contiguous()
        assert float((got.double().cpu() - ref).abs().max()) <= 2e-6 * scale * math.sqrt(A.shape[-1])
    else:
        assert got.data_ptr() == out.data_ptr()
        assert float((got.double().cpu() - ref).abs().max()) <= 2 ** -8 * scale
        mask = torch.ones_like(buf, dtype=torch.bool)
        mask[:, 0, :270, :209] = False
        assert float(buf[mask].float().abs().max()) == 0.0          # nothing outside the view is touched
    again = ops.param_gemm(A, B, out=None if out is None else torch.zeros_like(buf)[:, 0, :270, :209])
    assert torch.equal(again.float(), got.float())                 # fixed summation order: bitwise reproducible


def test_copy3d_strided(ops):
    g = torch.Generator().manual_seed(0)
    w = torch.randn(320, 270, 3, generator=g).to(DEV)                              # a conv weight [o][d][tap]
    dst = torch.zeros((270, 3, 384), dtype=torch.float32, device=DEV)
    ops.copy3d(dst[:, :, :320], w.permute(1, 2, 0))
    assert torch.equal(dst[:, :, :320], w.permute(1, 2, 0)) and float(dst[:, :, 320:].abs().max()) == 0.0
    col = torch.randn(270, generator=g).to(DEV)
    T1 = torch.zeros((270, 209), device=DEV)
    ops.copy3d(T1[:, 208:], col[:, None])
    assert torch.equal(T1[:, 208], col) and float(T1[:, :208].abs().max()) == 0.0
    out = ops.copy3d(torch.empty(270, device=DEV), T1[:, 208])
    assert torch.equal(out, col) and out.is_contiguous()


# ---------------------------------------------------------------------------------------------------------------
# the loss's embedding gradient as a streaming kernel (csrc/loss_gemm.hip)
# ---------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("Bm,Bn,F,T,dtype", [(256, 256, 1024, 360, torch.bfloat16),     # config 2: the benchmarked shape
                                            (12, 12, 64, 40, torch.bfloat16),           # far below one tile of anything
                                            (100, 300, 128, 50, torch.float16),          # ragged batch, two column blocks of 256
                                            (256, 70, 64, 17, torch.float16),
                                            (33, 64, 192, 9, torch.bfloat16),
                                            # more than 256 speech rows (a rank's block under data parallelism): the tiled form —
                                            # 256 x 256 tiles, 32-row K-steps, both operands read transposed (round 5)
                                            (512, 256, 64, 44, torch.bfloat16),          # 15 column tiles, one block of 256 columns
                                            (288, 70, 128, 48, torch.float16),           # ragged columns: the tile is wider than G's pitch
                                            (2048, 300, 64, 12, torch.bfloat16),         # two column blocks, 64 K-steps
                                            (320, 256, 256, 20, torch.float16)])
def test_clip_dz_streaming_kernel_against_fp64(ops, Bm, Bn, F, T, dtype):
    """dZ[j] = dloss * (cscale[j] * sum_i G[i][j] Y[i] - rscale[j] Z[j]) on row-layout embeddings: against fp64 on the same
    rounded operands, and against the general kernel it replaces on one GPU (wgrad_gemm's typed output)."""
    from speech_decoding_amd import lib as L
    g = torch.Generator().manual_seed(Bm + Bn)
    Y = torch.randn(Bm, F, T, generator=g)
    Z = torch.randn(Bn, F, T, generator=g)
    Yt, Zt = to_rows(ops, Y, dtype), to_rows(ops, Z, dtype)
    re = L.rows_tp(T) * Yt.shape[1]
    Gm = torch.zeros((Bm + 1, L.pad_channels(Bn)))
    Gm[:Bm, :Bn] = torch.randn(Bm, Bn, generator=g) * 0.3
    Gd = Gm.to(dtype).to(DEV)
    cs, rs = (torch.rand(Bn, generator=g) + 0.5).to(DEV), (torch.randn(Bn, generator=g) * 0.1).to(DEV)
    dloss = torch.tensor([1.7], device=DEV)
    assert L.load().sda_clip_dz_supported(Bm, Bn, re, ops.dt_code(dtype))
    out = ops.new_rows_uninit(Bn, T, Yt.shape[1], dtype, DEV)
    ops.clip_dz(Gd, Yt, Zt, out, rs, cs, Bm=Bm, Bn=Bn, row_elems=re, out_scale=dloss)
    old = ops.new_rows_uninit(Bn, T, Yt.shape[1], dtype, DEV)
    ops.matmul_tn_typed(Gd, Yt, old, Zt, rs, M_rows=Bm, N_valid=Bn, K_cols=re, pitch=re, out_scale=dloss, acc_scale=cs)
    Yq, Zq, Gq = q(Y, dtype).double().reshape(Bm, -1), q(Z, dtype).double().reshape(Bn, -1), Gd[:Bm, :Bn].double().cpu()
    ref = 1.7 * (cs.double().cpu()[:, None] * (Gq.t() @ Yq) - rs.double().cpu()[:, None] * Zq)
    got = ops.rows_view(out, Bn, F, T).double().cpu().reshape(Bn, -1)
    was = ops.rows_view(old, Bn, F, T).double().cpu().reshape(Bn, -1)
    eps = 2 ** -8 if dtype == torch.bfloat16 else 2 ** -11               # one rounding of the stored result
    assert float((got - ref).abs().max()) <= 1.01 * eps * float(ref.abs().max())
    assert float((got - was).abs().max()) <= 1.01 * eps * float(ref.abs().max())
    full = out.float().cpu()
    Tp = L.rows_tp(T)
    for b in range(Bn):                                                   # pad rows come out as exact zeros, pad channels too
        assert float(full[b * Tp: b * Tp + L.ROW_PAD].abs().max()) == 0.0
    assert float(full[:, F:].abs().max()) == 0.0 if full.shape[1] > F else True


# -----------------------------------------------------------------------------------------------
# the similarity matmul on 256 x 256 tiles (csrc/sim_gemm.hip; loss.py:68): against fp64 on the SAME 16-bit operands (the only
# rounding left is the fp32 accumulation order), ragged row counts on both sides (rows past the operands are clamped), a
# row pitch larger than the contraction, and bit-equal repeats (the K slices are summed in fixed order)
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("M,N,K,pitch", [(256, 256, 16384, 16384), (300, 200, 4096, 4096 + 64), (64, 24, 2048, 2048), (2048, 256, 8192, 8192),
                                         (513, 257, 1024, 1024)])
def test_sim_gemm_256_tiles_against_fp64(ops, dtype, M, N, K, pitch):
    from speech_decoding_amd import lib as L
    g = torch.Generator(device=DEV).manual_seed(M * 7 + N)
    X = torch.randn(M, pitch, device=DEV, generator=g).to(dtype)
    W = torch.randn(N, pitch, device=DEV, generator=g).to(dtype)
    assert ops.SIM_GEMM_TILES256 and L.load().sda_sim_gemm_ksplit(M, N, K, ops.dt_code(dtype)) > 0
    S = ops.matmul_nt_splitk(X, W, M, N, K, pitch)
    assert S.shape == (M, L.pad_channels(N)) and S.dtype == torch.float32
    ref = X[:, :K].double() @ W[:, :K].double().t()
    err = (S[:, :N].double() - ref).abs().max().item()
    assert err <= 2e-5 * math.sqrt(K) * 4, err            # fp32 accumulation of K products of O(1) terms
    again = ops.matmul_nt_splitk(X, W, M, N, K, pitch)
    assert torch.equal(S[:, :N], again[:, :N])
    # the fp32 storage type is not served by this kernel: the split-K matrix mode of conv_gemm takes it
    assert L.load().sda_sim_gemm_ksplit(M, N, K, ops.dt_code(torch.float32)) == 0
    # and equals conv_gemm's split-K matrix mode within the accumulation order
    ops.SIM_GEMM_TILES256 = False
    try:
        old = ops.matmul_nt_splitk(X, W, M, N, K, pitch)
    finally:
        ops.SIM_GEMM_TILES256 = True
    assert (old[:, :N] - S[:, :N]).abs().max().item() <= 2e-5 * math.sqrt(K) * 4


def test_priority_stream_entry_point(ops):
    """sda_stream_create_priority: the three HIP priorities give usable streams (a kernel runs on each), anything else is an error."""
    from speech_decoding_amd import lib as L
    for prio in (-1, 0, 1):
        st = torch.cuda.ExternalStream(ops.stream_create_priority(prio), device=DEV)
        with torch.cuda.stream(st):
            t = ops.upload_small(np.arange(7, dtype=np.int32), DEV)
        st.synchronize()
        assert t.cpu().tolist() == list(range(7))
    with pytest.raises(L.SdaError):
        ops.stream_create_priority(5)


# ------------------------------------------------------------------------------------------------------ ABI 4 (round 5)
def test_fill_zero(ops):
    for shape in [(10, 320), (3,), (1, 7), (4097,)]:
        t = ops.zeros(shape, torch.float32, DEV)
        assert tuple(t.shape) == shape and float(t.abs().max()) == 0.0
    big = torch.full((5000,), 3.0, device=DEV)
    from speech_decoding_amd import lib as L
    L.check(L.load().sda_fill_zero(big.data_ptr() + 16 * 4, 4 * 1001, None if False else torch.cuda.current_stream().cuda_stream), "fill_zero")
    ref = torch.full((5000,), 3.0)
    ref[16:16 + 1001] = 0
    assert torch.equal(big.cpu(), ref)                 # nothing before the start, nothing behind the last byte


@pytest.mark.parametrize("dtype", DTYPES)
def test_gather_samples_is_index_select_plus_pack(ops, dtype):
    """The resident feed's embedding gather (data.ResidentSegmentFeed.pack_embeddings): a batch gathered from the row-layout
    table equals sda_pack_rows of the fp32 batch — bit for bit, pad rows and slack included."""
    from speech_decoding_amd import lib as L
    N, F, T, B = 23, 96, 52, 9
    g = torch.Generator().manual_seed(3)
    Y = torch.randn(N, F, T, generator=g)
    Tp, Fp = L.rows_tp(T), L.pad_channels(F)
    table = torch.zeros((N * Tp + L.rows_alloc(1, T) - Tp, Fp), dtype=dtype, device=DEV)
    for k in range(0, N, 7):
        ops.pack_rows(Y[k: k + 7].to(DEV), table[k * Tp:])
    idx = torch.tensor([5, 0, 22, 22, 7, 13, 1, 21, 5])
    got = ops.gather_samples(table, idx.to(DEV), B, T)
    want = to_rows(ops, Y[idx], dtype)
    assert got.shape == want.shape and torch.equal(got.view(torch.uint8).cpu(), want.view(torch.uint8).cpu())
    view = ops.rows_view(got, B, F, T)
    assert torch.equal(view.float().cpu(), Y[idx].to(dtype).float())


def test_clip_merge_rows_equals_the_torch_merge(ops):
    from speech_decoding_amd.distributed import combine_row_stats
    g = torch.Generator().manual_seed(11)
    for world, Bg in [(1, 40), (8, 2048), (3, 257)]:
        t = torch.randn(world, 3, Bg, generator=g)
        t[:, 0] *= 30                                   # row maxima tens apart: exp(m_r - M) underflows for most ranks
        t[:, 1] = t[:, 1].abs() + 1.0
        lse, diag = ops.clip_merge_rows(t.to(DEV).contiguous())
        wl, wd = combine_row_stats(t.double())
        np.testing.assert_allclose(lse.cpu().numpy(), wl.numpy(), rtol=2e-6, atol=2e-6)
        np.testing.assert_allclose(diag.cpu().numpy(), wd.numpy(), rtol=1e-6, atol=1e-6)
