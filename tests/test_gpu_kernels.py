"""GPU parity tests, kernel level: each C-ABI entry point against the CPU oracle on seeded inputs.
Run on the MI355X box:  python -m pytest tests -m gpu"""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _dev():
    assert torch.cuda.is_available(), 'these tests need the GPU box'
    return torch.device('cuda:0')


def grid_from_nchw(a, cs, geo, extent_plus=0, offset=1):
    """numpy NCHW (B,C,h,w) -> flat grid buffer (alloc*cs), stored at (offset, offset)."""
    B, C, h, w = a.shape
    buf = np.zeros((geo.alloc, cs), np.float32)
    g = buf[:geo.NQ].reshape(B, geo.R, geo.P, cs)
    g[:, offset:offset + h, offset:offset + w, :C] = a.transpose(0, 2, 3, 1)
    return buf.reshape(-1)


def nchw_from_grid(buf, cs, C, geo, h, w, offset):
    g = buf.reshape(geo.alloc, cs)[:geo.NQ].reshape(geo.B, geo.R, geo.P, cs)
    return g[:, offset:offset + h, offset:offset + w, :C].transpose(0, 3, 1, 2).copy(), g


CONV_SHAPES = [(27, 70), (70, 70), (280, 280), (280, 1), (2, 2), (280, 108), (108, 108), (8, 8), (32, 32)]


@pytest.mark.parametrize('cin,cout', CONV_SHAPES)
@pytest.mark.parametrize('pad', [1, 0])
@pytest.mark.parametrize('mode', ['f32', 'bf16x6', 'f16x3'])
def test_conv_forward_and_border(oracle, cin, cout, pad, mode, monkeypatch):
    from mmlf_amd import engine
    monkeypatch.setattr(engine, 'CONV_MODE', mode)
    dev = _dev()
    rs = np.random.RandomState(cin * 1000 + cout + pad)
    B, H, W = 3, 9, 13          # odd, non-square; 3*11*15 positions = 2 tiles with a ragged tail
    geo = engine.Geometry(B, H, W)
    w = rs.uniform(-0.5, 0.5, (cout, cin, 2, 2)).astype(np.float32)
    b = rs.uniform(-0.5, 0.5, (cout,)).astype(np.float32)
    cs_in, cs_out = engine.cs_of(cin), engine.cs_of(cout)
    if pad == 1:   # input extent (H,W) at (1,1) -> output extent (H+1,W+1) at (0,0)
        x = rs.uniform(-1, 1, (B, cin, H, W)).astype(np.float32)
        xg = grid_from_nchw(x, cs_in, geo, offset=1)
        shift, vh, vw, oh, ow, ooff = 0, H + 1, W + 1, H + 1, W + 1, 0
    else:          # input extent (H+1,W+1) at (0,0) -> output extent (H,W) at (1,1)
        x = rs.uniform(-1, 1, (B, cin, H + 1, W + 1)).astype(np.float32)
        xg = grid_from_nchw(x, cs_in, geo, offset=0)
        shift, vh, vw, oh, ow, ooff = geo.P + 1, H, W, H, W, 1
    for relu in (False, True):
        ref = oracle.conv2x2(x, w, b, pad, relu=relu)
        tw, tb = torch.from_numpy(w).to(dev), torch.from_numpy(b).to(dev)
        pk = engine.pack_filter(tw, 0, False)
        out = torch.full((geo.alloc * cs_out,), float('nan'), device=dev)
        out[:(geo.P + 1) * cs_out] = 0
        out[geo.NQ * cs_out:] = 0
        engine.conv(geo, torch.from_numpy(xg).to(dev), cs_in, cin, pk, tb, cout, out, cs_out, shift, vh, vw, relu)
        got, g = nchw_from_grid(out.cpu().numpy(), cs_out, cout, geo, oh, ow, ooff)
        np.testing.assert_allclose(got, ref, rtol=2e-5, atol=2e-5)
        # everything outside the stored extent (border, pad channels, slack) must be exactly zero
        full = out.cpu().numpy()
        assert np.isfinite(full).all()
        g2 = g.copy()
        g2[:, ooff:ooff + oh, ooff:ooff + ow, :cout] = 0
        assert not g2.any()
        assert not full.reshape(geo.alloc, cs_out)[geo.NQ:].any()


@pytest.mark.parametrize('mode', ['f32', 'bf16x6', 'f16x3'])
@pytest.mark.parametrize('B,H,W', [(1, 3, 400), (1, 1, 1), (2, 2, 700), (5, 96, 96), (3, 5, 29), (3, 5, 30), (2, 4, 61),
                                   (1, 2, 381), (1, 2, 382)])
def test_conv_wide_and_degenerate_frames(oracle, mode, B, H, W, monkeypatch):
    """Pitch > 383 switches the split kernel to its two-segment activation window; 1x1 images and a
    batch whose position count is not a tile multiple exercise the ragged ends.  Widths 29 / 61 (30) make the
    activation window a whole number of 32-position DMA pieces (one position more), 381 / 382 are the widest
    single-window pitches: the ends of the last-piece clamp."""
    from mmlf_amd import engine
    monkeypatch.setattr(engine, 'CONV_MODE', mode)
    dev = _dev()
    rs = np.random.RandomState(B * 7 + W)
    cin, cout = 70, 70
    geo = engine.Geometry(B, H, W)
    w = rs.uniform(-0.5, 0.5, (cout, cin, 2, 2)).astype(np.float32)
    b = rs.uniform(-0.5, 0.5, (cout,)).astype(np.float32)
    cs_in, cs_out = engine.cs_of(cin), engine.cs_of(cout)
    for pad in (1, 0):
        ih, iw, ioff = (H, W, 1) if pad else (H + 1, W + 1, 0)
        oh, ow, ooff = (H + 1, W + 1, 0) if pad else (H, W, 1)
        x = rs.uniform(-1, 1, (B, cin, ih, iw)).astype(np.float32)
        ref = oracle.conv2x2(x, w, b, pad, relu=True)
        pk = engine.pack_filter(torch.from_numpy(w).to(dev), 0, False)
        out = torch.zeros(geo.alloc * cs_out, device=dev)
        engine.conv(geo, torch.from_numpy(grid_from_nchw(x, cs_in, geo, offset=ioff)).to(dev), cs_in, cin, pk,
                    torch.from_numpy(b).to(dev), cout, out, cs_out, 0 if pad else geo.P + 1, oh, ow, True)
        got, _ = nchw_from_grid(out.cpu().numpy(), cs_out, cout, geo, oh, ow, ooff)
        np.testing.assert_allclose(got, ref, rtol=2e-5, atol=2e-5)


@pytest.mark.parametrize('variant', [0, 1, 2])
def test_filter_variants_equal_image_transforms(oracle, variant):
    """Transposed / transposed+flipped filters on the plain image == plain filters on the
    transformed image (reference feed_forward.py:236-256)."""
    from mmlf_amd import engine
    dev = _dev()
    rs = np.random.RandomState(variant)
    B, S, cin, cout = 2, 12, 27, 70
    geo = engine.Geometry(B, S, S)
    x = rs.uniform(-1, 1, (B, cin, S, S)).astype(np.float32)
    w = rs.uniform(-0.5, 0.5, (cout, cin, 2, 2)).astype(np.float32)
    b = rs.uniform(-0.5, 0.5, (cout,)).astype(np.float32)
    if variant == 0:
        ref = oracle.conv2x2(x, w, b, 1)
    elif variant == 1:
        ref = oracle.conv2x2(np.ascontiguousarray(x.transpose(0, 1, 3, 2)), w, b, 1).transpose(0, 1, 3, 2)
    else:
        xt = np.ascontiguousarray(x.transpose(0, 1, 3, 2)[..., ::-1])
        ref = oracle.conv2x2(xt, w, b, 1)[..., ::-1].transpose(0, 1, 3, 2)
    cs_in, cs_out = engine.cs_of(cin), engine.cs_of(cout)
    pk = engine.pack_filter(torch.from_numpy(w).to(dev), variant, False)
    out = torch.zeros(geo.alloc * cs_out, device=dev)
    engine.conv(geo, torch.from_numpy(grid_from_nchw(x, cs_in, geo)).to(dev), cs_in, cin, pk,
                torch.from_numpy(b).to(dev), cout, out, cs_out, 0, S + 1, S + 1, False)
    got, _ = nchw_from_grid(out.cpu().numpy(), cs_out, cout, geo, S + 1, S + 1, 0)
    np.testing.assert_allclose(got, ref, rtol=2e-5, atol=2e-5)


@pytest.mark.parametrize('cin,cout', [(27, 70), (70, 70), (280, 280), (280, 2), (2, 2), (280, 108), (32, 8), (79, 80), (31, 33)])
@pytest.mark.parametrize('pad', [1, 0])
@pytest.mark.parametrize('variant', [0, 2])
@pytest.mark.parametrize('mode', ['f32', 'bf16x6', 'f16x3'])
def test_conv_backward(oracle, cin, cout, pad, variant, mode, monkeypatch):
    """data gradient (with fused ReLU mask), weight and bias gradients vs the oracle."""
    from mmlf_amd import engine, _lib
    from mmlf_amd._lib import call, ptr
    if variant and cin != cout and cin != 27:
        pytest.skip('stream variants only occur on stream layers')
    monkeypatch.setattr(engine, 'CONV_MODE', mode)
    dev = _dev()
    rs = np.random.RandomState(cin * 7 + cout * 3 + pad + variant)
    B, H, W = 2, 11, 9
    geo = engine.Geometry(B, H, W)
    w = rs.uniform(-0.5, 0.5, (cout, cin, 2, 2)).astype(np.float32)
    cs_in, cs_out = engine.cs_of(cin), engine.cs_of(cout)
    ih, iw, ioff = (H, W, 1) if pad == 1 else (H + 1, W + 1, 0)
    oh, ow, ooff = (H + 1, W + 1, 0) if pad == 1 else (H, W, 1)
    x = rs.uniform(-1, 1, (B, cin, ih, iw)).astype(np.float32)
    x = np.maximum(x, 0)      # plays the role of a post-ReLU activation (mask source)
    go = rs.uniform(-1, 1, (B, cout, oh, ow)).astype(np.float32)
    # the oracle works on the plain filter; a variant filter w_v is the master with taps permuted
    from tests_helpers import variant_filter
    wv = variant_filter(w, variant)
    gin, gw_v, gb = oracle.conv2x2_bwd(x, wv, go, pad)
    gin = gin * (x > 0)
    gw = variant_filter(gw_v, variant, inverse=True)
    xg = torch.from_numpy(grid_from_nchw(x, cs_in, geo, offset=ioff)).to(dev)
    gg = torch.from_numpy(grid_from_nchw(go, cs_out, geo, offset=ooff)).to(dev)
    tw = torch.from_numpy(w).to(dev)
    fwd_shift = 0 if pad == 1 else geo.P + 1
    # weight / bias gradient (accumulate on top of a known value)
    tgw = torch.full((cout, cin, 2, 2), 0.5, device=dev)
    tgb = torch.full((cout,), -0.25, device=dev)
    ws = torch.empty(int(_lib.load().mmlf_wgrad_workspace_floats(cin, cout, B, H, W)), device=dev)
    engine.wgrad(geo, xg, cs_in, cin, gg, cs_out, cout, fwd_shift, tgw, tgb, variant, ws)
    scale = np.abs(gw).max()
    np.testing.assert_allclose(tgw.cpu().numpy() - 0.5, gw, rtol=1e-4, atol=2e-5 * scale)
    np.testing.assert_allclose(tgb.cpu().numpy() + 0.25, gb, rtol=1e-4, atol=2e-5 * np.abs(gb).max())
    # data gradient
    pk = engine.pack_filter(tw, variant, True)
    dx = torch.zeros(geo.alloc * cs_in, device=dev)
    engine.conv(geo, gg, cs_out, cout, pk, None, cin, dx, cs_in, geo.P + 1 - fwd_shift,
                ih, iw, False, ref=xg, cs_ref=cs_in)
    got, _ = nchw_from_grid(dx.cpu().numpy(), cs_in, cin, geo, ih, iw, ioff)
    np.testing.assert_allclose(got, gin, rtol=2e-5, atol=2e-5 * max(1.0, np.abs(gin).max()))


@pytest.mark.parametrize('C,B,H,W', [(70, 3, 10, 14), (70, 2, 33, 139), (6, 1, 5, 6)])
def test_batchnorm_apply_of_four_streams_equals_four_slice_passes(C, B, H, W):
    """mmlf_bn_apply_relu4 (one pass writing whole rows of the concat buffer; reference feed_forward.py:266-267) against
    four mmlf_bn_apply_relu slice passes: the same bits in every position (borders and slack rows included) and the same
    amax array."""
    import ctypes
    from mmlf_amd import engine, _lib
    from mmlf_amd._lib import call, ptr
    dev = _dev()
    geo = engine.Geometry(B, H, W)
    cs, cs_y = engine.cs_of(C), 4 * C
    gen = torch.Generator(device=dev).manual_seed(C + W)
    zs, scs, shs = [], [], []
    for k in range(4):
        z = torch.randn((geo.alloc, cs), device=dev, generator=gen) * (k + 1)      # junk in border positions too
        zs.append(z.reshape(-1))
        scs.append(torch.rand(C, device=dev, generator=gen) + 0.5)
        shs.append(torch.randn(C, device=dev, generator=gen))
    ref = torch.full((geo.alloc * cs_y,), 7.0, device=dev)
    ref_amax = torch.zeros(geo.amax_n, device=dev)
    for k in range(4):
        call('mmlf_bn_apply_relu', ptr(zs[k]), cs, C, ptr(scs[k]), ptr(shs[k]), ptr(ref), cs_y, k * C, C, B, H, W,
             ptr(ref_amax), _lib.stream_ptr())
    got = torch.full((geo.alloc * cs_y,), 7.0, device=dev)
    amax = torch.zeros(geo.amax_n, device=dev)
    arr = lambda ts: (ctypes.c_void_p * 4)(*[ptr(t) for t in ts])
    call('mmlf_bn_apply_relu4', arr(zs), cs, C, arr(scs), arr(shs), ptr(got), cs_y, B, H, W, ptr(amax), _lib.stream_ptr())
    assert torch.equal(got, ref)
    assert torch.equal(amax, ref_amax)              # (row r raises shard r % 64 in both forms)
    assert float(amax[:geo.amax_head].max()) > 0
    # argument checks: odd channel count, a row stride that is not 4 * C
    with pytest.raises(RuntimeError):
        call('mmlf_bn_apply_relu4', arr(zs), cs, C - 1, arr(scs), arr(shs), ptr(got), cs_y, B, H, W, None, _lib.stream_ptr())
    with pytest.raises(RuntimeError):
        call('mmlf_bn_apply_relu4', arr(zs), cs, C, arr(scs), arr(shs), ptr(got), cs_y + 8, B, H, W, None, _lib.stream_ptr())


@pytest.mark.parametrize('C,cs_y,c_off', [(70, 72, 0), (70, 280, 70), (70, 280, 210), (280, 280, 0), (8, 32, 24)])
def test_batchnorm_train_eval_and_backward(oracle, C, cs_y, c_off):
    from mmlf_amd import engine, _lib
    from mmlf_amd._lib import call, ptr
    dev = _dev()
    rs = np.random.RandomState(C + c_off)
    B, H, W = 3, 10, 14
    geo = engine.Geometry(B, H, W)
    cs = engine.cs_of(C)
    z = (rs.normal(size=(B, C, H, W)) * rs.uniform(0.5, 2, (1, C, 1, 1)) + rs.uniform(-3, 3, (1, C, 1, 1))).astype(np.float32)
    gamma = rs.uniform(0.5, 1.5, C).astype(np.float32)
    beta = rs.uniform(-0.5, 0.5, C).astype(np.float32)
    rm, rv = rs.uniform(-1, 1, C).astype(np.float32), rs.uniform(0.5, 2, C).astype(np.float32)
    rm_o, rv_o = rm.copy(), rv.copy()
    y_ref, sm, si = oracle.bn_train(z, gamma, beta, rm_o, rv_o, 0.1)
    t = lambda a: torch.from_numpy(a).to(dev)
    zg = t(grid_from_nchw(z, cs, geo))
    tg, tb, trm, trv = t(gamma), t(beta), t(rm), t(rv)
    coef = torch.empty(4 * C, device=dev)
    part = torch.empty(2 * C * 1024, dtype=torch.float64, device=dev)
    call('mmlf_bn_stats_train', ptr(zg), cs, C, ptr(tg), ptr(tb), ptr(trm), ptr(trv), 0.1, 1e-5, ptr(coef[2 * C:]),
         ptr(coef[3 * C:]), ptr(coef), ptr(coef[C:]), ptr(part), 1024, B, H, W, _lib.stream_ptr())
    np.testing.assert_allclose(coef[2 * C:3 * C].cpu().numpy(), sm, rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(coef[3 * C:].cpu().numpy(), si, rtol=2e-6)
    np.testing.assert_allclose(trm.cpu().numpy(), rm_o, rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(trv.cpu().numpy(), rv_o, rtol=2e-6)
    y = torch.full((geo.alloc * cs_y,), 7.0, device=dev)
    c_store = C if cs_y != cs else cs
    amax = torch.zeros(geo.amax_n, device=dev)
    call('mmlf_bn_apply_relu', ptr(zg), cs, C, ptr(coef), ptr(coef[C:]), ptr(y), cs_y, c_off, c_store, B, H, W,
         ptr(amax), _lib.stream_ptr())
    full = y.cpu().numpy().reshape(geo.alloc, cs_y)
    got, g = nchw_from_grid(full.reshape(-1), cs_y, cs_y, geo, H, W, 1)
    # the amax array the f16 split scales by: max |y| of the tensor, then of every grid row
    hd = geo.amax_head                  # 64 shards of the tensor maximum, then the rows
    assert float(amax[:hd].max()) == float(got[:, c_off:c_off + C].max())
    rows = np.abs(g[..., c_off:c_off + C]).max(axis=(2, 3)).reshape(-1)
    np.testing.assert_array_equal(amax[hd:hd + rows.size].cpu().numpy(), rows)
    assert not amax[hd + rows.size:].any()
    shards = amax[:hd].cpu().numpy().reshape(-1, geo.amax_stride)
    assert not shards[:, 1:].any() and np.count_nonzero(shards[:, 0]) > 1       # spread over slots 256 bytes apart
    np.testing.assert_allclose(got[:, c_off:c_off + C], y_ref, rtol=1e-5, atol=2e-6)
    assert (g[:, 0, :, c_off:c_off + C] == 0).all() and (g[:, :, 0, c_off:c_off + C] == 0).all()
    other = np.delete(full[:geo.NQ], np.s_[c_off:c_off + c_store], axis=1)
    assert (other == 7.0).all()      # a slice write touches nothing else
    # eval coefficients
    erm, erv = t(rm), t(rv)   # keep the tensors alive across the call
    call('mmlf_bn_coeffs_eval', ptr(tg), ptr(tb), ptr(erm), ptr(erv), 1e-5, ptr(coef), ptr(coef[C:]), C,
         _lib.stream_ptr())
    ye = oracle.bn_eval(z, gamma, beta, rm, rv)
    sc, sh = coef[:C].cpu().numpy(), coef[C:2 * C].cpu().numpy()
    np.testing.assert_allclose(np.maximum(z * sc[None, :, None, None] + sh[None, :, None, None], 0), ye, rtol=1e-5, atol=2e-6)
    # backward (train): gradient arrives as a channel slice of a wider buffer
    call('mmlf_bn_stats_train', ptr(zg), cs, C, ptr(tg), ptr(tb), None, None, 0.1, 1e-5, ptr(coef[2 * C:]),
         ptr(coef[3 * C:]), ptr(coef), ptr(coef[C:]), ptr(part), 1024, B, H, W, _lib.stream_ptr())
    gy = rs.normal(size=(B, C, H, W)).astype(np.float32)
    gy_masked = oracle.relu_bwd(y_ref, gy)
    gx_ref, gg_ref, gb_ref = oracle.bn_train_bwd(z, gy_masked, gamma, sm, si)
    wide = np.zeros((B, cs_y, H, W), np.float32)
    wide[:, c_off:c_off + C] = gy
    gyg = t(grid_from_nchw(wide, cs_y, geo))
    k = torch.empty(3 * C, device=dev)
    dgam, dbet = torch.ones(C, device=dev), torch.ones(C, device=dev)
    call('mmlf_bn_bwd_reduce', ptr(gyg), cs_y, c_off, ptr(zg), cs, C, ptr(coef), ptr(coef[C:]), ptr(tg),
         ptr(coef[2 * C:]), ptr(coef[3 * C:]), ptr(dgam), ptr(dbet), 1, ptr(k), ptr(part), 1024, B, H, W, _lib.stream_ptr())
    dz = torch.full((geo.alloc * cs,), float('nan'), device=dev)
    amax.zero_()
    call('mmlf_bn_bwd_apply', ptr(gyg), cs_y, c_off, ptr(zg), cs, C, ptr(coef), ptr(coef[C:]), ptr(coef[2 * C:]),
         ptr(k), ptr(dz), cs, B, H, W, ptr(amax), _lib.stream_ptr())
    assert torch.equal(geo.amax_canonical(amax), geo.amax_of(dz, cs))
    np.testing.assert_allclose(dgam.cpu().numpy() - 1, gg_ref, rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(dbet.cpu().numpy() - 1, gb_ref, rtol=1e-4, atol=1e-4)
    got, g = nchw_from_grid(dz.cpu().numpy(), cs, C, geo, H, W, 1)
    np.testing.assert_allclose(got, gx_ref, rtol=1e-4, atol=1e-5)
    assert (g[:, 0] == 0).all() and (g[:, -1] == 0).all() and (g[:, :, 0] == 0).all() and (g[:, :, -1] == 0).all()


@pytest.mark.parametrize('C,cs,W', [(27, 32, 12), (108, 112, 12), (132, 136, 12),     # 132 = DPP head at 11 views
                                    (280, 280, 140), (288, 288, 5)])
def test_pack_unpack_roundtrip(C, cs, W):
    from mmlf_amd import engine, _lib
    from mmlf_amd._lib import call, ptr
    dev = _dev()
    rs = np.random.RandomState(0)
    B, H = 2, 7
    geo = engine.Geometry(B, H, W)
    x = rs.uniform(-1, 1, (B, C, H, W)).astype(np.float32)
    xd = torch.from_numpy(x).to(dev)
    g = torch.full((geo.alloc * cs,), float('nan'), device=dev)
    g[geo.NQ * cs:] = 0
    amax = torch.zeros(geo.amax_n, device=dev)
    call('mmlf_pack_nchw', ptr(xd), C, ptr(g), cs, B, H, W, ptr(amax), _lib.stream_ptr())
    assert float(amax[:geo.amax_head].max()) == float(np.abs(x).max())
    assert torch.equal(geo.amax_canonical(amax), geo.amax_of(g, cs))
    np.testing.assert_array_equal(g.cpu().numpy(), grid_from_nchw(x, cs, geo))
    back = torch.empty((B, C, H, W), device=dev)
    call('mmlf_unpack_nchw', ptr(g), cs, ptr(back), C, B, H, W, _lib.stream_ptr())
    np.testing.assert_array_equal(back.cpu().numpy(), x)


def test_error_convention():
    from mmlf_amd import _lib
    from mmlf_amd._lib import call
    with pytest.raises(RuntimeError, match='mmlf_conv2x2'):
        call('mmlf_conv2x2', None, 32, 27, None, None, 70, None, 72, 72, 0, 1, 1, 1, 4, 4, 0, None, 0, None)
    x = torch.zeros(16, device=_dev())
    for bad in ((0, 4, 4), (1, 0, 4), (1, 4, -1)):            # empty batch / empty image: rejected, not launched
        with pytest.raises(RuntimeError, match='bad shape'):
            call('mmlf_conv2x2', x.data_ptr(), 32, 27, x.data_ptr(), None, 70, x.data_ptr(), 72, 72, 0, 1, 1, *bad, 0,
                 None, 0, None)
    assert _lib.load().mmlf_grid_alloc_positions(0, 4, 4) == -1
    with pytest.raises(RuntimeError, match='cs_in'):
        call('mmlf_conv2x2', x.data_ptr(), 30, 27, x.data_ptr(), None, 70, x.data_ptr(), 72, 72, 0, 1, 1, 1, 4, 4, 0,
             None, 0, None)


@pytest.mark.parametrize('mode', ['bf16x6', 'f16x3'])
@pytest.mark.parametrize('cin,cout,pad', [(280, 280, 1), (280, 280, 0), (70, 70, 1), (27, 70, 1)])
def test_full_size_adjoint_identities(cin, cout, pad, mode, monkeypatch):
    """BASELINE.json's full size (bs=512, ps=96), where the oracle is too slow: the three conv kernels must be
    each other's adjoints, <conv(x; W), g> = <x, dgrad(g; W)> = <W, wgrad(x, g)> (+ bias term), which does
    not depend on the size.  Inputs live directly on the padded grid (device-side random, zero borders)."""
    from mmlf_amd import engine, _lib
    monkeypatch.setattr(engine, 'CONV_MODE', mode)
    dev = _dev()
    B, H, W = 512, 96, 96
    geo = engine.Geometry(B, H, W)
    cs_in, cs_out = engine.cs_of(cin), engine.cs_of(cout)
    gen = torch.Generator(device=dev).manual_seed(cin + cout + pad)
    ih, iw, ioff = (H, W, 1) if pad == 1 else (H + 1, W + 1, 0)
    oh, ow, ooff = (H + 1, W + 1, 0) if pad == 1 else (H, W, 1)

    def grid_tensor(C, cs, h, w, off):
        t = torch.zeros(geo.alloc * cs, device=dev)
        v = t[:geo.NQ * cs].view(B, geo.R, geo.P, cs)
        v[:, off:off + h, off:off + w, :C] = torch.rand((B, h, w, C), device=dev, generator=gen) - 0.5
        return t

    x = grid_tensor(cin, cs_in, ih, iw, ioff)
    g = grid_tensor(cout, cs_out, oh, ow, ooff)
    w = (torch.rand((cout, cin, 2, 2), device=dev, generator=gen) - 0.5) * 0.1
    bias = torch.rand(cout, device=dev, generator=gen) - 0.5
    fwd_shift = 0 if pad == 1 else geo.P + 1
    out = torch.zeros(geo.alloc * cs_out, device=dev)
    engine.conv(geo, x, cs_in, cin, engine.pack_filter(w, 0, False), bias, cout, out, cs_out, fwd_shift, oh, ow, False)
    dx = torch.zeros(geo.alloc * cs_in, device=dev)
    engine.conv(geo, g, cs_out, cout, engine.pack_filter(w, 0, True), None, cin, dx, cs_in, geo.P + 1 - fwd_shift,
                ih, iw, False)
    gw = torch.zeros_like(w)
    gb = torch.zeros(cout, device=dev)
    ws = torch.empty(int(_lib.load().mmlf_wgrad_workspace_floats(cin, cout, B, H, W)), device=dev)
    engine.wgrad(geo, x, cs_in, cin, g, cs_out, cout, fwd_shift, gw, gb, 0, ws)
    lhs = torch.dot(out.double(), g.double())
    via_x = torch.dot(x.double(), dx.double()) + torch.dot(bias.double(), gb.double())
    via_w = torch.dot(w.double().reshape(-1), gw.double().reshape(-1)) + torch.dot(bias.double(), gb.double())
    scale = float(out.double().abs().mul(g.double().abs()).sum())       # sum |out*g|: the rounding-noise scale
    assert abs(float(lhs - via_x)) <= 1e-6 * scale, (float(lhs), float(via_x), scale)
    assert abs(float(lhs - via_w)) <= 1e-6 * scale, (float(lhs), float(via_w), scale)
    # the bias gradient is the plain column sum of g
    ref_gb = g[:geo.NQ * cs_out].view(-1, cs_out)[:, :cout].double().sum(0)
    assert torch.allclose(gb.double(), ref_gb, rtol=1e-5, atol=1e-5 * float(ref_gb.abs().max()))


def test_split_arithmetic_is_at_f32_accuracy():
    """The split-precision kernels (bf16 3-way / six passes, f16 2-way / three passes) must not be a precision downgrade: against a float64 evaluation
    of the same convolution (K = 4 x 280 products per output, the dominant layer) their error is no larger
    than that of the exact-f32 MFMA kernels."""
    from mmlf_amd import engine
    dev = _dev()
    rs = np.random.RandomState(11)
    B, H, W, cin, cout = 2, 24, 24, 280, 280
    geo = engine.Geometry(B, H, W)
    cs = engine.cs_of(cin)
    x = rs.uniform(-1, 1, (B, cin, H, W)).astype(np.float32)
    w = rs.uniform(-0.06, 0.06, (cout, cin, 2, 2)).astype(np.float32)
    xp = np.zeros((B, cin, H + 2, W + 2), np.float64)
    xp[:, :, 1:-1, 1:-1] = x
    ref = np.zeros((B, cout, H + 1, W + 1), np.float64)
    mag = np.zeros_like(ref)                                    # sum |a*b|: the scale rounding errors live on
    for dy in range(2):
        for dx in range(2):
            patch = xp[:, :, dy:dy + H + 1, dx:dx + W + 1]
            ref += np.einsum('bchw,oc->bohw', patch, w[:, :, dy, dx].astype(np.float64))
            mag += np.einsum('bchw,oc->bohw', np.abs(patch), np.abs(w[:, :, dy, dx]).astype(np.float64))
    xg = torch.from_numpy(grid_from_nchw(x, cs, geo, offset=1)).to(dev)
    err = {}
    for mode in ('f32', 'bf16x6', 'f16x3'):
        engine.CONV_MODE, keep = mode, engine.CONV_MODE
        try:
            pk = engine.pack_filter(torch.from_numpy(w).to(dev), 0, False)
            out = torch.zeros(geo.alloc * cs, device=dev)
            engine.conv(geo, xg, cs, cin, pk, None, cout, out, cs, 0, H + 1, W + 1, False)
        finally:
            engine.CONV_MODE = keep
        got, _ = nchw_from_grid(out.cpu().numpy(), cs, cout, geo, H + 1, W + 1, 0)
        rel = np.abs(got.astype(np.float64) - ref) / mag
        err[mode] = (rel.mean(), rel.max())
    for mode in ('bf16x6', 'f16x3'):
        assert err[mode][0] <= 1.1 * err['f32'][0], err           # mean error relative to sum|a*b|
        assert err[mode][1] <= 1.5 * err['f32'][1], err           # worst element
        assert err[mode][0] < 5e-8, err


@pytest.mark.parametrize('cin,cout', [(280, 280), (70, 70)])
def test_fused_batchnorm_statistics_match_the_two_pass_form(cin, cout):
    """The f16-split convolution accumulates BatchNorm's training statistics of its output in the epilogue
    (mmlf_conv2x2_h2(bn_partial) + mmlf_bn_stats_finalize); they must equal mmlf_bn_stats_train run on the
    stored output, including the running-statistics update."""
    from mmlf_amd import engine, _lib
    from mmlf_amd._lib import call, ptr
    dev = _dev()
    B, H, W = 6, 23, 31
    geo = engine.Geometry(B, H, W)
    cs_in, cs_out = engine.cs_of(cin), engine.cs_of(cout)
    gen = torch.Generator(device=dev).manual_seed(cin)
    x = torch.zeros(geo.alloc * cs_in, device=dev)
    x[:geo.NQ * cs_in].view(B, geo.R, geo.P, cs_in)[:, :H + 1, :W + 1, :cin] = torch.rand((B, H + 1, W + 1, cin), device=dev, generator=gen)
    w = (torch.rand((cout, cin, 2, 2), device=dev, generator=gen) - 0.5) * 0.1
    bias = torch.rand(cout, device=dev, generator=gen) - 0.5
    n = int(_lib.load().mmlf_packed_filter_h2_bytes(cs_in, cout))
    pk = torch.empty(n // 4, device=dev)
    call('mmlf_pack_filter_h2', ptr(w), ptr(pk), cout, cin, 0, 0, _lib.stream_ptr())
    amax = geo.amax_of(x, cs_in)
    z = geo.buf(cs_out, dev)
    nblk = int(_lib.load().mmlf_conv2x2_blocks(cin, cout, B, H, W))
    partial = torch.full((nblk * 2 * cout + 8,), float('nan'), dtype=torch.float64, device=dev)
    call('mmlf_conv2x2_h2', ptr(x), cs_in, cin, ptr(pk), ptr(bias), cout, ptr(z), cs_out, cs_out, geo.P + 1, H, W,
         B, H, W, 0, None, 0, ptr(amax), ptr(z.absmax), ptr(partial), None, None, _lib.stream_ptr())
    gamma, beta = torch.rand(cout, device=dev) + 0.5, torch.rand(cout, device=dev) - 0.5
    out = {}
    for mode in ('fused', 'two_pass'):
        rm, rv = torch.full((cout,), 0.25, device=dev), torch.full((cout,), 2.0, device=dev)
        c = torch.empty(4 * cout, device=dev)
        if mode == 'fused':
            call('mmlf_bn_stats_finalize', ptr(partial), nblk, cout, ptr(gamma), ptr(beta), ptr(rm), ptr(rv), 0.1, 1e-5,
                 ptr(c[2 * cout:]), ptr(c[3 * cout:]), ptr(c), ptr(c[cout:]), B, H, W, _lib.stream_ptr())
        else:
            part = torch.empty(2 * cout * 1024, dtype=torch.float64, device=dev)
            call('mmlf_bn_stats_train', ptr(z), cs_out, cout, ptr(gamma), ptr(beta), ptr(rm), ptr(rv), 0.1, 1e-5,
                 ptr(c[2 * cout:]), ptr(c[3 * cout:]), ptr(c), ptr(c[cout:]), ptr(part), 1024, B, H, W, _lib.stream_ptr())
        out[mode] = (c.cpu().numpy(), rm.cpu().numpy(), rv.cpu().numpy())
    for a, b in zip(out['fused'], out['two_pass']):
        np.testing.assert_allclose(a, b, rtol=2e-6, atol=1e-7)
    assert torch.equal(geo.amax_canonical(z.absmax), geo.amax_of(z, cs_out))      # the conv epilogue's tensor and grid-row maxima (P >= 32: exact)


def test_slack_and_amax_zeroing():
    from mmlf_amd import engine, _lib
    from mmlf_amd._lib import call, ptr
    dev = _dev()
    B, H, W, cs = 3, 5, 7, 8
    geo = engine.Geometry(B, H, W)
    assert geo.amax_n >= geo.amax_head + B * geo.R
    buf = torch.full((geo.alloc * cs,), 3.0, device=dev)
    amax = torch.full((geo.amax_n + 3,), 5.0, device=dev)
    call('mmlf_zero_slack', ptr(buf), cs, B, H, W, ptr(amax), _lib.stream_ptr())
    v = buf.cpu().numpy()
    assert not v[:(geo.P + 1) * cs].any() and not v[geo.NQ * cs:].any()
    assert (v[(geo.P + 1) * cs:geo.NQ * cs] == 3.0).all()
    assert not amax[:geo.amax_n].any() and (amax[geo.amax_n:] == 5.0).all()
    call('mmlf_zero_slack', ptr(buf), cs, B, H, W, None, _lib.stream_ptr())      # the amax array is optional


@pytest.mark.parametrize('cin,cout', [(280, 280), (70, 70), (27, 70)])
def test_relu_bit_mask_replaces_the_activation_reference(cin, cout):
    """The forward convolution leaves (out > 0) as bits (relu_mask_out); the data gradient of the layer above reads
    them (relu_mask_in) instead of the activations (relu_ref): both forms of nn.ReLU's backward must give the same
    gradient bit for bit, on a grid with ragged tiles."""
    from mmlf_amd import engine
    dev = _dev()
    B, H, W = 3, 21, 35
    geo = engine.Geometry(B, H, W)
    cs_in, cs_mid = engine.cs_of(cin), engine.cs_of(cout)
    gen = torch.Generator(device=dev).manual_seed(cin)

    def grid(cs, c, h, w, off):
        t = geo.buf(cs, dev)
        v = t[:geo.NQ * cs].view(B, geo.R, geo.P, cs)
        v.zero_()
        v[:, off:off + h, off:off + w, :c] = torch.rand((B, h, w, c), device=dev, generator=gen) - 0.5
        t.absmax = geo.amax_of(t, cs)
        return t

    x = grid(cs_in, cin, H, W, 1)
    w1 = (torch.rand((cout, cin, 2, 2), device=dev, generator=gen) - 0.5) * 0.2
    b1 = torch.rand(cout, device=dev, generator=gen) - 0.5
    y, mask = geo.buf(cs_mid, dev), geo.relu_mask(dev)
    mask.fill_(-1)
    engine.conv(geo, x, cs_in, cin, engine.pack_filter(w1, 0, False), b1, cout, y, cs_mid, 0, H + 1, W + 1, True, mask_out=mask)
    frac = float((y[:geo.NQ * cs_mid] > 0).float().mean())
    assert 0.05 < frac < 0.6                       # a real mix of kept and dropped elements
    w2 = (torch.rand((cout, cout, 2, 2), device=dev, generator=gen) - 0.5) * 0.2
    dz = grid(cs_mid, cout, H, W, 1)
    pk = engine.pack_filter(w2, 0, True)
    a, b = geo.buf(cs_mid, dev), geo.buf(cs_mid, dev)
    engine.conv(geo, dz, cs_mid, cout, pk, None, cout, a, cs_mid, 0, H + 1, W + 1, False, ref=y, cs_ref=cs_mid)
    engine.conv(geo, dz, cs_mid, cout, pk, None, cout, b, cs_mid, 0, H + 1, W + 1, False, mask_in=mask)
    assert torch.equal(a, b)
    assert torch.equal(geo.amax_canonical(a.absmax), geo.amax_canonical(b.absmax))
    assert float(a.abs().max()) > 0


@pytest.mark.parametrize('cin,cout', [(280, 1), (280, 2), (70, 2)])
@pytest.mark.parametrize('pad', [1, 0])
@pytest.mark.parametrize('variant', [0, 2])
def test_thin_convolution_and_weight_gradient(oracle, cin, cout, pad, variant):
    """mmlf_conv2x2_thin / mmlf_conv2x2_wgrad_thin (the BASE / UPR head's 280 -> 1 | 2 convolution as a
    matrix-vector product) against the oracle: forward with bias + ReLU, zero border, amax rows; weight and bias
    gradient accumulated on top of a known value."""
    from mmlf_amd import engine, _lib
    from mmlf_amd._lib import call, ptr
    from tests_helpers import variant_filter
    dev = _dev()
    rs = np.random.RandomState(cin + 10 * cout + pad + variant)
    B, H, W = 3, 9, 37
    geo = engine.Geometry(B, H, W)
    cs_in, cs_out = engine.cs_of(cin), engine.cs_of(cout)
    w = rs.uniform(-0.5, 0.5, (cout, cin, 2, 2)).astype(np.float32)
    b = rs.uniform(-0.5, 0.5, (cout,)).astype(np.float32)
    ih, iw, ioff = (H, W, 1) if pad == 1 else (H + 1, W + 1, 0)
    oh, ow, ooff = (H + 1, W + 1, 0) if pad == 1 else (H, W, 1)
    shift = 0 if pad == 1 else geo.P + 1
    x = rs.uniform(-1, 1, (B, cin, ih, iw)).astype(np.float32)
    wv = variant_filter(w, variant)
    ref = oracle.conv2x2(x, wv, b, pad, relu=True)
    xg = torch.from_numpy(grid_from_nchw(x, cs_in, geo, offset=ioff)).to(dev)
    out = geo.buf(cs_out, dev)
    out.fill_(float('nan'))
    out.absmax.zero_()
    ws = torch.empty(int(_lib.load().mmlf_conv2x2_thin_workspace_floats(B, H, W)), device=dev)
    tw, tb = torch.from_numpy(w).to(dev), torch.from_numpy(b).to(dev)       # (kept alive: the call takes raw pointers)
    call('mmlf_conv2x2_thin', ptr(xg), cs_in, cin, ptr(tw), ptr(tb), cout,
         ptr(out), cs_out, shift, oh, ow, B, H, W, 1, variant, ptr(ws), ptr(out.absmax), _lib.stream_ptr())
    full = out.cpu().numpy().reshape(geo.alloc, cs_out)
    got, g = nchw_from_grid(full.reshape(-1), cs_out, cout, geo, oh, ow, ooff)
    np.testing.assert_allclose(got, ref, rtol=2e-5, atol=2e-5)
    g2 = g.copy()
    g2[:, ooff:ooff + oh, ooff:ooff + ow, :cout] = 0
    assert not g2.any() and np.isfinite(full[:geo.NQ]).all()          # zero border and pad channels, everything written
    assert torch.equal(geo.amax_canonical(out.absmax), geo.amax_of(torch.nan_to_num(out), cs_out))
    # the engine routes this shape to the thin kernel in every conv mode
    out2 = geo.buf(cs_out, dev)
    engine.conv(geo, xg, cs_in, cin, None, tb, cout, out2, cs_out, shift, oh, ow, True, w_master=tw, variant=variant)
    assert torch.equal(out2[:geo.NQ * cs_out], out[:geo.NQ * cs_out])
    # weight / bias gradient
    go = rs.uniform(-1, 1, (B, cout, oh, ow)).astype(np.float32)
    _, gw_v, gb = oracle.conv2x2_bwd(x, wv, go, pad)
    gw = variant_filter(gw_v, variant, inverse=True)
    gg = torch.from_numpy(grid_from_nchw(go, cs_out, geo, offset=ooff)).to(dev)
    tgw, tgb = torch.full((cout, cin, 2, 2), 0.5, device=dev), torch.full((cout,), -0.25, device=dev)
    engine.wgrad(geo, xg, cs_in, cin, gg, cs_out, cout, shift, tgw, tgb, variant, None)
    np.testing.assert_allclose(tgw.cpu().numpy() - 0.5, gw, rtol=1e-4, atol=2e-5 * np.abs(gw).max())
    np.testing.assert_allclose(tgb.cpu().numpy() + 0.25, gb, rtol=1e-4, atol=2e-5 * np.abs(gb).max())


@pytest.mark.parametrize('cin', [27, 70])
@pytest.mark.parametrize('pad', [1, 0])
def test_register_streamed_conv_on_wide_pitch(oracle, cin, pad):
    """pitches above 127 positions (full frames: the 512x512 ESE scenes) send the 27 -> 70 and 70 -> 70 layers to
    conv4tap_rs_kernel by default (the sixteen-wave tiled variant's window does not fit there); forward with bias and
    ReLU, fused BatchNorm statistics and the ReLU bit mask against the oracle, everything outside the extent exactly zero"""
    from mmlf_amd import engine, _lib
    dev = _dev()
    cout = 70
    rs = np.random.RandomState(cin + pad)
    B, H, W = 2, 5, 139          # pitch 141
    geo = engine.Geometry(B, H, W)
    assert geo.P > 127
    w = rs.uniform(-0.5, 0.5, (cout, cin, 2, 2)).astype(np.float32)
    b = rs.uniform(-0.5, 0.5, (cout,)).astype(np.float32)
    cs_in, cs_out = engine.cs_of(cin), engine.cs_of(cout)
    if pad == 1:
        x = rs.uniform(-1, 1, (B, cin, H, W)).astype(np.float32)
        xg = grid_from_nchw(x, cs_in, geo, offset=1)
        shift, vh, vw, oh, ow, ooff = 0, H + 1, W + 1, H + 1, W + 1, 0
    else:
        x = rs.uniform(-1, 1, (B, cin, H + 1, W + 1)).astype(np.float32)
        xg = grid_from_nchw(x, cs_in, geo, offset=0)
        shift, vh, vw, oh, ow, ooff = geo.P + 1, H, W, H, W, 1
    tw, tb = torch.from_numpy(w).to(dev), torch.from_numpy(b).to(dev)
    pk = engine.pack_filter(tw, 0, False)
    xd = torch.from_numpy(xg).to(dev)
    for relu in (False, True):
        ref = oracle.conv2x2(x, w, b, pad, relu=relu)
        out = geo.buf(cs_out, dev)
        mask = torch.zeros_like(geo.relu_mask(dev)) if relu else None
        ws = engine._Workspace.get(dev)
        ws.partial.fill_(float('nan'))
        engine.conv(geo, xd, cs_in, cin, pk, tb, cout, out, cs_out, shift, vh, vw, relu, mask_out=mask,
                    bn_partial=None if relu else ws.partial)
        got, g = nchw_from_grid(out.cpu().numpy(), cs_out, cout, geo, oh, ow, ooff)
        np.testing.assert_allclose(got, ref, rtol=2e-5, atol=2e-5)
        g2 = g.copy()
        g2[:, ooff:ooff + oh, ooff:ooff + ow, :cout] = 0
        assert not g2.any()
        true = geo.amax_of(out, cs_out)
        got_amax = geo.amax_canonical(out.absmax)
        assert float(got_amax[0]) == float(true[0]) and bool((got_amax >= true).all())
        if relu:
            bits = int(np.unpackbits(mask.cpu().numpy().view(np.uint8)).sum())
            assert bits == int((got > 0).sum())                      # one bit per positive stored output
        else:
            nblk = int(_lib.load().mmlf_conv2x2_blocks(cin, cout, B, H, W))
            part = ws.partial[:nblk * 2 * cout].view(nblk, 2, cout).sum(0).cpu().numpy()
            assert np.isfinite(part).all()
            np.testing.assert_allclose(part[0], ref.sum(axis=(0, 2, 3)), rtol=1e-4, atol=1e-3)
            np.testing.assert_allclose(part[1], (ref.astype(np.float64) ** 2).sum(axis=(0, 2, 3)), rtol=1e-4)


def test_narrow_conv_kernel_variants_agree(tmp_path):
    """The three kernels that can run a 70-channel layer: the tiled eight-wave kernel, its 512-position / sixteen-wave
    variant, and the register-streamed kernel (round 4, the default: whole filter in LDS, activations global ->
    registers, no barrier in the loop).  All index masks, statistics and scales by the global 32-position group.  The two
    tiled kernels add every accumulator's products in the same order: same BYTES (output, ReLU mask words, row maxima,
    masked data gradient; BatchNorm partial sums equal once summed over workgroups).  The register-streamed kernel walks
    K tap by tap (32 channels of one tap per matrix instruction instead of 8 channels of four taps): the same sums in
    another order -- equal to float32 rounding, mask bits equal except where an output is within rounding of zero.
    Compared across processes (the switches are read once per process)."""
    import subprocess
    import sys
    script = r'''
import sys, numpy as np, torch
sys.path.insert(0, %r)
from mmlf_amd import engine, _lib
dev = torch.device('cuda:0')
B, H, W, cin, cout = 5, 37, 41, 70, 70
geo = engine.Geometry(B, H, W)
rs = np.random.RandomState(3)
cs = engine.cs_of(cin)
x = geo.buf(cs, dev)
v = x[:geo.NQ * cs].view(B, geo.R, geo.P, cs)
v[:, 1:H + 1, 1:W + 1, :cin] = torch.from_numpy(rs.standard_normal((B, H, W, cin)).astype(np.float32) * np.exp(rs.uniform(-6, 6, (B, H, 1, 1))).astype(np.float32)).to(dev)
x.absmax = geo.amax_of(x, cs)
w = torch.from_numpy(rs.uniform(-0.5, 0.5, (cout, cin, 2, 2)).astype(np.float32)).to(dev)
b = torch.from_numpy(rs.uniform(-0.5, 0.5, (cout,)).astype(np.float32)).to(dev)
pk = engine.pack_filter(w, 0, False)
y = geo.buf(cs, dev)
mask = geo.relu_mask(dev)
engine.conv(geo, x, cs, cin, pk, b, cout, y, cs, 0, H + 1, W + 1, True, mask_out=mask)
ws = engine._Workspace.get(dev)
z = geo.buf(cs, dev)
engine.conv(geo, y, cs, cout, pk, b, cout, z, cs, geo.P + 1, H, W, False, bn_partial=ws.partial)
nblk = int(_lib.load().mmlf_conv2x2_blocks(cout, cout, B, H, W))
part = ws.partial[:nblk * 2 * cout].double().view(nblk, 2, cout).sum(0)
g = geo.buf(cs, dev)
engine.conv(geo, z, cs, cout, engine.pack_filter(w, 0, True), None, cin, g, cs, 0, H + 1, W + 1, False, mask_in=mask)
torch.cuda.synchronize()
np.savez(sys.argv[1], y=y.cpu().numpy(), z=z.cpu().numpy(), g=g.cpu().numpy(), mask=mask.cpu().numpy(),
         ay=geo.amax_canonical(y.absmax).cpu().numpy(), az=geo.amax_canonical(z.absmax).cpu().numpy(), part=part.cpu().numpy())
''' % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    outs = []
    for rs_on, nw in (('0', '0'), ('0', '1'), ('1', '1')):
        out = str(tmp_path / f'rs{rs_on}_nw{nw}.npz')
        env = dict(os.environ, MMLF_CONV_NW16=nw, MMLF_CONV_RS=rs_on)
        res = subprocess.run([sys.executable, '-c', script, out], env=env, capture_output=True, text=True, timeout=600)
        assert res.returncode == 0, res.stderr[-2000:]
        outs.append(np.load(out))
    a, b, c = outs
    for key in ('y', 'z', 'g', 'mask', 'ay', 'az'):
        assert np.array_equal(a[key], b[key]), key
    # the per-workgroup partial sums are grouped differently (positions per workgroup differ): equal after the sum
    np.testing.assert_allclose(a['part'], b['part'], rtol=1e-12)
    for key in ('y', 'z', 'g'):
        scale = np.abs(a[key]).max()
        assert np.abs(a[key] - c[key]).max() <= 4e-6 * scale, (key, np.abs(a[key] - c[key]).max() / scale)
    flips = np.unpackbits((a['mask'] ^ c['mask']).view(np.uint8)).sum() / (a['mask'].size * 32.0)
    assert flips <= 1e-4, flips                       # (out > 0) of outputs within rounding of zero
    np.testing.assert_allclose(a['ay'], c['ay'], rtol=2e-6)
    np.testing.assert_allclose(a['az'], c['az'], rtol=2e-6)
    np.testing.assert_allclose(a['part'], c['part'], rtol=1e-5, atol=1e-5 * np.abs(a['part']).max())


def test_batched_filter_packing_and_slack_zeroing_equal_the_per_layer_calls():
    """mmlf_pack_filters_h2 (every filter of a step from one launch) and mmlf_zero_slack4 (a block's buffers from one
    launch) must write the bytes of the per-filter / per-buffer entry points"""
    from mmlf_amd import engine, _lib
    from mmlf_amd._lib import call, ptr
    dev = _dev()
    gen = torch.Generator(device=dev).manual_seed(12)
    shapes = [(70, 27, 1, False), (70, 70, 0, False), (70, 70, 2, True), (280, 280, 0, False), (280, 280, 0, True),
              (1, 280, 0, True), (108, 280, 0, False), (108, 108, 0, True), (2, 2, 0, False)]
    ws = [(torch.rand((co, ci, 2, 2), device=dev, generator=gen) - 0.5) * (10.0 ** (k - 4)) for k, (co, ci, _, _) in enumerate(shapes)]
    items = [((f'w{k}', var, dg), w, var, dg) for k, (w, (_, _, var, dg)) in enumerate(zip(ws, shapes))]
    work = engine._Workspace.get(dev)
    packs = work.packed_filters(items)
    for (key, w, var, dg) in items:
        single = engine.pack_filter(w, var, dg)
        assert packs[key].numel() == single.numel(), key
        assert torch.equal(packs[key].view(torch.int32), single.view(torch.int32)), key
    # a second call with the same weight storage reuses table and buffers and re-packs the (changed) contents
    ws[3].mul_(3.0)
    again = work.packed_filters(items)
    assert again[items[3][0]].data_ptr() == packs[items[3][0]].data_ptr()
    assert torch.equal(again[items[3][0]].view(torch.int32), engine.pack_filter(ws[3], 0, False).view(torch.int32))
    # zero_slack4
    geo = engine.Geometry(3, 7, 11)
    css = [8, 72, 280]
    a = [torch.full((geo.alloc * cs,), float('nan'), device=dev) for cs in css]
    am = [torch.full((geo.amax_n,), float('nan'), device=dev) for _ in css]
    b = [t.clone() for t in a]
    bm = [t.clone() for t in am]
    import ctypes
    call('mmlf_zero_slack4', (ctypes.c_void_p * 4)(*[ptr(t) for t in a], None), (ctypes.c_int * 4)(*css, 0),
         (ctypes.c_void_p * 4)(*[ptr(t) for t in am], None), geo.B, geo.H, geo.W, _lib.stream_ptr())
    for t, m, cs in zip(b, bm, css):
        call('mmlf_zero_slack', ptr(t), cs, geo.B, geo.H, geo.W, ptr(m), _lib.stream_ptr())
    for x, y in zip(a + am, b + bm):
        assert torch.equal(torch.nan_to_num(x, nan=-7.0), torch.nan_to_num(y, nan=-7.0))
    assert all(float(m.abs().sum()) == 0 for m in am)


@pytest.mark.parametrize('cin,cout', [(70, 70), (280, 280), (27, 70)])
def test_transposed_epilogue_writes_the_same_bits(tmp_path, cin, cout):
    """Round 5: the launch kinds plain / ReLU / ReLU + mask-out / mask-in run with TRANSPOSED accumulator tiles (weights as
    the matrix instruction's A operand): a lane holds four consecutive channels of one position and stores 16 bytes at a
    time (csrc/conv.hip conv_epilogue16_tr).  The values are formed by the same operations as in the other orientation, so
    every stored byte, every row maximum and the NUMBER of ReLU bits must be identical with MMLF_CONV_TR=0 and =1 (the
    mask words themselves have another layout); the masked data gradient -- which reads those words -- as well.  Also
    against a channel slice at an odd offset, which the transposed form must decline (8-byte aligned rows)."""
    import subprocess
    import sys
    script = r'''
import sys, numpy as np, torch
sys.path.insert(0, %r)
from mmlf_amd import engine, _lib
dev = torch.device('cuda:0')
cin, cout = int(sys.argv[2]), int(sys.argv[3])
B, H, W = 3, 19, 45
geo = engine.Geometry(B, H, W)
rs = np.random.RandomState(11)
cs_in, cs = engine.cs_of(cin), engine.cs_of(cout)
x = geo.buf(cs_in, dev)
v = x[:geo.NQ * cs_in].view(B, geo.R, geo.P, cs_in)
v[:, 1:H + 1, 1:W + 1, :cin] = torch.from_numpy(rs.standard_normal((B, H, W, cin)).astype(np.float32) * np.exp(rs.uniform(-4, 4, (B, H, 1, 1))).astype(np.float32)).to(dev)
x.absmax = geo.amax_of(x, cs_in)
w1 = torch.from_numpy(rs.uniform(-0.3, 0.3, (cout, cin, 2, 2)).astype(np.float32)).to(dev)
w2 = torch.from_numpy(rs.uniform(-0.3, 0.3, (cout, cout, 2, 2)).astype(np.float32)).to(dev)
b = torch.from_numpy(rs.uniform(-0.5, 0.5, (cout,)).astype(np.float32)).to(dev)
y, mask = geo.buf(cs, dev), torch.zeros_like(geo.relu_mask(dev))
engine.conv(geo, x, cs_in, cin, engine.pack_filter(w1, 0, False), b, cout, y, cs, 0, H + 1, W + 1, True, mask_out=mask)   # ReLU + mask-out
y2 = geo.buf(cs, dev)
engine.conv(geo, x, cs_in, cin, engine.pack_filter(w1, 0, False), b, cout, y2, cs, 0, H + 1, W + 1, True)                 # ReLU
z = geo.buf(cs, dev)
engine.conv(geo, y, cs, cout, engine.pack_filter(w2, 0, False), b, cout, z, cs, geo.P + 1, H, W, False)                   # plain, pad 0
g = geo.buf(cs, dev)
engine.conv(geo, z, cs, cout, engine.pack_filter(w2, 0, True), None, cout, g, cs, 0, H + 1, W + 1, False, mask_in=mask)   # mask-in
# a channel slice of a wider buffer at an offset that is not a multiple of four floats (evaluation: the streams' folded
# last convolution writes [70 s, 70 s + 70) of the concat buffer): must take the other form, whatever the switch says
wide = geo.buf(4 * cs, dev)
engine.conv(geo, y, cs, cout, engine.pack_filter(w2, 0, False), b, cout, wide, 4 * cs, geo.P + 1, H, W, True, n_store=cout, out_off=cout + 2)
torch.cuda.synchronize()
np.savez(sys.argv[1], y=y.cpu().numpy(), y2=y2.cpu().numpy(), z=z.cpu().numpy(), g=g.cpu().numpy(), wide=wide.cpu().numpy(),
         bits=np.unpackbits(mask.cpu().numpy().view(np.uint8)).sum(), ay=geo.amax_canonical(y.absmax).cpu().numpy(),
         az=geo.amax_canonical(z.absmax).cpu().numpy(), ag=geo.amax_canonical(g.absmax).cpu().numpy())
''' % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    outs = []
    for tr in ('0', '1'):
        out = str(tmp_path / f'tr{tr}.npz')
        res = subprocess.run([sys.executable, '-c', script, out, str(cin), str(cout)], env=dict(os.environ, MMLF_CONV_TR=tr),
                             capture_output=True, text=True, timeout=600)
        assert res.returncode == 0, res.stderr[-3000:]
        outs.append(np.load(out))
    a, b = outs
    assert int(a['bits']) == int(b['bits']) == int((a['y'] > 0).sum())
    for key in ('y', 'y2', 'z', 'g', 'wide', 'ay', 'az', 'ag'):
        assert np.array_equal(a[key], b[key]), key
    assert np.array_equal(a['y'], a['y2'])
