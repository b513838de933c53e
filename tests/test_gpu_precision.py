"""Precision of the default arithmetic (f16x3: 2 x f16 operands of 22 significant bits, three MFMA passes)
where it could break: tensors whose magnitude differs by many orders INSIDE one tensor -- between patches,
between channels, between image regions.  The reference arithmetic is float32 nn.Conv2d (reference
mmlf/model/feed_forward.py:123,125): a region of small values keeps its full relative precision there.
Every case compares the f16x3 kernels and the exact-f32 MFMA kernels with a float64 evaluation of the same
convolution and requires f16x3 to be no worse than float32 by more than a stated factor, PER PATCH / PER
CHANNEL, not only in aggregate.  (The kernels take their operand scales per 32-position wave from the maxima
of the grid rows its taps read and per output channel for the weights: include/mmlf_hip.h, mmlf_conv2x2_h2.)
Run on the MI355X box:  python -m pytest tests -m gpu"""
import numpy as np
import pytest
import torch

from test_gpu_kernels import grid_from_nchw, nchw_from_grid, _dev

pytestmark = pytest.mark.gpu


def conv_f64(x, w, pad):
    """float64 cross-correlation (k=2, pad 1|0) and sum |a*b| (the scale rounding errors live on)."""
    B, C, H, W = x.shape
    if pad:
        xp = np.zeros((B, C, H + 2, W + 2), np.float64)
        xp[:, :, 1:-1, 1:-1] = x
        oh, ow = H + 1, W + 1
    else:
        xp = x.astype(np.float64)
        oh, ow = H - 1, W - 1
    ref = np.zeros((B, w.shape[0], oh, ow), np.float64)
    mag = np.zeros_like(ref)
    for dy in range(2):
        for dx in range(2):
            patch = xp[:, :, dy:dy + oh, dx:dx + ow]
            ref += np.einsum('bchw,oc->bohw', patch, w[:, :, dy, dx].astype(np.float64))
            mag += np.einsum('bchw,oc->bohw', np.abs(patch), np.abs(w[:, :, dy, dx]).astype(np.float64))
    return ref, mag


def run_conv(engine, mode, geo, x, w, pad, H, W, dgrad=False, ref_mask=None):
    """forward conv (or, dgrad=True, the data gradient: x plays the output gradient) through the C ABI"""
    dev = _dev()
    cout, cin = w.shape[0], w.shape[1]
    K, N = (cout, cin) if dgrad else (cin, cout)
    cs_in, cs_out = engine.cs_of(K), engine.cs_of(N)
    keep, engine.CONV_MODE = engine.CONV_MODE, mode
    try:
        pk = engine.pack_filter(torch.from_numpy(w).to(dev), 0, dgrad)
        out = torch.zeros(geo.alloc * cs_out, device=dev)
        if not dgrad:
            ioff = 1 if pad else 0
            shift, oh, ow, ooff = (0, H + 1, W + 1, 0) if pad else (geo.P + 1, H, W, 1)
        else:       # gradient of a pad-1 conv lives at offset 0 and yields extent (H, W) at (1,1); pad 0 the reverse
            ioff = 0 if pad else 1
            shift, oh, ow, ooff = (geo.P + 1, H, W, 1) if pad else (0, H + 1, W + 1, 0)
        xg = torch.from_numpy(grid_from_nchw(x, cs_in, geo, offset=ioff)).to(dev)
        engine.conv(geo, xg, cs_in, K, pk, None, N, out, cs_out, shift, oh, ow, False)
    finally:
        engine.CONV_MODE = keep
    got, _ = nchw_from_grid(out.cpu().numpy(), cs_out, N, geo, oh, ow, ooff)
    return got.astype(np.float64)


@pytest.mark.parametrize('cin,cout', [(280, 280), (70, 70)])
@pytest.mark.parametrize('pad', [1, 0])
def test_conv_patches_of_very_different_magnitude(cin, cout, pad):
    """patch b scaled by 2^(-10 b) (1 ... 9.3e-10, big and small patches adjacent in memory, both orders):
    every patch's error relative to ITS OWN sum |a*b| stays at the float32 kernel's level."""
    from mmlf_amd import engine
    rs = np.random.RandomState(cin + pad)
    B, H, W = 6, 33, 37                       # pitch 39 >= 32
    geo = engine.Geometry(B, H, W)
    ih, iw = (H, W) if pad else (H + 1, W + 1)
    expo = np.array([0, -10, -20, -30, 0, -30])          # small after big and big after small
    x = rs.uniform(-1, 1, (B, cin, ih, iw)).astype(np.float32) * np.exp2(expo).astype(np.float32)[:, None, None, None]
    w = rs.uniform(-0.06, 0.06, (cout, cin, 2, 2)).astype(np.float32)
    ref, mag = conv_f64(x, w, pad)
    err = {}
    for mode in ('f32', 'f16x3'):
        got = run_conv(engine, mode, geo, x, w, pad, H, W)
        rel = np.abs(got - ref) / mag
        err[mode] = (rel.mean(axis=(1, 2, 3)), rel.max(axis=(1, 2, 3)))
    for b in range(B):
        assert err['f16x3'][0][b] <= 1.25 * err['f32'][0][b], (b, err)       # mean error of patch b
        assert err['f16x3'][1][b] <= 1.5 * err['f32'][1][b], (b, err)        # its worst element
        assert err['f16x3'][0][b] < 5e-8, (b, err)


@pytest.mark.parametrize('pad', [1, 0])
def test_dgrad_patches_of_very_different_magnitude(pad):
    """the same for the data gradient (packed filter transposed and tap-reversed)"""
    from mmlf_amd import engine
    rs = np.random.RandomState(5 + pad)
    B, H, W, cin, cout = 5, 33, 37, 280, 280
    geo = engine.Geometry(B, H, W)
    gh, gw_ = (H + 1, W + 1) if pad else (H, W)
    expo = np.array([0, -12, -24, 0, -36])
    g = rs.normal(size=(B, cout, gh, gw_)).astype(np.float32) * np.exp2(expo).astype(np.float32)[:, None, None, None]
    w = rs.uniform(-0.06, 0.06, (cout, cin, 2, 2)).astype(np.float32)
    # dgrad of a pad-p conv = conv of the gradient with the flipped, transposed filter at pad 1-p
    wt = np.ascontiguousarray(w.transpose(1, 0, 2, 3)[:, :, ::-1, ::-1])
    ref, mag = conv_f64(g, wt, 1 - pad)
    err = {}
    for mode in ('f32', 'f16x3'):
        got = run_conv(engine, mode, geo, g, w, pad, H, W, dgrad=True)
        rel = np.abs(got - ref) / mag
        err[mode] = (rel.mean(axis=(1, 2, 3)), rel.max(axis=(1, 2, 3)))
    for b in range(B):
        assert err['f16x3'][0][b] <= 1.25 * err['f32'][0][b], (b, err)
        assert err['f16x3'][1][b] <= 1.5 * err['f32'][1][b], (b, err)


def test_conv_channels_of_very_different_magnitude():
    """output channel n's filter scaled by 2^(-7 (n % 5)) and input channel c scaled by 2^(-5 (c % 4)): every
    OUTPUT CHANNEL's error relative to its own sum |a*b| stays at the float32 kernel's level (the packed filter
    carries one power-of-two scale per output channel; small input channels only lose bits that float32
    accumulation loses as well)."""
    from mmlf_amd import engine
    rs = np.random.RandomState(3)
    B, H, W, cin, cout = 2, 33, 37, 280, 280
    geo = engine.Geometry(B, H, W)
    x = rs.uniform(-1, 1, (B, cin, H, W)).astype(np.float32) * np.exp2(-5.0 * (np.arange(cin) % 4)).astype(np.float32)[None, :, None, None]
    w = rs.uniform(-0.06, 0.06, (cout, cin, 2, 2)).astype(np.float32) * np.exp2(-7.0 * (np.arange(cout) % 5)).astype(np.float32)[:, None, None, None]
    ref, mag = conv_f64(x, w, 1)
    err = {}
    for mode in ('f32', 'f16x3'):
        rel = np.abs(run_conv(engine, mode, geo, x, w, 1, H, W) - ref) / mag
        err[mode] = (rel.mean(axis=(0, 2, 3)), rel.max(axis=(0, 2, 3)))
    assert (err['f16x3'][0] <= 1.25 * err['f32'][0]).all(), err
    assert (err['f16x3'][1] <= 1.6 * err['f32'][1]).all(), err


def test_conv_rows_of_very_different_magnitude():
    """magnitude halving with every image row inside ONE image (row 95 is 2^-95 of row 0): a wave's scale comes
    from the 2-3 grid rows it reads, so every output row keeps float32-level error relative to its own
    sum |a*b|.  (What this arithmetic does NOT give: float32's precision for a small value next to a value
    2^18 larger inside the same 2-3 rows -- there the small one's products are rounded at 2^-40 of the large one.)"""
    from mmlf_amd import engine
    rs = np.random.RandomState(8)
    B, H, W, cin, cout = 1, 96, 96, 70, 70
    geo = engine.Geometry(B, H, W)
    ramp = np.exp2(-1.0 * np.arange(H)).astype(np.float32)
    x = rs.uniform(-1, 1, (B, cin, H, W)).astype(np.float32) * ramp[None, None, :, None]
    w = rs.uniform(-0.1, 0.1, (cout, cin, 2, 2)).astype(np.float32)
    ref, mag = conv_f64(x, w, 1)
    err = {}
    for mode in ('f32', 'f16x3'):
        rel = np.abs(run_conv(engine, mode, geo, x, w, 1, H, W) - ref) / mag
        err[mode] = rel.mean(axis=(0, 1, 3))                      # per output row
    assert (err['f16x3'] <= 1.5 * err['f32'] + 1e-9).all(), (err['f16x3'] / err['f32']).max()


@pytest.mark.parametrize('cin,cout', [(280, 280), (70, 70)])
def test_wgrad_patches_of_very_different_magnitude(cin, cout):
    """weight gradient = a sum over ALL positions: (a) every patch's activations scaled by 2^(-10 b) with
    gradients of one scale (patch 0 dominates the sum), (b) activations scaled by 2^(-10 b) AND gradients by
    2^(+10 b), so that every patch contributes equally although its operands sit 2^30 below their tensor's
    maximum.  Error relative to sum |in * g| must stay at the float32 kernel's level in both."""
    from mmlf_amd import engine, _lib
    dev = _dev()
    rs = np.random.RandomState(cin)
    B, H, W = 4, 33, 37
    geo = engine.Geometry(B, H, W)
    cs_in, cs_out = engine.cs_of(cin), engine.cs_of(cout)
    x0 = rs.uniform(-1, 1, (B, cin, H, W)).astype(np.float32)
    g0 = rs.normal(size=(B, cout, H + 1, W + 1)).astype(np.float32)
    for g_expo in (0.0, 10.0):
        sx = np.exp2(-10.0 * np.arange(B)).astype(np.float32)[:, None, None, None]
        sg = np.exp2(g_expo * np.arange(B)).astype(np.float32)[:, None, None, None]
        x, g = x0 * sx, g0 * sg
        xp = np.zeros((B, cin, H + 2, W + 2), np.float64)
        xp[:, :, 1:-1, 1:-1] = x
        ref = np.zeros((cout, cin, 2, 2), np.float64)
        mag = np.zeros_like(ref)
        for dy in range(2):
            for dx in range(2):
                patch = xp[:, :, dy:dy + H + 1, dx:dx + W + 1]
                ref[:, :, dy, dx] = np.einsum('bohw,bchw->oc', g.astype(np.float64), patch)
                mag[:, :, dy, dx] = np.einsum('bohw,bchw->oc', np.abs(g).astype(np.float64), np.abs(patch))
        xg = torch.from_numpy(grid_from_nchw(x, cs_in, geo, offset=1)).to(dev)
        gg = torch.from_numpy(grid_from_nchw(g, cs_out, geo, offset=0)).to(dev)
        ws = torch.empty(int(_lib.load().mmlf_wgrad_workspace_floats(cin, cout, B, H, W)), device=dev)
        err = {}
        for mode in ('f32', 'f16x3'):
            keep, engine.CONV_MODE = engine.CONV_MODE, mode
            try:
                gw, gb = torch.zeros((cout, cin, 2, 2), device=dev), torch.zeros(cout, device=dev)
                engine.wgrad(geo, xg, cs_in, cin, gg, cs_out, cout, 0, gw, gb, 0, ws)
            finally:
                engine.CONV_MODE = keep
            rel = np.abs(gw.cpu().numpy().astype(np.float64) - ref) / mag
            gb_ref = g.astype(np.float64).sum(axis=(0, 2, 3))
            gb_rel = np.abs(gb.cpu().numpy() - gb_ref) / np.abs(g).astype(np.float64).sum(axis=(0, 2, 3))
            err[mode] = (rel.mean(), rel.max(), gb_rel.max())
        assert err['f16x3'][0] <= 1.25 * err['f32'][0], (g_expo, err)
        assert err['f16x3'][1] <= 2.0 * err['f32'][1], (g_expo, err)
        assert err['f16x3'][2] <= 2.0 * err['f32'][2] + 1e-8, (g_expo, err)
