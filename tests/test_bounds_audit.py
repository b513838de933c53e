"""Bounds audit of the convolution / weight-gradient launches (round 5; no GPU needed: only size queries and the
host-side extent derivations of the library are called).

Why it exists: in round 4 one invocation of tools/kbench.py died with `Memory access fault by GPU` 0.3 s after its first
GPU activity and the identical command passed on the next lease.  The cause could not be reproduced, so every launch that
tool makes before its first timed loop -- and every launch kind of a training step -- has its extents DERIVED
(mmlf_audit_conv_h2 / mmlf_audit_wgrad_h2, csrc/conv.hip and csrc/wgrad.hip) and held here against what the ABI's size queries tell a caller
to allocate, over a sweep of shapes (training patches, tiny images, non-square frames, full 512x512 frames; every channel
combination the network has).  tests/test_gpu_bounds.py is the GPU half: a -DMMLF_BOUNDS_DEBUG build counts real accesses.

Replaces nothing in the reference; the kernels audited replace nn.Conv2d forward / backward
(reference mmlf/model/feed_forward.py:123-125, autograd via mmlf/train/cli.py:257).
"""
import ctypes

import pytest

from mmlf_amd import _lib

CONV = dict(IN=0, PACKED=1, BIAS=2, OUT=3, REF=4, IN_AMAX=5, OUT_AMAX=6, BN_PARTIAL=7, MASK=8)
WG = dict(IN=0, G=1, GW=2, GB=3, WORKSPACE=4, IN_AMAX=5, G_AMAX=6)
# what mmlf_amd.engine allocates once per (device, thread) for the fused BatchNorm statistics: engine._Workspace.partial
PARTIAL_BYTES = (2 * 512 * 1024 + 8) * 8

SHAPES = [(512, 96, 96), (64, 96, 96), (1, 96, 96), (3, 16, 16), (2, 9, 9), (5, 5, 6), (7, 32, 24), (2, 130, 100),
          (1, 512, 512), (70, 512, 512), (8, 127, 126), (8, 125, 129)]
# (Cin, Cout) of every convolution the default network and its DPP / UPR heads hold (thin 280 -> 1|2 layers run other kernels)
LAYERS = [(27, 70), (70, 70), (280, 280), (280, 108), (108, 108), (2, 2), (1, 1)]


def cs_of(c):
    return (c + 7) // 8 * 8


def lib():
    return _lib.load()


@pytest.mark.parametrize('B,H,W', SHAPES)
def test_conv_launch_extents_fit_the_allocations_the_abi_prescribes(B, H, W):
    L = lib()
    P = W + L.mmlf_grid_pad_w()
    alloc = L.mmlf_grid_alloc_positions(B, H, W)
    amax = L.mmlf_amax_entries(B, H, W) * 4
    mask = L.mmlf_relu_mask_words(B, H, W) * 4
    if alloc * 288 * 4 >= 2 ** 63 or B * (H + 2) * P + 2 * P + 600 >= 2 ** 31:
        pytest.skip('engine.Geometry refuses this size')
    for cin, cout in LAYERS:
        for dgrad in (False, True):
            K, N = (cout, cin) if dgrad else (cin, cout)
            cs_in, cs_out = cs_of(K), cs_of(N)
            if L.mmlf_packed_filter_h2_columns(N) < 0:
                continue
            for out_shift in (0, P + 1):
                for n_store, c_off in ((cs_out, 0), (N, 0), (N, cs_out - N)):       # whole rows / exact channels / a channel slice
                    e = (ctypes.c_int64 * 9)()
                    assert L.mmlf_audit_conv_h2(cs_in, K, N, cs_out, n_store, out_shift, cs_out, B, H, W, e) == 0, _lib.last_error()
                    tag = f'{K}->{N} B={B} {H}x{W} shift={out_shift} n_store={n_store}'
                    assert e[CONV['IN']] <= alloc * cs_in * 4, tag
                    assert e[CONV['PACKED']] == L.mmlf_packed_filter_h2_bytes(cs_in, N), tag
                    assert e[CONV['BIAS']] == N * 4, tag
                    assert e[CONV['OUT']] <= alloc * cs_out * 4 - c_off * 4, tag      # `out` as passed = base + c_off floats
                    assert e[CONV['REF']] <= alloc * cs_out * 4, tag
                    assert e[CONV['IN_AMAX']] <= amax and e[CONV['OUT_AMAX']] <= amax, tag
                    assert e[CONV['BN_PARTIAL']] <= PARTIAL_BYTES, tag
                    assert e[CONV['MASK']] == mask, tag


@pytest.mark.parametrize('B,H,W', SHAPES)
def test_weight_gradient_launch_extents_fit_the_allocations_the_abi_prescribes(B, H, W):
    L = lib()
    P = W + L.mmlf_grid_pad_w()
    alloc = L.mmlf_grid_alloc_positions(B, H, W)
    amax = L.mmlf_amax_entries(B, H, W) * 4
    if B * (H + 2) * P + 2 * P + 600 >= 2 ** 31:
        pytest.skip('engine.Geometry refuses this size')
    for cin, cout in LAYERS:
        cs_in, cs_g = cs_of(cin), cs_of(cout)
        ws = L.mmlf_wgrad_workspace_floats(cin, cout, B, H, W) * 4
        assert ws > 0
        for g_shift in (0, P + 1):
            e = (ctypes.c_int64 * 7)()
            assert L.mmlf_audit_wgrad_h2(cs_in, cin, cs_g, cout, g_shift, B, H, W, e) == 0, _lib.last_error()
            tag = f'{cin}->{cout} B={B} {H}x{W} g_shift={g_shift}'
            assert e[WG['IN']] <= alloc * cs_in * 4, tag
            assert e[WG['G']] <= alloc * cs_g * 4, tag
            assert e[WG['GW']] == cout * cin * 16 and e[WG['GB']] == cout * 4, tag
            assert 0 < e[WG['WORKSPACE']] <= ws, tag
            assert e[WG['IN_AMAX']] <= amax and e[WG['G_AMAX']] <= amax, tag


def test_the_engine_allocates_what_the_audit_assumes():
    """the constants this file holds the extents against are the ones mmlf_amd.engine uses"""
    import inspect
    from mmlf_amd import engine
    assert engine.BN_BLOCKS == 1024 and engine.LOSS_BLOCKS == 1024
    src = inspect.getsource(engine._Workspace.__init__)
    assert '2 * 512 * max(BN_BLOCKS, LOSS_BLOCKS) + 8' in src
    src = inspect.getsource(engine.Geometry.__init__)
    assert 'mmlf_grid_alloc_positions' in src and 'mmlf_amax_entries' in src


# ------------------------------------------------------------------ the loader refuses what it must not call
class _Stub:
    def __init__(self, abi, ablation, info=b'abi=7 git=stub MMLF_ABL_TERMS=2 ablation=1'):
        self._abi, self._abl, self._info = abi, ablation, info

    def mmlf_abi_version(self):
        return self._abi

    def mmlf_build_is_ablation(self):
        return self._abl

    def mmlf_build_info(self):
        return self._info


def test_loader_refuses_ablation_builds_unless_asked():
    with pytest.raises(RuntimeError, match='WRONG results'):
        _lib.validate(_Stub(_lib.ABI_VERSION, 1), 'variants/lib_x.so', environ={})
    with pytest.raises(RuntimeError, match='WRONG results'):
        _lib.validate(_Stub(_lib.ABI_VERSION, 1), 'variants/lib_x.so', environ={'MMLF_ALLOW_ABLATION': '0'})
    info = _lib.validate(_Stub(_lib.ABI_VERSION, 1), 'variants/lib_x.so', environ={'MMLF_ALLOW_ABLATION': '1'})
    assert 'ablation=1' in info
    assert _lib.validate(_Stub(_lib.ABI_VERSION, 0, b'abi=7 ablation=0'), 'x.so', environ={}) == 'abi=7 ablation=0'


def test_loader_refuses_other_abi_versions_and_anonymous_builds():
    with pytest.raises(RuntimeError, match='ABI version'):
        _lib.validate(_Stub(_lib.ABI_VERSION - 1, 0), 'old.so', environ={})

    class Anonymous:
        def mmlf_abi_version(self):
            return _lib.ABI_VERSION
    with pytest.raises(RuntimeError, match='mmlf_build_info'):
        _lib.validate(Anonymous(), 'anon.so', environ={})


def test_the_product_library_says_what_it_is():
    info = _lib.build_info()
    assert f'abi={_lib.ABI_VERSION} ' in info and 'ablation=0' in info and 'MMLF_ABL_TERMS=3' in info
    assert 'MMLF_BOUNDS_DEBUG=0' in info and 'MMLF_GRID_PAD_W=2' in info
    assert lib().mmlf_build_is_ablation() == 0
