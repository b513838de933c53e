"""SURVEY section 8 f3, second half: PFM / NPY result writers (reference mmlf/utils/pfm.py:6-93,
mmlf/data/hci4d.py:295-413) against bytes written by the reference's own pfm.save (tests/golden/g10_pfm.npz)."""
import os

import numpy as np
import pytest
import torch

from conftest import load_golden
from mmlf_amd import pfm, results

CASES = ['grey', 'grey1', 'colour', 'flipped', 'scaled']


@pytest.mark.parametrize('case', CASES)
def test_pfm_save_is_byte_exact_and_load_round_trips(tmp_path, case):
    g = load_golden('g10_pfm.npz')
    arr = g[f'{case}/array']
    if case == 'flipped':                      # the reference was handed a negatively strided view: do the same
        arr = np.flip(np.flip(arr, 0).copy(), 0)
        assert arr.strides[0] < 0
    fn = str(tmp_path / f'{case}.pfm')
    pfm.save(fn, arr, **({'scale': 2.5} if case == 'scaled' else {}))
    assert open(fn, 'rb').read() == g[f'{case}/bytes'].tobytes()
    back = pfm.load(fn)
    np.testing.assert_array_equal(back, g[f'{case}/loaded'])
    assert back.dtype == np.float32 and back.flags.writeable


def test_pfm_load_big_endian_and_errors(tmp_path):
    g = load_golden('g10_pfm.npz')
    fn = str(tmp_path / 'be.pfm')
    open(fn, 'wb').write(g['bigendian/bytes'].tobytes())
    np.testing.assert_array_equal(pfm.load(fn), g['bigendian/loaded'])
    with pytest.raises(Exception, match='float32'):
        pfm.save(fn, np.zeros((2, 2), dtype=np.float64))
    with pytest.raises(Exception, match='dimensions'):
        pfm.save(fn, np.zeros((2, 2, 2), dtype=np.float32))
    open(fn, 'wb').write(b'P6\n1 1\n255\n')
    with pytest.raises(Exception, match='Not a PFM'):
        pfm.load(fn)


def _check_layout(root, names, gt, result, uncert, gmm, nll, posterior, runtime):
    for b, scene in enumerate(names):
        sd = os.path.join(root, 'scenes', scene)
        # PFMs hold the vertically flipped maps (hci4d.py:358-366): loading gives the flipped array back
        np.testing.assert_array_equal(pfm.load(os.path.join(sd, 'gt.pfm')), gt[b][::-1])
        np.testing.assert_array_equal(pfm.load(os.path.join(sd, 'result.pfm')), result[b][::-1])
        np.testing.assert_array_equal(pfm.load(os.path.join(sd, 'uncert.pfm')), uncert[b][::-1])
        assert (open(os.path.join(sd, 'result.pfm'), 'rb').read()
                == open(os.path.join(root, 'ours', 'disp_maps', f'{scene}.pfm'), 'rb').read())
        np.testing.assert_array_equal(np.load(os.path.join(sd, 'gmm.npy')), gmm[:, :, b])
        np.testing.assert_array_equal(np.load(os.path.join(sd, 'nll.npy')), nll[b])
        np.testing.assert_array_equal(np.load(os.path.join(sd, 'posterior.npy')), posterior[b])
        assert open(os.path.join(root, 'ours', 'runtimes', f'{scene}.txt')).read() == str(runtime / len(names))
        for png in ('gt.png', 'diff.png', 'result.png', 'uncert.png', 'center.png', 'view_h_0.png', 'view_d_2.png'):
            assert os.path.getsize(os.path.join(sd, png)) > 0
        # pixel values (the PNG container is the encoder's business): reference dl.py:93-106 + skimage.img_as_ubyte
        lo, hi = gt[b].min(), gt[b].max()
        for png, src in (('gt.png', gt[b]), ('diff.png', np.abs(gt[b] - result[b])), ('uncert.png', uncert[b]),
                         ('result.png', np.clip((result[b] - lo) / (hi - lo), 0.0, 1.0))):
            np.testing.assert_array_equal(_png_pixels(os.path.join(sd, png)), _ubyte_like_the_reference(src), err_msg=png)


def _png_pixels(fname):
    from PIL import Image
    return np.asarray(Image.open(fname))


def _ubyte_like_the_reference(arr):
    """dl.py:93-103 (min-max normalise when the array leaves [0, 1], CHW -> HWC) followed by skimage.img_as_ubyte on a
    float image: x * 255 in the image's own float type, rounded to nearest even, clipped (skimage/util/dtype.py _convert)"""
    a_min, a_max = np.min(arr), np.max(arr)
    if a_min < 0.0 or a_max > 1.0:
        arr = (arr - a_min) / (a_max - a_min)
    if arr.ndim == 3:
        arr = np.transpose(arr, (1, 2, 0))
    out = np.multiply(arr, 255, dtype=arr.dtype)
    np.rint(out, out=out)
    np.clip(out, 0, 255, out=out)
    return out.astype(np.uint8)


def test_save_img_pixels_rgb_and_float32_ties(tmp_path):
    """colour images go CHW -> HWC; float32 maps are scaled in float32 (a float64 detour moves pixels that sit on a tie)"""
    rs = np.random.RandomState(11)
    rgb = rs.rand(3, 9, 7).astype(np.float32)
    fn = str(tmp_path / 'rgb.png')
    results.save_img(fn, torch.from_numpy(rgb))
    np.testing.assert_array_equal(_png_pixels(fn), _ubyte_like_the_reference(rgb))
    # values whose product with 255 rounds differently in float32 and float64
    k = np.arange(0, 255, dtype=np.float64) + 0.5
    x = (k / 255.0).astype(np.float32).reshape(15, 17)
    x[0, 0] = 0.0                                   # keeps the array inside [0, 1]: no normalisation
    fn = str(tmp_path / 'ties.png')
    results.save_img(fn, x)
    want = _ubyte_like_the_reference(x)
    np.testing.assert_array_equal(_png_pixels(fn), want)
    assert (want != np.clip(np.rint(x.astype(np.float64) * 255.0), 0, 255).astype(np.uint8)).any()   # the case is not vacuous


@pytest.mark.parametrize('dev', ['cpu', pytest.param('cuda', marks=pytest.mark.gpu)])
def test_save_batch_layout(tmp_path, dev):
    rs = np.random.RandomState(5)
    names = ['boxes', 'cotton', 'dino']
    B, K, H, W = 2, 4, 12, 10
    gt = rs.uniform(-2, 2, (B, H, W)).astype(np.float32)
    result = (gt + 0.1 * rs.randn(B, H, W)).astype(np.float32)
    uncert = rs.rand(B, H, W).astype(np.float32)
    gmm = rs.randn(2, K, B, H, W).astype(np.float32)
    nll = rs.randn(B, K, H, W).astype(np.float32)
    posterior = rs.rand(B, K, H, W).astype(np.float32)
    center = rs.rand(B, 3, H, W).astype(np.float32)
    views = [rs.rand(B, 3, 3, H, W).astype(np.float32) for _ in range(4)]
    index = np.array([[2], [0]])
    t = (lambda a: torch.from_numpy(a).to(dev))
    results.save_batch(str(tmp_path), names, torch.from_numpy(index), gt=t(gt), result=t(result), uncert=t(uncert),
                       runtime=0.5, gmm=t(gmm), nll=t(nll), posterior=t(posterior), center=t(center),
                       views=[t(v) for v in views])
    _check_layout(str(tmp_path), ['dino', 'boxes'], gt, result, uncert, gmm, nll, posterior, 0.5)
    # the PFM bytes are what the reference's writer produces for the flipped map (header + rows, little endian)
    raw = open(os.path.join(str(tmp_path), 'scenes', 'dino', 'result.pfm'), 'rb').read()
    assert raw == b'Pf\n%d %d\n%f\n' % (W, H, -1.0) + np.ascontiguousarray(result[0][::-1]).tobytes()


@pytest.mark.gpu
def test_save_batch_takes_model_outputs_on_the_gpu(tmp_path):
    """validate/cli.py:289-311 hands the output dict's maps to save_batch: cuda tensors go straight in"""
    from conftest import TINY_KW
    from mmlf_amd import synth
    from mmlf_amd.feed_forward import FeedForward
    kw = dict(TINY_KW, model_uncert=True)
    model = FeedForward(**kw)
    state = synth.synth_state(synth.param_spec(**kw), 3)
    model.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in state.items()})
    model.cuda().eval()
    stacks, gt, _ = synth.synth_inputs(2, 16, seed=4)
    with torch.no_grad():
        out = model(*[torch.from_numpy(s).cuda() for s in stacks])
    results.save_batch(str(tmp_path), ['a', 'b'], np.array([[0], [1]]), gt=gt, result=out['mean'], uncert=out['logvar'],
                       posterior=out['posterior'], images=False)
    np.testing.assert_array_equal(pfm.load(str(tmp_path / 'scenes' / 'b' / 'result.pfm')),
                                  out['mean'][1].cpu().numpy()[::-1])
    np.testing.assert_array_equal(np.load(str(tmp_path / 'scenes' / 'a' / 'posterior.npy')),
                                  out['posterior'][0].cpu().numpy())
