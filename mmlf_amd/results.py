"""Result writers: the directory layout of reference mmlf/data/hci4d.py:295-413 (``HCI4D.save_batch``), which the
reference's downstream analysis scripts and the 4D light field benchmark evaluation read:

    <path>/scenes/<scene>/gt.pfm, result.pfm, uncert.pfm          vertically flipped float32 PFMs (pfm.py)
    <path>/scenes/<scene>/gmm.npy, nll.npy, posterior.npy         np.save of gmm[:, :, b] / nll[b] / posterior[b]
    <path>/scenes/<scene>/center.png, gt.png, diff.png, result.png, uncert.png, view_{h,v,i,d}_<j>.png
    <path>/ours/disp_maps/<scene>.pfm                             the result again, benchmark layout
    <path>/ours/runtimes/<scene>.txt                              str(runtime / batch size)

The reference method lives on its dataset object and re-loads the scene from disk (``self.__getitem__(i)``); dataset
disk IO is outside this build, so the scene-side arrays (gt, centre view, view stacks) are
arguments here.  Every array may be a numpy array or a torch tensor on any device (``Ensamble`` / ``FeedForward``
outputs go straight in).  PNGs follow reference dl.py:77-106 (min-max normalise when outside [0, 1], 8 bit) and
skimage.img_as_ubyte's float rule (x * 255 in the image's own float type, nearest even, clipped); they are written with
PIL -- the reference's skimage is not installed in this image, so the PNG bytes are not pinned by a fixture; the decoded
PIXELS are checked against a restatement of those two rules (tests/test_results_io.py).
"""
import os

import numpy as np

from . import pfm


def _np(x):
    if x is None:
        return None
    if hasattr(x, 'detach'):
        x = x.detach().cpu().numpy()
    return np.asarray(x)


def save_img(fname, arr):
    """reference dl.py:77-106: (3, h, w) rgb or (h, w) grey, normalised to [0, 1] if it leaves that range"""
    from PIL import Image
    arr = _np(arr)
    if not np.issubdtype(arr.dtype, np.floating):
        arr = arr.astype(np.float64)
    # the arithmetic stays in the array's own float type, as in the reference (float32 maps normalise and scale in
    # float32: a pixel that sits on a rounding tie comes out the same)
    a_min, a_max = np.min(arr), np.max(arr)
    if a_min < 0.0 or a_max > 1.0:
        arr = (arr - a_min) / (a_max - a_min)
    if arr.ndim == 3:
        arr = np.transpose(arr, (1, 2, 0))
    # skimage.img_as_ubyte on [0, 1] floats: multiply by 255 in the input's float type, round to nearest even, clip
    img = np.clip(np.rint(np.multiply(arr, 255, dtype=arr.dtype)), 0, 255).astype(np.uint8)
    Image.fromarray(img).save(fname)


def save_views(scene_dir, h_views, v_views, i_views=None, d_views=None):
    """reference utils/lf.py:6-57"""
    os.makedirs(scene_dir, exist_ok=True)
    for tag, views in (('h', h_views), ('v', v_views), ('i', i_views), ('d', d_views)):
        views = _np(views)
        if views is None:
            continue
        if views.ndim == 5:
            views = views[0]
        for j in range(views.shape[0]):
            save_img(os.path.join(scene_dir, f'view_{tag}_{j}.png'), views[j])


def save_batch(path, scene_names, index, gt=None, result=None, uncert=None, runtime=None, gmm=None, nll=None,
               posterior=None, center=None, views=None, images=True):
    """Counterpart of ``HCI4D.save_batch(path, index, result, uncert, runtime, gmm, nll, posterior)``.

    scene_names: list of scene names (``dataset.scenes_names``); index: (b, 1) array of indices into it, as the
    data loader delivers it; gt: (b, h, w) ground truth of the batch (the reference re-reads it from the dataset);
    result / uncert: (b, h, w); gmm: (2, K, b, h, w); nll, posterior: (b, K, h, w); runtime: seconds for the batch;
    center: (b, 3, h, w) and views: four (b, n, 3, h, w) stacks for the PNG side (optional; images=False skips PNGs).
    """
    gt, result, uncert, gmm, nll, posterior, center = map(_np, (gt, result, uncert, gmm, nll, posterior, center))
    scenes, ours = os.path.join(path, 'scenes'), os.path.join(path, 'ours')
    disp_maps, runtimes = os.path.join(ours, 'disp_maps'), os.path.join(ours, 'runtimes')
    for d in (scenes, ours, disp_maps, runtimes):
        os.makedirs(d, exist_ok=True)
    idx = _np(index)
    idx = idx.squeeze(1) if idx.ndim == 2 else idx.reshape(-1)
    for arr_i, i in enumerate(idx.tolist()):
        scene = scene_names[int(i)]
        scene_dir = os.path.join(scenes, scene)
        os.makedirs(scene_dir, exist_ok=True)
        if images and views is not None:
            save_views(scene_dir, *[None if v is None else _np(v)[arr_i] for v in views])
        if images and center is not None:
            save_img(os.path.join(scene_dir, 'center.png'), center[arr_i])
        if gt is not None:
            g = gt[arr_i].astype(np.float32, copy=False)
            if images:
                save_img(os.path.join(scene_dir, 'gt.png'), g)
                if result is not None:
                    save_img(os.path.join(scene_dir, 'diff.png'), np.abs(g - result[arr_i]))
            pfm.save(os.path.join(scene_dir, 'gt.pfm'), np.flip(g.copy(), 0))
        if result is not None:
            res_out = np.flip(result[arr_i].astype(np.float32).copy(), 0)
            pfm.save(os.path.join(scene_dir, 'result.pfm'), res_out)
            pfm.save(os.path.join(disp_maps, f'{scene}.pfm'), res_out)
            if images and gt is not None:       # normalised to the ground-truth range and clipped (hci4d.py:368-378)
                lo, hi = np.min(gt[arr_i]), np.max(gt[arr_i])
                save_img(os.path.join(scene_dir, 'result.png'), np.clip((result[arr_i] - lo) / (hi - lo), 0.0, 1.0))
        if uncert is not None:
            pfm.save(os.path.join(scene_dir, 'uncert.pfm'), np.flip(uncert[arr_i].astype(np.float32).copy(), 0))
            if images:
                save_img(os.path.join(scene_dir, 'uncert.png'), uncert[arr_i])
        if gmm is not None:
            np.save(os.path.join(scene_dir, 'gmm.npy'), gmm[:, :, arr_i])
        if nll is not None:
            np.save(os.path.join(scene_dir, 'nll.npy'), nll[arr_i, ...])
        if posterior is not None:
            np.save(os.path.join(scene_dir, 'posterior.npy'), posterior[arr_i, ...])
        if runtime is not None:
            with open(os.path.join(runtimes, f'{scene}.txt'), 'w') as f:
                f.write(str(runtime / float(idx.shape[0])))
