"""On-device training patch pipeline.

The reference produces a training sample on the CPU: ``HCI4D.__getitem__`` deep-copies a cached
512x512 scene and runs the transform chain of mmlf/train/cli.py:72-94 over the whole frame
(mmlf/data/hci4d.py:289-291) -- RandomDownSampling, RandomShift(1.0), RandomCrop(ps+16),
CenterCrop(ps), RandomRotate, RedistColor, Brightness, Contrast -- in 4 DataLoader workers.  At
several hundred patches/s per GPU that is the bottleneck.  Here the scenes stay in HBM and a batch is ONE
fused gather (mmlf_patch_gather) plus the Contrast pass (mmlf_patch_contrast): only the ps x ps output
pixels are computed.

Random parameters are drawn on the host from Python's ``random`` module in exactly the reference's call
order, so with the same seed a sample is the reference's sample (float tolerance: the Contrast mean is
summed in double here, pairwise float32 in numpy).
"""
import random as _random

import numpy as np
import torch

from . import _lib
from ._lib import call, ptr
from .ensamble import shift_table

FLAG_SHIFT, FLAG_COLOR, FLAG_BRIGHT = 1, 2, 4


def rotation_sources(views):
    """rot_src[r][stack][view] = source stack*views + source view after r x Rotate90
    (hci4d.py:1063-1069: new H = old V, new V = old H with the view order flipped; I/D likewise)."""
    lab = [[(s, n) for n in range(views)] for s in range(4)]
    tabs = []
    for _ in range(4):
        tabs.append([[s * views + n for (s, n) in row] for row in lab])
        lab = [lab[1], lab[0][::-1], lab[3], lab[2][::-1]]
    return np.asarray(tabs, dtype=np.int32)


def draw_sample(frame_hw, ps, max_downscale=4, augment=True, rng=None):
    """The random draws of one sample in the reference's order (train/cli.py:79-88 and the transforms'
    __call__ bodies).  Returns a dict of plain numbers."""
    rng = rng or _random
    H, W = frame_hw
    p = dict(f=1, disp=0.0, rot=0, mat=np.eye(3), bright=1.0, contrast=1.0, augment=bool(augment))
    if augment:
        p['f'] = rng.randint(1, max_downscale)                   # RandomDownSampling, hci4d.py:526
        p['disp'] = rng.uniform(-1.0, 1.0)                       # RandomShift(1.0), hci4d.py:1024
    big = ps + 2 * 4 * 2
    h, w = -(-H // p['f']), -(-W // p['f'])                      # frame[..., ::f, ::f]
    if not (h > big and w > big):
        raise ValueError(f'frame {h}x{w} after downsampling by {p["f"]} is not larger than the {big}-px crop '
                         '(hci4d.py:656-657)')
    y = rng.randint(0, h - big)                                  # RandomCrop, hci4d.py:659-660
    x = rng.randint(0, w - big)
    off = int((big - ps) / 2)                                    # CenterCrop, hci4d.py:617-618
    p['y0'], p['x0'] = y + off, x + off
    if augment:
        p['rot'] = rng.randint(0, 3)                             # RandomRotate, hci4d.py:1082
        mat = np.zeros((3, 3))                                   # RedistColor, hci4d.py:687-697
        mat[0, 0] = rng.uniform(0.0, 1.0)
        mat[0, 1] = rng.uniform(0.0, 1.0 - mat[0, 0])
        mat[1, 0] = rng.uniform(0.0, 1.0 - mat[0, 0])
        mat[1, 1] = rng.uniform(0.0, 1.0 - max(mat[0, 1], mat[1, 0]))
        mat[0, 2] = 1.0 - mat[0, 0] - mat[0, 1]
        mat[1, 2] = 1.0 - mat[1, 0] - mat[1, 1]
        mat[2, 0] = 1.0 - mat[0, 0] - mat[1, 0]
        mat[2, 1] = 1.0 - mat[0, 1] - mat[1, 1]
        mat[2, 2] = mat[0, 0] + mat[0, 1] + mat[1, 0] + mat[1, 1] - 1.0
        p['mat'] = mat
        p['bright'] = rng.uniform(-0.9, 0.9) + 1.0               # Brightness(), hci4d.py:773
        p['contrast'] = rng.uniform(-0.9, 0.9) + 1.0             # Contrast(), hci4d.py:740
    return p


class PatchPipeline:
    """Scenes cached on the GPU + the fused transform chain.

    scenes: list of tuples (h, v, i, d, center, gt, mpi, mask, index) as ``HCI4D.load_scene`` returns
    them (hci4d.py:150-254), all of one frame size.  ``train_shift`` is the CLI's fixed pre-shift
    (train/cli.py:93-94), applied once to the cached scenes.
    """

    def __init__(self, scenes, ps, max_downscale=4, augment=True, train_shift=0.0, device='cuda'):
        if not torch.cuda.is_available():
            raise RuntimeError('PatchPipeline needs the HIP library and an MI355X; use the reference '
                               'DataLoader path on CPU')
        _lib.load()
        self.ps, self.max_downscale, self.augment = int(ps), int(max_downscale), bool(augment)
        dev = torch.device(device)
        f32 = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(dev)   # noqa: E731
        self.stacks = torch.stack([torch.stack([f32(s[k]) for k in range(4)]) for s in scenes])  # (S,4,V,3,H,W)
        self.center = torch.stack([f32(s[4]) for s in scenes])
        self.gt = torch.stack([f32(s[5]) for s in scenes])
        self.mpi = torch.stack([f32(s[6]) for s in scenes])
        self.mask = torch.stack([torch.from_numpy(np.ascontiguousarray(s[7]).astype(np.int32)).to(dev)
                                 for s in scenes])
        self.index = [np.atleast_1d(s[8]) for s in scenes]
        self.S, _, self.V, _, self.Hf, self.Wf = self.stacks.shape
        self.P = self.mpi.shape[1]
        self.rot_src = torch.from_numpy(rotation_sources(self.V)).to(dev)
        if train_shift != 0.0:
            self._preshift(float(train_shift))

    def _preshift(self, disp):
        """hci4d.Shift(train_shift) on every cached scene (stacks, gt, mpi[:, 4]; hci4d.py:907-990)."""
        tab_s, tab_w = shift_table([disp], self.V)
        dev = self.stacks.device
        ts, tw = torch.from_numpy(tab_s).to(dev), torch.from_numpy(tab_w).to(dev)
        out = torch.empty_like(self.stacks)
        for s in range(self.S):
            src, dst = self.stacks[s], out[s]
            call('mmlf_shift_views', ptr(src[0]), ptr(src[1]), ptr(src[2]), ptr(src[3]), ptr(dst[0]), ptr(dst[1]),
                 ptr(dst[2]), ptr(dst[3]), ptr(ts), ptr(tw), 1, self.V, self.Hf, self.Wf, _lib.stream_ptr())
        self.stacks = out
        self.gt = self.gt - np.float32(disp)
        self.mpi[:, :, 4] -= np.float32(disp)

    def sample(self, scene_indices, rng=None):
        """One batch: (h, v, i, d, center, gt, mpi, mask, index), batch-first like the collated output of
        the reference DataLoader (stacks (B,V,3,ps,ps), ...).  Draws come from `rng` (default: the
        ``random`` module), one sample after the other, in the reference's order."""
        B, ps, V, dev = len(scene_indices), self.ps, self.V, self.stacks.device
        ip = np.zeros((B, 8), np.int32)
        fp = np.zeros((B, 4), np.float32)
        mat = np.zeros((B, 9), np.float64)
        alpha = np.zeros((B, 2), np.float32)
        disps = []
        for b, sc in enumerate(scene_indices):
            p = draw_sample((self.Hf, self.Wf), ps, self.max_downscale, self.augment, rng)
            flags = (FLAG_SHIFT | FLAG_COLOR | FLAG_BRIGHT) if self.augment else 0
            ip[b, :6] = (int(sc) % self.S, p['f'], p['y0'], p['x0'], p['rot'], flags)
            fp[b, :3] = (float(p['f']), p['disp'], p['bright'])
            mat[b] = p['mat'].reshape(-1)
            alpha[b] = (p['contrast'], 1.0 - p['contrast'])
            disps.append(p['disp'])
        tab_s, tab_w = shift_table(disps, V)
        up = lambda a: torch.from_numpy(a).to(dev)   # noqa: E731
        ip_d, fp_d, mat_d, al_d, ts_d, tw_d = up(ip), up(fp), up(mat), up(alpha), up(tab_s), up(tab_w)
        o_st = torch.empty((4, B, V, 3, ps, ps), dtype=torch.float32, device=dev)
        o_c = torch.empty((B, 3, ps, ps), dtype=torch.float32, device=dev)
        o_gt = torch.empty((B, ps, ps), dtype=torch.float32, device=dev)
        o_mpi = torch.empty((B, self.P, 5, ps, ps), dtype=torch.float32, device=dev)
        o_mask = torch.empty((B, ps, ps), dtype=torch.int32, device=dev)
        msum = torch.zeros(B, dtype=torch.float64, device=dev)
        call('mmlf_patch_gather', ptr(self.stacks), ptr(self.center), ptr(self.gt), ptr(self.mpi), ptr(self.mask),
             self.S, V, self.P, self.Hf, self.Wf, ptr(ip_d), ptr(fp_d), ptr(ts_d), ptr(tw_d), ptr(mat_d),
             ptr(self.rot_src), ptr(o_st), ptr(o_c), ptr(o_gt), ptr(o_mpi), ptr(o_mask), ptr(msum), B, ps,
             _lib.stream_ptr())
        if self.augment:
            call('mmlf_patch_contrast', ptr(o_st), ptr(o_c), ptr(msum), ptr(al_d), B, V, ps, _lib.stream_ptr())
        index = torch.from_numpy(np.stack([self.index[int(sc) % self.S] for sc in scene_indices]))
        return o_st[0], o_st[1], o_st[2], o_st[3], o_c, o_gt, o_mpi, o_mask, index
