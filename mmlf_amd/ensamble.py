"""Ensamble: drop-in for reference mmlf/model/ensamble.py:9-118 (the ESE method).

A 70-member shift ensemble with shared weights: for every disparity offset in
``np.arange(disp_min, disp_max, disp_step)`` the four view stacks are sheared by that offset
(``Shift``, reference mmlf/data/hci4d.py:894-990), the UPR model predicts (mean, logvar), and the
members are fused per pixel (arg-min logvar) plus a Laplace-mixture posterior.

On CUDA tensors: one HIP kernel shears all members at once (``mmlf_shift_views``), the members run
through the model as ONE batch when the model is in eval mode (eval-mode BatchNorm is a per-channel
affine map, so batching members is exact), and one kernel does the fusion (``mmlf_ensamble_reduce``).
On CPU tensors the same arithmetic runs in torch ops, member by member.
"""
import math
import os

import numpy as np
import torch
import torch.nn as nn

from . import _lib
from ._lib import call, ptr
from .feed_forward import laplacian


# MMLF_ESE_FUSED=0: the members go through mmlf_shift_views and the module's own forward (A/B of the fused member path)
FUSED_MEMBERS = os.environ.get('MMLF_ESE_FUSED', '1') != '0'


def shift_table(disps, views):
    """(shift0, shift1, 1-alpha, alpha) per (member, view), exactly as hci4d.py:934-938."""
    half = int(views / 2)
    tab_s = np.zeros((len(disps), views, 2), np.int32)
    tab_w = np.zeros((len(disps), views, 2), np.float32)
    for s, disp in enumerate(disps):
        for k in range(views):
            alpha, s0 = math.modf(float(disp) * (k - half))
            alpha = abs(alpha)
            s1 = s0 + math.copysign(1.0, s0)
            tab_s[s, k] = (int(s0), int(s1))
            tab_w[s, k] = (1.0 - alpha, alpha)
    return tab_s, tab_w


def _roll(x, s, dim):
    # cat([x[-s:], x[:-s]]) with Python slice clamping: |s| >= size leaves x unchanged
    return x if abs(s) >= x.shape[dim] else torch.roll(x, s, dim)


def shift_views_torch(stacks, disp):
    """hci4d.Shift on torch tensors (..., views, 3, H, W); returns new tensors (CPU path)."""
    h, v, i, d = [t.clone() for t in stacks]
    views = h.shape[-4]
    tab_s, tab_w = shift_table([disp], views)
    for k in range(views):
        (s0, s1), (w0, w1) = tab_s[0, k], tab_w[0, k]
        for t in (h, i, d):
            x = t[..., k, :, :, :]
            t[..., k, :, :, :] = _roll(x, int(s0), -1) * float(w0) + _roll(x, int(s1), -1) * float(w1)
    for k in range(views):
        (s0, s1), (w0, w1) = tab_s[0, k], tab_w[0, k]
        for t, sign in ((v, 1), (i, -1), (d, 1)):
            x = t[..., k, :, :, :]
            t[..., k, :, :, :] = _roll(x, sign * int(s0), -2) * float(w0) + _roll(x, sign * int(s1), -2) * float(w1)
    return h, v, i, d


class Ensamble(nn.Module):
    def __init__(self, model, val_disp_min, val_disp_max, val_disp_step, **kwarg):
        super().__init__()
        self.disp_min, self.disp_max, self.disp_step = val_disp_min, val_disp_max, val_disp_step
        assert self.disp_min < self.disp_max
        assert self.disp_step > 0.0
        self.model = model
        self.member_budget_bytes = 24e9     # per activation buffer when members are batched

    def members(self):
        return np.arange(self.disp_min, self.disp_max, self.disp_step)

    def forward(self, h_views, v_views, i_views=None, d_views=None):
        if i_views is None or d_views is None:
            # The reference's signature has the same defaults (ensamble.py:40), but its own loop cannot run without the
            # diagonal stacks: it hands Shift a 2-tuple (ensamble.py:63-64) and Shift reads data[2] and data[3]
            # unconditionally (hci4d.py:927-928) -- IndexError before the first member.  Same exception type here.
            raise IndexError('Ensamble: i_views and d_views are required (the reference\'s Shift indexes data[2] and '
                             'data[3] for a two-stack ensemble too, hci4d.py:927-928: list index out of range)')
        disps = self.members()
        S = len(disps)
        if h_views.is_cuda:
            means, logvars = self._members_hip(h_views, v_views, i_views, d_views, disps)
        else:
            means, logvars = [], []
            for sd in disps:
                out = self.model(*shift_views_torch((h_views, v_views, i_views, d_views), float(sd)))
                means.append(out['mean'] + float(sd))
                logvars.append(out['logvar'])
            means, logvars = torch.stack(means), torch.stack(logvars)
        b, hh, ww = means.shape[1:]
        grid = torch.from_numpy(np.linspace(self.disp_min, self.disp_max, S)).float().to(means.device)
        if means.is_cuda:
            means, logvars = means.contiguous(), logvars.contiguous()
            mean = torch.empty((b, hh, ww), dtype=torch.float32, device=means.device)
            logvar = torch.empty_like(mean)
            posterior = torch.empty((b, S, hh, ww), dtype=torch.float32, device=means.device)
            call('mmlf_ensamble_reduce', ptr(means), ptr(logvars), ptr(grid), ptr(mean), ptr(logvar),
                 ptr(posterior), S, b, hh, ww, _lib.stream_ptr())
        else:
            idx = torch.min(logvars, 0)[1].unsqueeze(0)
            mean, logvar = means.gather(0, idx)[0], logvars.gather(0, idx)[0]
            g = grid.view(1, -1, 1, 1).expand(b, S, hh, ww)
            posterior = torch.zeros((b, S, hh, ww))
            for k in range(S):
                posterior += laplacian(g, means[k], torch.exp(logvars[k]))
            posterior /= float(S)
        return {'mean': mean, 'logvar': logvar, 'means': means, 'logvars': logvars, 'posterior': posterior}

    def _members_hip(self, h, v, i, d, disps):
        b, views, c, hh, ww = h.shape
        S = len(disps)
        dev = h.device
        for name, t in (('v_views', v), ('i_views', i), ('d_views', d)):
            if t.device != dev or t.dtype != torch.float32 or tuple(t.shape) != tuple(h.shape):
                # raw pointers go to the shift kernel: a tensor of another device would fault inside it
                raise ValueError(f'Ensamble: {name} must be a float32 tensor of shape {tuple(h.shape)} on {dev}, '
                                 f'got {t.dtype} {tuple(t.shape)} on {t.device}')
        model = self.model
        inner = model.module if hasattr(model, 'module') else model
        tab_s, tab_w = shift_table(disps, views)
        tab_s, tab_w = torch.from_numpy(tab_s).to(dev), torch.from_numpy(tab_w).to(dev)
        per_member = (hh + 2) * (ww + 2) * 4 * max(getattr(inner, 'chs', 70), 8) * 4
        chunk = S if not inner.training else 1
        chunk = int(max(1, min(chunk, self.member_budget_bytes // per_member)))
        offs = torch.from_numpy(disps.astype(np.float32)).to(dev)
        means = torch.empty((S, b, hh, ww), dtype=torch.float32, device=dev)
        logvars = torch.empty_like(means)
        # Fused member path (round 6): the shear writes the trunk's own input layout (mmlf_shift_pack: no (n, views, 3, H, W)
        # intermediate, no pack pass) and the members' UPR `posterior` -- 108 planes per member that ensamble.py:66-76 never
        # reads -- is not formed.  Same bits as the path below (tests/test_ensamble.py).  For the uncertainty model itself, in
        # evaluation, outside autograd; anything else (a DataParallel wrapper, another head) takes the module's own forward.
        fused = (FUSED_MEMBERS and model is inner and getattr(inner, '_native_ok', False) and getattr(inner, 'uncert', False)
                 and not inner.training and not torch.is_grad_enabled() and views == inner.views and c == 3)
        for bi in range(b):
            src = [t[bi].contiguous() for t in (h, v, i, d)]
            for s0 in range(0, S, chunk):
                n = min(chunk, S - s0)
                if fused:
                    from . import engine
                    _lib.load()
                    with torch.cuda.device(dev):
                        geo = engine.Geometry(n, hh, ww)
                        cs = engine.cs_of(views * c)
                        xs = geo.bufs([cs] * 4, dev)
                        for kind, (t, x) in enumerate(zip(src, xs)):
                            call('mmlf_shift_pack', ptr(t), kind, ptr(x), cs, ptr(tab_s[s0:s0 + n]), ptr(tab_w[s0:s0 + n]),
                                 n, views, hh, ww, ptr(x.absmax), _lib.stream_ptr())
                        out, _ = inner._trunk.forward(inner._tensor_dict(), None, False, False, packed=(geo, xs))
                        del xs
                    means[s0:s0 + n, bi] = out[:, 0] + offs[s0:s0 + n].view(-1, 1, 1)
                    logvars[s0:s0 + n, bi] = out[:, 1]
                    continue
                outs = [torch.empty((n, views, c, hh, ww), dtype=torch.float32, device=dev) for _ in range(4)]
                call('mmlf_shift_views', *[ptr(t) for t in src], *[ptr(t) for t in outs],
                     ptr(tab_s[s0:s0 + n]), ptr(tab_w[s0:s0 + n]), n, views, hh, ww, _lib.stream_ptr())
                out = model(*outs)
                means[s0:s0 + n, bi] = out['mean'] + offs[s0:s0 + n].view(-1, 1, 1)
                logvars[s0:s0 + n, bi] = out['logvar']
        return means, logvars
