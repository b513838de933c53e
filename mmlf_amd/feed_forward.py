"""FeedForward: drop-in for reference mmlf/model/feed_forward.py:15-305.

Same constructor keywords, same ``forward(h_views, v_views, i_views, d_views)`` signature, same
five-key output dict and the same ``state_dict`` key set, so reference
checkpoints load unchanged and the reference's train / validate drivers can use it as is.

On CUDA (= HIP on ROCm) tensors with the default flags (k=2, BatchNorm, four streams) the whole
trunk runs in hand-written gfx950 kernels through the C ABI (engine.Trunk); there is no fallback on
that path: a missing libmmlf_hip.so raises.  CPU tensors, and the non-default flags the README
recipes never use (model_cross, odd ksize, model_no_batchnorm, model_unet), run the module tree with stock
torch ops ("plumbing path": BASELINE.json configs[0], CPU tests, gloo rehearsal).
"""
import numpy as np
import torch
import torch.nn as nn

from . import _lib
from ._lib import call, ptr
from .engine import Trunk


def laplacian(x, mu, b):
    """Laplace density, reference feed_forward.py:9-12 (kept for Ensamble and callers)."""
    mu = mu.unsqueeze(1)
    b = b.unsqueeze(1)
    return 1.0 / (2.0 * b) * torch.exp(-torch.abs(x - mu) / b)


def _conv_block(cin, cout, ksize, pad1, pad2, bn, momentum):
    layers = [nn.Conv2d(cin, cout, ksize, padding=pad1), nn.ReLU(),
              nn.Conv2d(cout, cout, ksize, padding=pad2)]
    if bn is not None:
        if bn:
            layers.append(nn.BatchNorm2d(cout, momentum=momentum))
        layers.append(nn.ReLU())
    return nn.Sequential(*layers)


class _UNetStage(nn.Module):
    """two 3x3 conv + ReLU + BatchNorm pairs; parameter names `block.{0,2,3,5}` as in reference unet.py:81-101"""

    def __init__(self, cin, cout):
        super().__init__()
        self.block = nn.Sequential(nn.Conv2d(cin, cout, 3, padding=1), nn.ReLU(), nn.BatchNorm2d(cout),
                                   nn.Conv2d(cout, cout, 3, padding=1), nn.ReLU(), nn.BatchNorm2d(cout))

    def forward(self, x):
        return self.block(x)


class _UNetUp(nn.Module):
    """transposed-conv upsampling, centre-cropped skip connection, conv stage (reference unet.py:104-135)"""

    def __init__(self, cin, cout):
        super().__init__()
        self.up = nn.ConvTranspose2d(cin, cout, kernel_size=2, stride=2)
        self.conv_block = _UNetStage(cin, cout)

    def forward(self, x, skip):
        x = self.up(x)
        h, w = x.shape[2:]
        y0, x0 = (skip.shape[2] - h) // 2, (skip.shape[3] - w) // 2
        return self.conv_block(torch.cat([x, skip[:, :, y0:y0 + h, x0:x0 + w]], 1))


class _UNetTail(nn.Module):
    """--model_unet: the merge network as a depth-5 U-Net (reference feed_forward.py:189-204, unet.py:8-78 with
    padding=True, batch_norm=True, wf=6, 'upconv').  Stock torch ops only: no README recipe uses the flag, so it is
    outside the accelerated path ("stays accepted and takes a stock-torch fallback"); the
    state_dict keys (`down_path.i.block.*`, `up_path.i.{up,conv_block.block}.*`, `last.*`) match the reference's."""

    def __init__(self, cin, cout, depth=5, wf=6):
        super().__init__()
        widths = [2 ** (wf + i) for i in range(depth)]
        self.down_path = nn.ModuleList(_UNetStage(a, b) for a, b in zip([cin] + widths[:-1], widths))
        self.up_path = nn.ModuleList(_UNetUp(widths[i + 1], widths[i]) for i in reversed(range(depth - 1)))
        self.last = nn.Conv2d(widths[0], cout, kernel_size=1)

    def forward(self, x):
        skips = []
        for stage in self.down_path[:-1]:
            x = stage(x)
            skips.append(x)
            x = nn.functional.max_pool2d(x, 2)
        x = self.down_path[-1](x)
        for up in self.up_path:
            x = up(x, skips.pop())
        return self.last(x)


class _TrunkFn(torch.autograd.Function):
    """One autograd node for in_net_hv x2, in_net_id x2, concat and out_net."""

    @staticmethod
    def forward(ctx, module, train, save, h, v, i, d, *params):
        p = module._tensor_dict()                       # buffers; parameters as passed (a replica's per-device copies)
        p.update(zip(module._param_names, (t.detach() for t in params)))
        with torch.no_grad():
            out, tape = module._trunk.forward(p, [h, v, i, d], train, save)
        ctx.module, ctx.tape, ctx.p = module, tape, (p if save else None)
        return out

    @staticmethod
    def backward(ctx, gout):
        module, tape, p = ctx.module, ctx.tape, ctx.p
        if tape is None:
            raise RuntimeError('FeedForward: backward through a forward that saved nothing')
        ctx.tape = ctx.p = None
        names = module._param_names
        sizes = [p[n].numel() for n in names]
        flat = torch.zeros(sum(sizes), dtype=torch.float32, device=gout.device)
        grads, o = {}, 0
        for n, sz in zip(names, sizes):
            grads[n] = flat[o:o + sz].view_as(p[n])
            o += sz
        with torch.no_grad(), torch.cuda.device(gout.device):
            module._trunk.backward(p, tape, gout, grads)
        return (None, None, None, None, None, None, None) + tuple(grads[n] for n in names)


class _HeadUprFn(torch.autograd.Function):
    """posterior of the UPR head (reference feed_forward.py:292-302) as a differentiable function of the network output:
    forward = mmlf_head_upr, backward = mmlf_head_upr_bwd (round 5: the native path used to compute it from a detached
    tensor, so a loss on output['posterior'] trained nothing on cuda while it did on the stock-torch branch)."""

    @staticmethod
    def forward(ctx, output, grid, steps):
        o = output.detach().contiguous()
        b, _, hh, ww = o.shape
        posterior = torch.empty((b, steps, hh, ww), dtype=torch.float32, device=o.device)
        call('mmlf_head_upr', ptr(o), ptr(grid), ptr(posterior), steps, b, hh, ww, _lib.stream_ptr())
        ctx.save_for_backward(o, grid)
        ctx.steps = steps
        return posterior

    @staticmethod
    def backward(ctx, gpost):
        o, grid = ctx.saved_tensors
        b, _, hh, ww = o.shape
        gout = torch.empty_like(o)
        with torch.cuda.device(o.device):
            call('mmlf_head_upr_bwd', ptr(o), ptr(grid), ptr(gpost.contiguous()), ptr(gout), ctx.steps, b, hh, ww,
                 _lib.stream_ptr())
        return gout, None, None


class _HeadDppFn(torch.autograd.Function):
    """one_hot, posterior, mean, logvar of the DPP head (reference feed_forward.py:276-290): posterior and logvar are
    differentiable in the scores (mmlf_head_dpp_bwd); one_hot and the arg-max mean are constants of the graph, as in the
    reference (a comparison has no gradient).  Where all posterior mass sits on the arg-max bin the variance is 0, logvar is
    -inf and its gradient is inf / NaN -- in the reference's autograd graph (log at 0) as here; no floor is applied."""

    @staticmethod
    def forward(ctx, scores, grid_torch, grid_np, steps):
        sc = scores.detach().contiguous()
        b, _, hh, ww = sc.shape
        one_hot, posterior = torch.empty_like(sc), torch.empty_like(sc)
        mean = torch.empty((b, hh, ww), dtype=torch.float32, device=sc.device)
        logvar = torch.empty_like(mean)
        call('mmlf_head_dpp', ptr(sc), ptr(grid_torch), ptr(grid_np), ptr(one_hot), ptr(posterior), ptr(mean), ptr(logvar),
             steps, b, hh, ww, _lib.stream_ptr())
        ctx.save_for_backward(sc, grid_np, mean)
        ctx.steps = steps
        ctx.mark_non_differentiable(one_hot, mean)
        # an output nobody differentiates arrives as None in backward (not as a materialised zero tensor): the NULL-pointer
        # paths of mmlf_head_dpp_bwd are real, and a graph that uses neither posterior nor logvar launches nothing
        ctx.set_materialize_grads(False)
        return one_hot, posterior, mean, logvar

    @staticmethod
    def backward(ctx, g_one_hot, g_post, g_mean, g_lv):
        sc, grid_np, mean = ctx.saved_tensors
        if g_post is None and g_lv is None:
            return None, None, None, None
        b, _, hh, ww = sc.shape
        gsc = torch.empty_like(sc)
        with torch.cuda.device(sc.device):
            call('mmlf_head_dpp_bwd', ptr(sc), ptr(grid_np), ptr(mean), ptr(None if g_post is None else g_post.contiguous()),
                 ptr(None if g_lv is None else g_lv.contiguous()), ptr(gsc), ctx.steps, b, hh, ww, _lib.stream_ptr())
        return gsc, None, None, None


class FeedForward(nn.Module):
    def __init__(self, model_ksize, model_in_blocks, model_out_blocks, model_chs, model_views,
                 model_cross, model_uncert, model_unet, model_discrete, model_no_batchnorm,
                 model_batchnorm_momentum, val_disp_min, val_disp_max, **kwargs):
        super().__init__()
        assert model_in_blocks >= 1 and model_out_blocks >= 1
        self.ksize, self.chs, self.views = model_ksize, model_chs, model_views
        self.cross, self.uncert, self.discrete = model_cross, model_uncert, model_discrete
        self.no_batchnorm = model_no_batchnorm
        self.batchnorm_momentum = model_batchnorm_momentum
        self.disp_min, self.disp_max = val_disp_min, val_disp_max
        self.steps = (2 if model_cross else 4) * model_views * 3
        # even kernels: pad k//2 then k//2-1 keeps the block size-preserving (feed_forward.py:86-92)
        pad1 = model_ksize // 2
        pad2 = model_ksize // 2 - (0 if model_ksize % 2 else 1)
        bn = not model_no_batchnorm

        def in_net():
            blocks = [_conv_block(model_views * 3, model_chs, model_ksize, pad1, pad2, bn, model_batchnorm_momentum)]
            blocks += [_conv_block(model_chs, model_chs, model_ksize, pad1, pad2, bn, model_batchnorm_momentum)
                       for _ in range(model_in_blocks - 1)]
            return nn.Sequential(*blocks)

        self.in_net_hv = in_net()
        if not model_cross:
            self.in_net_id = in_net()
        c = (2 if model_cross else 4) * model_chs
        oc = 2 if model_uncert else (self.steps if model_discrete else 1)
        if model_unet:       # feed_forward.py:99-100,189-204: one or two output channels, whatever model_discrete says
            oc = 2 if model_uncert else 1
            self.out_net = _UNetTail(c, oc)
        else:
            blocks = [_conv_block(c, c, model_ksize, pad1, pad2, bn, model_batchnorm_momentum)
                      for _ in range(model_out_blocks - 1)]
            blocks.append(_conv_block(c, oc, model_ksize, pad1, pad2, None, None))
            self.out_net = nn.Sequential(*blocks)
        self.out_chs = oc
        self.unet = bool(model_unet)

        self._native_ok = (not model_unet and model_ksize == 2 and not model_cross and bn and model_chs % 2 == 0
                           and (4 * model_chs) % 8 == 0 and 4 * model_chs <= 288 and oc <= 288)
        self._trunk = (Trunk(model_chs, model_in_blocks, model_out_blocks, model_views, oc,
                             model_batchnorm_momentum) if self._native_ok else None)
        self._param_names = [n for n, _ in self.named_parameters()]
        self._grids = {}

    # ------------------------------------------------------------------ helpers
    def _tensor_dict(self):
        """state_dict-keyed parameters and buffers of THIS module instance.  A replica made by
        nn.DataParallel (reference train/cli.py:159; torch.nn.parallel.replicate) has no registered parameters:
        its per-device copies are plain attributes listed in `_former_parameters`, connected by autograd to the
        master's parameters on device 0 -- which is where their gradients are reduced to."""
        if getattr(self, '_is_replica', False):
            d = {}
            for prefix, mod in self.named_modules():
                dot = prefix + '.' if prefix else ''
                for k, t in getattr(mod, '_former_parameters', {}).items():
                    d[dot + k] = t
                for k, t in mod._buffers.items():
                    if t is not None:
                        d[dot + k] = t
            return d
        d = {n: t for n, t in self.named_parameters()}
        d.update({n: t for n, t in self.named_buffers()})
        return d

    def _grid(self, kind, device):
        key = (kind, str(device))
        if key not in self._grids:
            if kind == 'np':  # feed_forward.py:287-288,298-299: float64 linspace assigned to float32
                g = torch.from_numpy(np.linspace(self.disp_min, self.disp_max, self.steps)).float()
            else:             # dl.py:177: torch.linspace in float32
                g = torch.linspace(self.disp_min, self.disp_max, self.steps)
            self._grids[key] = g.to(device)
        return self._grids[key]

    def _torch_trunk(self, h, v, i, d):
        """Plumbing path (stock torch ops on the module tree), feed_forward.py:226-269."""
        b, n, c, hh, ww = h.shape
        h, v = h.view(b, n * c, hh, ww), v.view(b, n * c, hh, ww)
        feats = [self.in_net_hv(h.transpose(2, 3)).transpose(2, 3), self.in_net_hv(v)]
        if not self.cross:
            i, d = i.view(b, n * c, hh, ww), d.view(b, n * c, hh, ww)
            fi = self.in_net_id(i.transpose(2, 3).flip(-1)).flip(-1).transpose(2, 3)
            feats += [fi, self.in_net_id(d)]
        return self.out_net(torch.cat(feats, 1))

    # ------------------------------------------------------------------ forward
    def forward(self, h_views, v_views, i_views=None, d_views=None):
        b, n, c, hh, ww = h_views.shape
        if h_views.is_cuda and self._native_ok:
            _lib.load()  # raises if the HIP library is missing: no silent fallback on the GPU path
            stacks = [h_views, v_views, i_views, d_views]
            for t in stacks:
                if t is None or not t.is_contiguous() or t.dtype != torch.float32 or t.shape != h_views.shape:
                    raise ValueError('FeedForward: four contiguous float32 (b, n, 3, h, w) stacks required')
                if t.device != h_views.device:
                    raise ValueError(f'FeedForward: stacks on different devices ({t.device} vs {h_views.device})')
            td = self._tensor_dict()
            missing = [n for n in self._param_names if n not in td]
            if missing:
                raise RuntimeError(f'FeedForward: no tensor for parameter {missing[0]!r} on this module instance '
                                   '(an unsupported kind of module copy?); use mmlf_amd.train.TrainStep for data '
                                   'parallelism (one process per GPU)')
            params = [td[n] for n in self._param_names]
            if params[0].device != h_views.device:
                raise ValueError(f'FeedForward: parameters on {params[0].device}, input on {h_views.device}')
            save = torch.is_grad_enabled() and any(t.requires_grad for t in params)
            with torch.cuda.device(h_views.device):      # DataParallel worker threads: launch on the replica's device
                output = _TrunkFn.apply(self, self.training, save, *stacks, *params)
        else:
            output = self._torch_trunk(h_views, v_views, i_views, d_views)

        mean = output[:, 0]
        scores = one_hot = posterior = logvar = None
        native = output.is_cuda and self._native_ok
        if self.discrete:
            scores = output
            if native:
                with torch.cuda.device(scores.device):
                    one_hot, posterior, mean, logvar = _HeadDppFn.apply(scores, self._grid('torch', scores.device),
                                                                        self._grid('np', scores.device), self.steps)
            else:
                one_hot = (torch.max(scores, 1, keepdim=True)[0] == scores).float()
                e = torch.exp(scores)
                posterior = e / torch.sum(e, 1, keepdim=True)
                mean = torch.sum(self._grid('torch', scores.device).view(1, -1, 1, 1) * one_hot, 1)
                g = self._grid('np', scores.device).view(1, -1, 1, 1)
                logvar = torch.log(torch.sum((g - mean.unsqueeze(1)) ** 2.0 * posterior, 1))
        if self.uncert:
            logvar = output[:, 1]
            if native:
                with torch.cuda.device(output.device):
                    posterior = _HeadUprFn.apply(output, self._grid('np', output.device), self.steps)
            else:
                g = self._grid('np', output.device).view(1, -1, 1, 1).expand(b, self.steps, hh, ww)
                posterior = laplacian(g, mean, torch.exp(logvar))
        return {'mean': mean, 'logvar': logvar, 'scores': scores, 'one_hot': one_hot,
                'posterior': posterior}
