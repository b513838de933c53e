"""Losses and masks of the train / validate step, reference mmlf/model/loss.py.

Each module takes ``(output_dict, target, mask[, mask_padding])`` like the reference.  On CUDA
tensors the training losses -- single-mode, multimodal and padded -- run fused HIP kernels (value +
gradient in one call, ``mmlf_loss_fwd_bwd`` / ``mmlf_loss_multi_fwd_bwd``) behind a small autograd node;
on CPU they are plain torch expressions.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import _lib
from ._lib import call, ptr
from .engine import LOSS_BLOCKS

KIND_L1, KIND_UPR, KIND_CE = 0, 1, 2
KIND_MULTI_L1, KIND_MULTI_UPR, KIND_MULTI_CE, KIND_UPR_PADDED = 3, 4, 5, 6   # mmlf_loss_multi_fwd_bwd


def create_mask_margin(shape, margin=0):
    """Boolean mask that is False in a `margin`-wide frame of the last two dims (loss.py:6-26)."""
    assert margin >= 0
    mask = torch.ones(shape, dtype=torch.bool)
    if margin > 0:
        mask[..., :margin, :] = False
        mask[..., -margin:, :] = False
        mask[..., :margin] = False
        mask[..., -margin:] = False
    return mask


def native_loss(kind, output, gt, mask, grid_torch=None, half_step=0.0, want_grad=True, den_override=None):
    """Fused loss on the raw trunk output (B,oc,H,W).  Returns (loss scalar tensor, grad or None)."""
    B, oc, H, W = output.shape
    dev = output.device
    for name, t in (('target', gt), ('mask', mask), ('grid', grid_torch), ('den_override', den_override)):
        if t is not None and t.device != dev:       # a raw pointer of another device would fault inside the kernel
            raise ValueError(f'loss: {name} is on {t.device}, the model output on {dev}')
    output = output.contiguous()
    gt = gt.contiguous().float()
    mask = mask.contiguous().to(torch.int32)
    loss = torch.empty((), dtype=torch.float32, device=dev)
    grad = torch.empty_like(output) if want_grad else None
    if want_grad and kind != KIND_CE and oc > (1 if kind == KIND_L1 else 2):
        grad.zero_()
    scratch = torch.empty(2 * LOSS_BLOCKS + 2, dtype=torch.float64, device=dev)
    call('mmlf_loss_fwd_bwd', kind, ptr(output), oc, ptr(gt), ptr(mask), ptr(grid_torch), float(half_step),
         ptr(loss), ptr(grad), ptr(scratch), LOSS_BLOCKS, ptr(den_override), B, H, W, _lib.stream_ptr())
    return loss, grad


def native_multi_loss(kind, output, target, mask, mask_padding=None, grid_torch=None, half_step=0.0, want_grad=True,
                      den_override=None, aux_override=None):
    """Fused multimodal / padded loss on the raw trunk output (B,oc,H,W); target = mpi (B,P,5,H,W), or gt (B,H,W)
    for KIND_UPR_PADDED.  Returns (loss scalar tensor, grad or None)."""
    B, oc, H, W = output.shape
    dev = output.device
    for name, t in (('target', target), ('mask', mask), ('mask_padding', mask_padding), ('grid', grid_torch),
                    ('den_override', den_override), ('aux_override', aux_override)):
        if t is not None and t.device != dev:
            raise ValueError(f'loss: {name} is on {t.device}, the model output on {dev}')
    output = output.contiguous()
    target = target.contiguous().float()
    P = 0
    if kind != KIND_UPR_PADDED:
        if target.dim() != 5 or target.shape[0] != B or target.shape[2] != 5 or tuple(target.shape[3:]) != (H, W):
            raise ValueError(f'multimodal loss: target must be (B, P, 5, H, W), got {tuple(target.shape)}')
        P = target.shape[1]
    elif tuple(target.shape) != (B, H, W) or mask_padding is None:
        raise ValueError('padded loss: target (B, H, W) and mask_padding required')
    mask = mask.contiguous().to(torch.int32)
    if mask_padding is not None:
        mask_padding = mask_padding.contiguous().to(torch.int32)
    loss = torch.empty((), dtype=torch.float32, device=dev)
    grad = None
    if want_grad:
        full = kind == KIND_MULTI_CE or oc == (1 if kind == KIND_MULTI_L1 else 2)
        grad = torch.empty_like(output) if full else torch.zeros_like(output)
    n = int(_lib.load().mmlf_loss_multi_scratch_doubles(LOSS_BLOCKS))
    scratch = torch.empty(n, dtype=torch.float64, device=dev)
    call('mmlf_loss_multi_fwd_bwd', kind, ptr(output), oc, ptr(target), P, ptr(mask), ptr(mask_padding),
         ptr(grid_torch), float(half_step), ptr(loss), ptr(grad), ptr(scratch), LOSS_BLOCKS, ptr(den_override),
         ptr(aux_override), B, H, W, _lib.stream_ptr())
    return loss, grad


class _NativeMultiLossFn(torch.autograd.Function):
    """autograd node of the multimodal / padded losses for callers that go through the loss modules"""

    @staticmethod
    def forward(ctx, kind, target, mask, mask_padding, *heads):
        out = torch.stack(list(heads), 1)
        loss, grad = native_multi_loss(kind, out.detach(), target, mask, mask_padding)
        ctx.save_for_backward(grad)
        return loss

    @staticmethod
    def backward(ctx, gl):
        (grad,) = ctx.saved_tensors
        g = grad * gl
        return (None, None, None, None) + tuple(g[:, k] for k in range(g.shape[1]))


class _NativeLossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, kind, gt, mask, grid, half_step, *heads):
        # heads: (mean,) | (mean, logvar) | (scores,)
        out = heads[0].unsqueeze(1) if kind != KIND_CE else heads[0]
        if kind == KIND_UPR:
            out = torch.stack([heads[0], heads[1]], 1)
        loss, grad = native_loss(kind, out.detach(), gt, mask, grid, half_step, True)
        ctx.kind = kind
        ctx.save_for_backward(grad)
        return loss

    @staticmethod
    def backward(ctx, gl):
        (grad,) = ctx.saved_tensors
        g = grad * gl
        if ctx.kind == KIND_L1:
            hg = (g[:, 0],)
        elif ctx.kind == KIND_UPR:
            hg = (g[:, 0], g[:, 1])
        else:
            hg = (g,)
        return (None, None, None, None, None) + hg


def _masked_mean(loss, mask):
    count = mask.int().sum()
    loss = loss * mask.float()
    if count == 0:
        return loss.sum()
    return loss.sum() / count


class MaskedL1Loss(nn.Module):
    """loss.py:29-77"""

    def forward(self, input, target, mask):
        if input['mean'].is_cuda:
            return _NativeLossFn.apply(KIND_L1, target, mask, None, 0.0, input['mean'])
        return _masked_mean(torch.abs(input['mean'] - target), mask)


class MaskedMSELoss(nn.Module):
    """loss.py:106-122"""

    def forward(self, input, target, mask):
        return _masked_mean((input['mean'] - target) ** 2.0, mask)


class MaskedBadPix(nn.Module):
    """loss.py:163-187"""

    def __init__(self, t=0.07):
        super().__init__()
        self.t = t

    def forward(self, input, target, mask):
        bad = (torch.abs(input['mean'] - target) > self.t).int() * mask.int()
        count = mask.int().sum()
        if count == 0:
            return bad.sum()
        return bad.sum().float() / count


class ImprovedUncertaintyL1Loss(nn.Module):
    """loss.py:254-294"""

    def forward(self, input, target, mask, mask_padding=None):
        mean, logvar = input['mean'], input['logvar']
        if mean.is_cuda and mask_padding is None:
            return _NativeLossFn.apply(KIND_UPR, target, mask, None, 0.0, mean, logvar)
        if mean.is_cuda:
            return _NativeMultiLossFn.apply(KIND_UPR_PADDED, target, mask, mask_padding, mean, logvar)
        loss = torch.exp(-logvar) * torch.abs(mean - target) + logvar
        if mask_padding is not None:
            mp = mask_padding.float()
            loss = loss * mp
            if mp.sum() > 0:
                loss = loss * (mp.numel() / mp.sum())
            oor = 1.0 - mp
            loss_oor = -logvar * oor
            if oor.sum() > 0:
                loss_oor = loss_oor * (oor.numel() / oor.sum())
            loss = (loss + loss_oor) / 2.0
        return _masked_mean(loss, mask)


class MaskedCrossEntropy(nn.Module):
    """loss.py:137-160: relu(scores), -log(exp(<s,t>) / sum exp(s)), masked mean."""

    def forward(self, input, target, mask):
        s = F.relu(input['scores'])
        loss = -torch.log(torch.exp(torch.sum(s * target, 1)) / torch.sum(torch.exp(s), 1))
        return _masked_mean(loss, mask)


# ------------------------------------------------------------------ multimodal training losses
# (reached with --train_loss_multimodal, reference train/cli.py:120-123,224-225).
# `target` is the multi-plane tensor (B, P, 5, H, W): channel 3 = alpha, channel 4 = disparity.
class MultiMaskedL1Loss(nn.Module):
    """loss.py:80-103: alpha-weighted L1 to every plane, masked mean."""

    def forward(self, input, target, mask):
        if input['mean'].is_cuda:
            return _NativeMultiLossFn.apply(KIND_MULTI_L1, target, mask, None, input['mean'])
        weights, targets = target[:, :, 3], target[:, :, 4]
        diff = (torch.abs(input['mean'].unsqueeze(1) - targets) * weights).sum(1)
        return _masked_mean(diff, mask)


class ImprovedMultiUncertaintyL1Loss(nn.Module):
    """loss.py:336-372: alpha-weighted Laplace NLL normalised by the mean total alpha, plus a
    -logvar term on pixels without any surface (total alpha < 0.01), each side rescaled; masked mean."""

    def forward(self, input, target, mask, mask_padding=None):
        mean, logvar = input['mean'], input['logvar']
        if mean.is_cuda:
            return _NativeMultiLossFn.apply(KIND_MULTI_UPR, target, mask, None, mean, logvar)
        weights, targets = target[:, :, 3], target[:, :, 4]
        loss = torch.exp(-logvar).unsqueeze(1) * torch.abs(mean.unsqueeze(1) - targets) + logvar.unsqueeze(1)
        total = weights.sum(1)
        loss = (loss * weights).sum(1) / torch.mean(total)
        oor = (total < 0.01).float()
        loss_oor = -logvar * oor * (oor.numel() / torch.sum(oor))
        return _masked_mean((loss + loss_oor) / 2.0, mask)
