"""Validate-side distribution metrics: counterparts of the numpy helpers
in reference mmlf/validate/cli.py:17-187, evaluated on the device the model outputs already live on
(float64, like numpy's promotion in the reference), so the 70-component maps of the ensemble never
travel to the host.  The expensive one -- the discretised 70-member Laplace mixture, 70 x 109 x 512^2
float64 exponentials -- is one HIP kernel on CUDA tensors (mmlf_lmm_to_discrete); the cheap reductions
and the CPU path are torch expressions."""
import torch

from . import _lib
from ._lib import call, ptr


def _bin_edges(n_bins, x_min, x_max, device):
    step = (x_max - x_min) / n_bins
    return torch.linspace(x_min - step / 2.0, x_max + step / 2.0, n_bins + 1, dtype=torch.float64, device=device)


def _lmm_hip(n_bins, x_min, x_max, means, logvars):
    """means, logvars: (S, B, H, W) float32 CUDA tensors -> (B, n_bins, H, W) float64."""
    S, B, H, W = means.shape
    means, logvars = means.contiguous().float(), logvars.contiguous().float()
    edges = _bin_edges(n_bins, x_min, x_max, means.device)
    out = torch.empty((B, n_bins, H, W), dtype=torch.float64, device=means.device)
    call('mmlf_lmm_to_discrete', ptr(means), ptr(logvars), ptr(edges), ptr(out), S, B, n_bins, H * W,
         _lib.stream_ptr())
    return out


def _f64(x):
    return x.double()


def cdf_laplace(disp, mean, var):
    """validate/cli.py:74-87"""
    le = disp < mean
    res_le = torch.exp((disp - mean) / var) / 2
    res_ge = 1 - torch.exp(-(disp - mean) / var) / 2
    return torch.where(le, res_le, res_ge)


def laplace_to_discrete(n_bins, x_min, x_max, mean, logvar):
    """validate/cli.py:90-103: probability mass of Laplace(mean, exp(logvar)) in each of n_bins bins.
    The reference computes in numpy float32 x float64 -> float64; exp(logvar) stays float32."""
    if mean.is_cuda:
        return _lmm_hip(n_bins, x_min, x_max, mean.unsqueeze(0), logvar.unsqueeze(0))
    edges = _bin_edges(n_bins, x_min, x_max, mean.device).view(1, -1, 1, 1)
    mean = _f64(mean).unsqueeze(1)
    var = _f64(torch.exp(logvar)).unsqueeze(1)
    cdf = cdf_laplace(edges, mean, var)
    return cdf[:, 1:] - cdf[:, :-1]


def lmm_to_discrete(n_bins, x_min, x_max, means, logvars):
    """validate/cli.py:106-118: mean of the members' discretised Laplacians.  NOTE the reference's
    caller passes exp(logvars) under the name `logvars` (validate/cli.py:302,318); mirror that."""
    if means.is_cuda:
        return _lmm_hip(n_bins, x_min, x_max, means, logvars)
    out = torch.zeros((means.shape[1], n_bins, means.shape[2], means.shape[3]), dtype=torch.float64,
                      device=means.device)
    for i in range(means.shape[0]):
        out += laplace_to_discrete(n_bins, x_min, x_max, means[i], logvars[i])
    return out / means.shape[0]


def mean_to_discrete(n_bins, x_min, x_max, mean):
    """validate/cli.py:121-138"""
    step = (x_max - x_min) / n_bins
    centres = torch.linspace(x_min, x_max, n_bins, dtype=torch.float64, device=mean.device).view(1, -1, 1, 1)
    return (torch.abs(centres - _f64(mean).unsqueeze(1)) < step / 2.0).double()


def multimodal_mask(mpi, threshhold=0.3):
    """validate/cli.py:166-171"""
    return ((mpi[:, :, 3] > threshhold).sum(1) > 1).double()


def kl_divergence(dist, dist_gt, mask=None, inplace=False):
    """validate/cli.py:174-187 (batch size 1, as in validate).  The reference's helper smooths and renormalises ITS
    ARGUMENTS in place (`dist += epsilon; dist /= np.sum(dist, 1)`), and the loop that calls it three times in a row
    (validate/cli.py:333-335) passes the same arrays each time: `inplace=True` does the same to the tensors passed in
    (in their own dtype, as numpy does), `inplace=False` leaves them untouched."""
    eps = 0.00001
    if inplace:
        dist.add_(eps)
        dist_gt.add_(eps)
        dist.div_(dist.sum(1, keepdim=True))
        dist_gt.div_(dist_gt.sum(1, keepdim=True))
    else:
        dist = dist + eps
        dist_gt = dist_gt + eps
        dist = dist / dist.sum(1, keepdim=True)
        dist_gt = dist_gt / dist_gt.sum(1, keepdim=True)
    kld = (dist_gt * torch.log(dist_gt / dist)).sum(1)
    if mask is None:
        return kld.mean()
    return (kld * mask).sum() / mask.sum()


def nll_laplace(mpi, mean, logvar, mask=None):
    """validate/cli.py:26-49"""
    disp, alpha = mpi[:, :, 4], mpi[:, :, 3]
    mean = mean.unsqueeze(1)
    var = torch.exp(logvar.unsqueeze(1))
    prob = torch.exp(-(torch.abs(mean - disp)) / var) / var / 2.0 + 0.00001
    nllh = (alpha * -torch.log(prob)).sum(1)
    if mask is not None:
        return (nllh * mask).sum() / mask.sum()
    return nllh.mean()


def nll_discrete(weights, posterior, vmin=None, vmax=None, mask=None, inplace=False):
    """validate/cli.py:52-73: negative log likelihood of the multi-plane target under a discrete posterior (the 7.0 is
    the reference's bin-width constant).  Like `kl_divergence`, the reference's helper rewrites its arguments in place
    (`posterior += epsilon; posterior /= np.sum(posterior, 1) * 7.0`) -- and for --model_discrete the loop's `dist` IS that
    `posterior` array (validate/cli.py:321): `inplace=True` reproduces it, `inplace=False` leaves the inputs untouched."""
    eps = 0.00001
    if inplace:
        weights.add_(eps)
        posterior.add_(eps)
        weights.div_(weights.sum(1, keepdim=True))
        posterior.div_(posterior.sum(1, keepdim=True) * 7.0)
    else:
        weights = weights + eps                          # the array's own dtype in BOTH branches, as the reference's numpy
        posterior = posterior + eps                      # helper works (validate/cli.py:52-73): the two modes then differ
                                                         # by the in-place smoothing alone, as the docstring says
        weights = weights / weights.sum(1, keepdim=True)
        posterior = posterior / (posterior.sum(1, keepdim=True) * 7.0)
    nllh = (weights * -torch.log(posterior)).sum(1)
    if mask is not None:
        return (nllh * mask).sum() / mask.sum()
    return nllh.mean()
