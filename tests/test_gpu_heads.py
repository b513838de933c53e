"""The head outputs under autograd on the native path (round 5).

The reference builds `posterior` (UPR: a Laplace density over 108 depth bins; DPP: the softmax of the scores) and the DPP
`logvar` as differentiable functions of the network output (reference mmlf/model/feed_forward.py:276-302).  None of its
losses uses them (mmlf/model/loss.py:70,146,264), but the graph is there; the native path used to compute them from a
detached tensor.  They are autograd Functions over mmlf_head_upr / mmlf_head_dpp with backward kernels now: checked here
against torch's own autograd of the reference's expressions in float64, and end to end against the stock-torch branch of
the same module on the CPU."""
import numpy as np
import pytest
import torch

from conftest import TINY_KW
from mmlf_amd import synth

pytestmark = pytest.mark.gpu


def _laplacian(x, mu, b):                      # reference feed_forward.py:9-12
    return 1.0 / (2.0 * b.unsqueeze(1)) * torch.exp(-torch.abs(x - mu.unsqueeze(1)) / b.unsqueeze(1))


def test_upr_posterior_gradient_equals_autograd_of_the_reference_expression():
    from mmlf_amd.feed_forward import _HeadUprFn
    dev = torch.device('cuda:0')
    g = torch.Generator().manual_seed(4)
    B, S, H, W = 3, 108, 5, 7
    out = torch.stack([torch.randn((B, H, W), generator=g) * 1.5, torch.rand((B, H, W), generator=g) * 3 - 2], 1)
    go = torch.randn((B, S, H, W), generator=g)
    grid64 = torch.from_numpy(np.linspace(-3.5, 3.5, S))
    grid = grid64.float()
    # float64 autograd of the reference graph (the float32 grid values, as the module uses them)
    o64 = out.double().requires_grad_(True)
    post64 = _laplacian(grid.double().view(1, S, 1, 1).expand(B, S, H, W), o64[:, 0], torch.exp(o64[:, 1]))
    post64.backward(go.double())
    od = out.to(dev).requires_grad_(True)
    post = _HeadUprFn.apply(od, grid.to(dev), S)
    post.backward(go.to(dev))
    torch.testing.assert_close(post.detach().cpu().double(), post64.detach(), rtol=2e-5, atol=1e-7)
    ref = o64.grad
    err = (od.grad.cpu().double() - ref).abs().max() / ref.abs().max()
    assert float(err) <= 2e-5, float(err)


@pytest.mark.parametrize('which', ['posterior', 'logvar', 'both', 'posterior_alone', 'logvar_alone'])
def test_dpp_head_gradients_equal_autograd_of_the_reference_expressions(which):
    """*_alone (round 6): the other output is not part of the graph at all, so its gradient arrives as None
    (ctx.set_materialize_grads(False)) and mmlf_head_dpp_bwd runs with a NULL pointer for it"""
    from mmlf_amd.feed_forward import _HeadDppFn
    dev = torch.device('cuda:0')
    g = torch.Generator().manual_seed(5)
    B, S, H, W = 2, 108, 4, 6
    scores = torch.randn((B, S, H, W), generator=g) * 2
    go_post, go_lv = torch.randn((B, S, H, W), generator=g), torch.randn((B, H, W), generator=g)
    grid_t = torch.linspace(-3.5, 3.5, S)                                     # dl.py:177
    grid_np = torch.from_numpy(np.linspace(-3.5, 3.5, S)).float()             # feed_forward.py:287-288
    s64 = scores.double().requires_grad_(True)
    one_hot = (torch.max(s64, 1, keepdim=True)[0] == s64).double()
    e = torch.exp(s64)
    post64 = e / torch.sum(e, 1, keepdim=True)
    mean64 = torch.sum(grid_t.double().view(1, -1, 1, 1) * one_hot, 1)
    lv64 = torch.log(torch.sum((grid_np.double().view(1, -1, 1, 1) - mean64.unsqueeze(1)) ** 2.0 * post64, 1))
    alone = which.endswith('_alone')
    which = which.replace('_alone', '')
    loss64 = (post64 * go_post.double()).sum() * (which != 'logvar') + (lv64 * go_lv.double()).sum() * (which != 'posterior')
    loss64.backward()
    sd = scores.to(dev).requires_grad_(True)
    oh, post, mean, lv = _HeadDppFn.apply(sd, grid_t.to(dev), grid_np.to(dev), S)
    assert not oh.requires_grad and not mean.requires_grad and post.requires_grad and lv.requires_grad
    if alone:
        loss = (post * go_post.to(dev)).sum() if which == 'posterior' else (lv * go_lv.to(dev)).sum()
    else:
        loss = (post * go_post.to(dev)).sum() * (which != 'logvar') + (lv * go_lv.to(dev)).sum() * (which != 'posterior')
    loss.backward()
    torch.testing.assert_close(mean.cpu().double(), mean64.detach(), rtol=0, atol=1e-6)
    torch.testing.assert_close(lv.detach().cpu().double(), lv64.detach(), rtol=1e-5, atol=1e-5)
    ref = s64.grad
    err = (sd.grad.cpu().double() - ref).abs().max() / ref.abs().max()
    assert float(err) <= 2e-5, float(err)


@pytest.mark.parametrize('variant', ['upr', 'dpp'])
def test_a_loss_on_the_posterior_trains_the_network_on_the_native_path(variant):
    """end to end: d (sum w * posterior [+ logvar]) / d parameters, native cuda path against the stock-torch branch of the
    same module on the CPU (what the reference's graph gives)"""
    from mmlf_amd.feed_forward import FeedForward
    kw = dict(TINY_KW, model_uncert=variant == 'upr', model_discrete=variant == 'dpp')
    state = synth.synth_state(synth.param_spec(**kw), seed=9)
    stacks, _, _ = synth.synth_inputs(2, 16, seed=9)
    g = torch.Generator().manual_seed(9)
    wts = torch.rand((2, 108, 16, 16), generator=g)
    grads = {}
    for dev in ('cpu', 'cuda:0'):
        m = FeedForward(**kw)
        m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in state.items()})
        m.to(dev).train()
        out = m(*[torch.from_numpy(s).to(dev) for s in stacks])
        assert out['posterior'].requires_grad
        loss = (out['posterior'] * wts.to(dev)).sum() + (out['logvar'].sum() if variant == 'dpp' else 0.0)
        loss.backward()
        grads[dev] = {n: p.grad.detach().cpu().double() for n, p in m.named_parameters()}
    # the biases of the convolutions in front of a train-mode BatchNorm have an analytically ZERO gradient (rounding noise
    # of 1e-14 on both devices): compared in absolute terms; every other tensor relative to its own norm
    big = max(float(v.norm()) for v in grads['cpu'].values())
    assert big > 1.0
    worst = 0.0
    for n, ref in grads['cpu'].items():
        got = grads['cuda:0'][n]
        if float(ref.norm()) > 1e-4 * big:
            worst = max(worst, float((got - ref).norm() / ref.norm()))
        else:
            assert n.endswith('.2.bias') and float(got.norm()) <= 1e-6 * big, (n, float(got.norm()))
    assert worst <= 1e-3, worst          # (float32 trunks on two devices: 1e-5 measured; the head kernels are pinned above at 2e-5)
