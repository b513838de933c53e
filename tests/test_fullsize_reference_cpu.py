"""The independent reference of the full-size GPU tests, checked by itself (no GPU).

tests/test_gpu_fullsize.py compares the native path at bs=512 against stock torch ops run through two helpers written in
that file -- `_ChunkedConv` (nn.Conv2d evaluated in batch chunks, its weight gradient summed over the chunks in float64)
and `_torch_forward_checkpointed` (the module tree block by block under torch.utils.checkpoint) -- because plain
autograd neither fits nor is right at that size (the stock full-batch weight gradient is wrong there).  Those helpers
are pure torch: here they are held, in float64 on the CPU and with chunks that do not divide the batch, against the
plain forward + autograd of the same module (reference mmlf/model/feed_forward.py:226-269, mmlf/train/cli.py:257), so
that what the GPU test trusts is verified independently of the kernels it judges."""
import numpy as np
import torch

from conftest import BASE_KW
from mmlf_amd import synth


def _module(kw, seed):
    from mmlf_amd.feed_forward import FeedForward
    m = FeedForward(**kw)
    m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in
                       synth.synth_state(synth.param_spec(**kw), seed=seed).items()})
    return m.double().train()


def test_chunked_checkpointed_reference_equals_plain_autograd(monkeypatch):
    import test_gpu_fullsize as full
    monkeypatch.setattr(full._ChunkedConv, 'CH', 2)           # chunks of 2, 2, 1 over a batch of 5
    kw = dict(BASE_KW, model_uncert=True)
    gen = torch.Generator().manual_seed(3)
    B, ps = 5, 10
    stacks = [torch.rand((B, 9, 3, ps, ps), generator=gen, dtype=torch.float64) for _ in range(4)]
    gt = 4.0 * torch.rand((B, ps, ps), generator=gen, dtype=torch.float64) - 2.0
    res = {}
    for path in ('plain', 'helpers'):
        m = _module(kw, seed=23)
        if path == 'plain':
            out = m(*stacks)                                   # the stock-torch branch of the module (CPU tensors)
            mean, logvar = out['mean'], out['logvar']
        else:
            o = full._torch_forward_checkpointed(m, *stacks)
            mean, logvar = o[:, 0], o[:, 1]
        lossv = (torch.exp(-logvar) * torch.abs(mean - gt) + logvar).mean()
        lossv.backward()
        res[path] = (float(lossv), mean.detach().clone(), {n: p.grad.clone() for n, p in m.named_parameters()})
    assert abs(res['plain'][0] - res['helpers'][0]) <= 1e-12 * abs(res['plain'][0])
    assert torch.allclose(res['plain'][1], res['helpers'][1], rtol=1e-11, atol=1e-13)
    scale = max(float(g.abs().max()) for g in res['plain'][2].values())
    for n, g in res['plain'][2].items():
        # the helper hands its float64 weight-gradient sum back as float32 (the GPU tests' models are float32): 2^-24 per element
        # is the bar; a conv bias in front of BatchNorm has a true-zero gradient: held absolutely against the largest gradient
        assert float((res['helpers'][2][n] - g).abs().max()) <= 2.0 ** -22 * max(float(g.abs().max()), 1e-3 * scale), n
