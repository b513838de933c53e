"""CPU tests of the drop-in boundary: state_dict key set, plumbing-path forward vs goldens,
C-ABI symbol export.  No GPU compute."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

from conftest import BASE_KW, ROOT, TINY_KW, VARIANTS, load_golden
from mmlf_amd import synth
from mmlf_amd.feed_forward import FeedForward


def _load(model, state):
    model.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in state.items()})


@pytest.mark.parametrize('variant', list(VARIANTS))
def test_state_dict_keys_and_shapes(variant):
    kw = dict(BASE_KW, **VARIANTS[variant])
    model = FeedForward(**kw, train_lr=1e-3, some_other_cli_flag=True)  # extra CLI kwargs are swallowed
    spec = synth.param_spec(**kw)
    sd = model.state_dict()
    assert list(sd.keys()) == [n for n, _, _ in spec]
    for n, shape, _ in spec:
        assert tuple(sd[n].shape) == tuple(shape), n
    n_params = sum(p.numel() for p in model.parameters())
    assert n_params == {'base': 4612166, 'upr': 4613300, 'dpp': 4778872}[variant]
    assert model.steps == 108 and model.disp_min == -3.5 and model.disp_max == 3.5


@pytest.mark.parametrize('variant', list(VARIANTS))
def test_cpu_plumbing_forward_matches_reference_golden(variant):
    g = load_golden(f'g1_tiny_{variant}.npz')
    kw = dict(TINY_KW, **VARIANTS[variant])
    model = FeedForward(**kw)
    _load(model, {k[len('state/'):]: v for k, v in g.items() if k.startswith('state/')})
    model.eval()
    with torch.no_grad():
        out = model(*[torch.from_numpy(g[f'in{i}']) for i in range(4)])
    assert set(out) == {'mean', 'logvar', 'scores', 'one_hot', 'posterior'}
    for k, v in out.items():
        if v is None:
            assert f'eval_{k}' not in g
        else:
            np.testing.assert_allclose(v.numpy(), g[f'eval_{k}'], rtol=1e-5, atol=1e-6, err_msg=k)


def test_config0_base_full_size_cpu_forward():
    """BASELINE.json configs[0]: BASE forward on one synthetic 96x96 patch, CPU only."""
    g = load_golden('g2_full_base.npz')
    model = FeedForward(**BASE_KW)
    _load(model, synth.synth_state(synth.param_spec(**BASE_KW), seed=21))
    stacks, _, _ = synth.synth_inputs(1, 96, seed=7)
    model.eval()
    with torch.no_grad():
        out = model(*[torch.from_numpy(s) for s in stacks])
    assert np.abs(out['mean'].numpy() - g['eval_mean']).mean() < 1e-5


def test_non_default_flags_take_the_stock_torch_path():
    m = FeedForward(**dict(TINY_KW, model_cross=True))
    assert not hasattr(m, 'in_net_id') and m.steps == 54 and not m._native_ok


@pytest.mark.parametrize('dev', ['cpu', pytest.param('cuda', marks=pytest.mark.gpu)])
def test_model_unet_fallback_matches_reference_golden(dev):
    """--model_unet stays accepted and runs stock torch ops (reference feed_forward.py:99-100,
    189-204 + unet.py); same state_dict keys, same outputs, loss and gradients as the reference module."""
    from mmlf_amd import loss as loss_mod
    g = load_golden('g10_unet.npz')
    kw = dict(TINY_KW, model_unet=True, model_uncert=True)
    model = FeedForward(**kw)
    assert list(model.state_dict().keys()) == list(g['keys']) and not model._native_ok
    _load(model, synth.formula_state([(k, v.shape) for k, v in model.state_dict().items()], seed=3))
    model.to(dev)
    stacks, gt, mask = synth.synth_inputs(2, 32, seed=9)
    t = [torch.from_numpy(s).to(dev) for s in stacks]
    tol = dict(rtol=1e-5, atol=1e-6) if dev == 'cpu' else dict(rtol=2e-3, atol=2e-4)   # MIOpen vs mkldnn summation order
    model.train()
    out = model(*t)
    m = torch.from_numpy(mask).int() * loss_mod.create_mask_margin(mask.shape, 11)
    val = loss_mod.ImprovedUncertaintyL1Loss()(out, torch.from_numpy(gt).to(dev), m.to(dev), None)
    val.backward()
    np.testing.assert_allclose(out['mean'].detach().cpu().numpy(), g['train_mean'], **tol)
    np.testing.assert_allclose(out['logvar'].detach().cpu().numpy(), g['train_logvar'], **tol)
    np.testing.assert_allclose(val.item(), g['loss'], rtol=tol['rtol'])
    for name in ('out_net.last.weight', 'in_net_hv.0.0.weight'):
        got = dict(model.named_parameters())[name].grad.cpu().numpy()
        ref = g[f'grad/{name}']
        # (cuda: MIOpen's kernels against the CPU golden; the first layer's gradient has crossed 4 max-pools and 20
        # train-mode BatchNorms over 2 x 32 x 32 samples: 4 % apart, measured)
        lim = 1e-4 if dev == 'cpu' else (2e-2 if name.startswith('out_net') else 1e-1)
        assert np.linalg.norm(got - ref) <= lim * np.linalg.norm(ref), name
    model.eval()
    with torch.no_grad():
        out = model(*t)
    np.testing.assert_allclose(out['mean'].cpu().numpy(), g['eval_mean'], **tol)
    np.testing.assert_allclose(out['logvar'].cpu().numpy(), g['eval_logvar'], **tol)


def test_c_abi_exports_every_declared_symbol():
    from mmlf_amd import _lib
    from mmlf_amd.csrc import build
    build.build(verbose=False)
    header = open(os.path.join(ROOT, 'include', 'mmlf_hip.h')).read()
    declared = set(re.findall(r'\b(mmlf_[a-z0-9_]+)\s*\(', header))
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for name in declared:
        assert hasattr(lib, name), name
    # one number, one place: the header's macro is what the library returns and what the binding checks
    macro = int(re.search(r'^#define\s+MMLF_ABI_VERSION\s+(\d+)', header, re.M).group(1))
    assert lib.mmlf_abi_version() == _lib.ABI_VERSION == macro
    lib.mmlf_grid_alloc_positions.restype = ctypes.c_int64
    assert lib.mmlf_grid_alloc_positions(2, 96, 96) >= 2 * 98 * 98 + 99
