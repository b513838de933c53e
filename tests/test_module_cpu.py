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


def test_unsupported_flags():
    with pytest.raises(NotImplementedError):
        FeedForward(**dict(TINY_KW, model_unet=True))
    m = FeedForward(**dict(TINY_KW, model_cross=True))
    assert not hasattr(m, 'in_net_id') and m.steps == 54


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
    assert lib.mmlf_abi_version() == 3
    lib.mmlf_grid_alloc_positions.restype = ctypes.c_int64
    assert lib.mmlf_grid_alloc_positions(2, 96, 96) >= 2 * 98 * 98 + 99
