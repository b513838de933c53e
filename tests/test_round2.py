"""Round-2 rows: eval-mode training (--train_eval_mode), the padded and multimodal losses on the HIP path,
a checkpoint written by the REFERENCE's own ModelSaver, metrics on the GPU, the nn.DataParallel replica
surface.  Goldens: tests/golden/g8_*.npz, g9_checkpoint.* (made by tests/golden/make_golden.py g8 from the
reference)."""
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN, TINY_KW, load_golden
from mmlf_amd import dl, loss, synth
from mmlf_amd.feed_forward import FeedForward
from mmlf_amd.train import TrainStep

DEVICES = ['cpu', pytest.param('cuda', marks=pytest.mark.gpu)]


def _fresh(kw, state, dev):
    m = FeedForward(**kw)
    m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in state.items()})
    return m.to(dev)


@pytest.mark.parametrize('dev', DEVICES)
def test_upr_loss_with_mask_padding(dev):
    """--train_loss_padding without --train_loss_multimodal (reference train/cli.py:221-222, loss.py:264-294):
    value and both gradients, torch expressions on CPU, mmlf_loss_multi_fwd_bwd(kind 6) on the GPU."""
    g5, g8 = load_golden('g5_losses.npz'), load_golden('g8_losses.npz')
    t = lambda a: torch.from_numpy(a).to(dev)
    o = {'mean': t(g5['mean']).requires_grad_(True), 'logvar': t(g5['logvar']).requires_grad_(True)}
    val = loss.ImprovedUncertaintyL1Loss()(o, t(g5['gt']), t(g5['mask']), t(g5['mask_padding']))
    val.backward()
    np.testing.assert_allclose(val.item(), g8['upr_padding'], rtol=2e-6)
    np.testing.assert_allclose(val.item(), g5['upr_padding'], rtol=2e-6)
    np.testing.assert_allclose(o['mean'].grad.cpu().numpy(), g8['dupr_padding_dmean'], rtol=2e-5, atol=1e-9)
    np.testing.assert_allclose(o['logvar'].grad.cpu().numpy(), g8['dupr_padding_dlogvar'], rtol=2e-5, atol=1e-9)


@pytest.mark.parametrize('dev', DEVICES)
def test_multimodal_cross_entropy_and_padded_multimodal_losses(dev):
    g6, g8 = load_golden('g6_multimodal.npz'), load_golden('g8_losses.npz')
    t = lambda a: torch.from_numpy(a).to(dev)
    mpi, mask = t(g6['mpi']), t(g6['mask'])
    # DPP with --train_loss_multimodal: cross entropy on mpi_to_weights(mpi) (train/cli.py:201-204,249)
    sc = t(g8['scores']).requires_grad_(True)
    if dev == 'cuda':
        grid = torch.linspace(-3.5, 3.5, 108).to(dev)
        val, grad = loss.native_multi_loss(loss.KIND_MULTI_CE, sc.detach(), mpi, mask, None, grid, 7.0 / 108 / 2.0)
    else:
        val = loss.MaskedCrossEntropy()({'scores': sc}, dl.mpi_to_weights(mpi, -3.5, 3.5, 108), mask)
        val.backward()
        grad = sc.grad
    np.testing.assert_allclose(float(val.detach()), g8['multi_ce'], rtol=5e-6)
    np.testing.assert_allclose(grad.cpu().numpy(), g8['dmulti_ce_dscores'], rtol=5e-5, atol=2e-9)
    # --train_loss_padding with --train_loss_multimodal: out-of-range planes lose their alpha (train/cli.py:219-220)
    mpi_p = mpi.clone()
    mpi_p[:, :, 3] *= (torch.abs(mpi_p[:, :, 4]) < float(g8['pad'])).float()
    for name, fn, keys in (('multi_l1_pad', loss.MultiMaskedL1Loss(), ['mean']),
                           ('multi_upr_pad', loss.ImprovedMultiUncertaintyL1Loss(), ['mean', 'logvar'])):
        o = {'mean': t(g6['mean']).requires_grad_(True), 'logvar': t(g6['logvar']).requires_grad_(True)}
        val = fn(o, mpi_p, mask)
        val.backward()
        np.testing.assert_allclose(val.item(), g8[name], rtol=2e-6)
        for k in keys:
            np.testing.assert_allclose(o[k].grad.cpu().numpy(), g8[f'd{name}_d{k}'], rtol=2e-5, atol=1e-9)


@pytest.mark.parametrize('dev', DEVICES)
@pytest.mark.parametrize('variant', ['base', 'upr', 'dpp'])
def test_train_step_multimodal_with_padding(dev, variant):
    """TrainStep(loss_multimodal=True, loss_padding=...) equals the same loss through plain autograd on the
    module path, for all three heads (the DPP target comes from the UNPADDED planes, train/cli.py:201-204)."""
    g6 = load_golden('g6_multimodal.npz')
    kw = dict(TINY_KW, model_uncert=(variant == 'upr'), model_discrete=(variant == 'dpp'))
    state = synth.synth_state(synth.param_spec(**kw), 9)
    stacks, _, _ = synth.synth_inputs(2, 14, seed=3, ps_w=18)
    data = [torch.from_numpy(s).to(dev) for s in stacks]
    mpi, mask = torch.from_numpy(g6['mpi']).to(dev), torch.from_numpy(g6['mask']).to(dev)
    st = TrainStep(_fresh(kw, state, dev), lr=1e-3, loss_margin=2, loss_multimodal=True, loss_padding=3.0)
    l1 = st(*data, mpi, mask, 1)
    m2 = _fresh(kw, state, 'cpu')          # reference arithmetic: stock torch ops + autograd on the CPU
    m2.train()
    cmpi, cmask = mpi.cpu(), st._mask(mask).cpu()
    out = m2(*[d.cpu() for d in data])
    if variant == 'dpp':
        l2 = loss.MaskedCrossEntropy()(out, dl.mpi_to_weights(cmpi, -3.5, 3.5, 108), cmask)
    else:
        mpi_p = cmpi.clone()
        mpi_p[:, :, 3] *= (torch.abs(mpi_p[:, :, 4]) < 3.0).float()
        fn = loss.ImprovedMultiUncertaintyL1Loss() if variant == 'upr' else loss.MultiMaskedL1Loss()
        l2 = fn(out, mpi_p, cmask)
    np.testing.assert_allclose(float(l1), float(l2.detach()), rtol=2e-5)
    l2.backward()
    for (n, o, cnt), (_, p) in zip(st.layout, m2.named_parameters()):
        ref, got = p.grad.reshape(-1), st.grad[o:o + cnt].cpu()
        assert float((got - ref).norm()) <= 2e-3 * float(ref.norm()) + 1e-6, n


@pytest.mark.parametrize('dev', DEVICES)
def test_train_eval_mode_vs_reference(dev):
    """--train_eval_mode (reference train/cli.py:227-230): the optimisation step runs with model.eval(), i.e.
    BatchNorm normalises with its RUNNING statistics, does not update them, and the gradient treats them as
    constants.  Loss, every parameter gradient and the post-Adam state against the reference."""
    g = load_golden('g8_evalmode_upr.npz')
    kw = dict(TINY_KW, model_uncert=True)
    state = synth.synth_state(synth.param_spec(**kw), seed=12)
    st = TrainStep(_fresh(kw, state, dev), lr=1e-3, loss_margin=0, train_eval_mode=True)
    data = [torch.from_numpy(g[f'in{i}']).to(dev) for i in range(4)]
    l = st(*data, torch.from_numpy(g['gt']).to(dev), torch.from_numpy(g['mask']).to(dev), 1)
    assert not st.model.training
    np.testing.assert_allclose(float(l), g['loss'], rtol=2e-5)
    for n, o, cnt in st.layout:
        ref = g[f'grad/{n}'].reshape(-1)
        got = st.grad[o:o + cnt].cpu().numpy()
        scale = max(np.abs(ref).max(), 1e-6)
        assert np.abs(got - ref).max() <= 5e-4 * scale + 5e-7, n
    for k, v in st.model.state_dict().items():
        ref = g[f'post/{k}']
        if 'running' in k or 'num_batches' in k:
            np.testing.assert_array_equal(v.cpu().numpy(), ref, err_msg=k)      # untouched in eval mode
            np.testing.assert_array_equal(ref, np.asarray(state[k]), err_msg=k)


@pytest.mark.gpu
def test_eval_mode_backward_through_autograd_equals_module_path():
    """backward through an eval-mode forward on the native path (model.eval() + loss.backward(), the
    reference's --train_eval_mode loop shape) against the stock-torch module path."""
    kw = dict(TINY_KW, model_uncert=True)
    state = synth.synth_state(synth.param_spec(**kw), seed=31)
    for k in list(state):                       # make running statistics non-trivial
        if k.endswith('running_mean'):
            state[k] = (np.asarray(state[k]) + 0.3).astype(np.float32)
    stacks, gt, mask = synth.synth_inputs(2, 20, seed=8)
    grads = {}
    for dev in ('cpu', 'cuda'):
        m = _fresh(kw, state, dev)
        m.eval()
        out = m(*[torch.from_numpy(s).to(dev) for s in stacks])
        l = loss.ImprovedUncertaintyL1Loss()(out, torch.from_numpy(gt).to(dev), torch.from_numpy(mask).to(dev), None)
        l.backward()
        grads[dev] = (float(l), {n: p.grad.cpu().numpy() for n, p in m.named_parameters()})
    np.testing.assert_allclose(grads['cuda'][0], grads['cpu'][0], rtol=2e-5)
    for n, ref in grads['cpu'][1].items():
        scale = max(np.abs(ref).max(), 1e-6)
        assert np.abs(grads['cuda'][1][n] - ref).max() <= 5e-4 * scale + 5e-7, n


@pytest.mark.parametrize('dev', DEVICES)
def test_resume_from_a_checkpoint_written_by_the_reference(dev):
    """f3: tests/golden/g9_checkpoint.pt was written by the reference's ModelSaver (mmlf/utils/dl.py:21-74) from a
    DataParallel-wrapped model and torch.optim.Adam after two steps.  Loading it like train/cli.py:137-157 into a
    TrainStep and taking step 3 must land where the reference's own resumed run landed."""
    g = load_golden('g9_checkpoint_next.npz')
    path = os.path.join(GOLDEN, 'g9_checkpoint.pt')
    ck = torch.load(path, map_location='cpu')
    assert set(ck) == {'model_state_dict', 'optimizer_state_dict', 'hyper_parameters', 'epoch', 'iteration', 'loss'}
    hp = ck['hyper_parameters']
    m = FeedForward(**hp).to(dev)               # validate/cli.py:214-227: the stored click kwargs rebuild the model
    st = TrainStep(m, lr=1e-5, loss_margin=0)
    it, _ = dl.load_checkpoint(path, st.model, st, lr=hp['train_lr'], map_location=dev)
    assert it == int(g['iteration']) == 2 and st.adam_steps == 2 and st.lr == 1e-3
    data = [torch.from_numpy(g[f'in{i}']).to(dev) for i in range(4)]
    l = st(*data, torch.from_numpy(g['gt']).to(dev), torch.from_numpy(g['mask']).to(dev), it + 1)
    np.testing.assert_allclose(float(l), g['loss3'], rtol=2e-5)
    for k, v in st.model.state_dict().items():
        if 'num_batches' in k:
            assert int(v) == int(g[f'post/{k}']), k
        elif dev == 'cuda' and k.endswith('.2.bias') and not k.startswith('out_net.2.'):
            # a conv bias in front of BatchNorm has a true-zero gradient: Adam turns the rounding noise of ANY
            # arithmetic into +-lr steps (the reference's own CPU and GPU runs would differ here as well)
            assert np.abs(v.cpu().numpy() - g[f'post/{k}']).max() <= 2.5e-3, k
        else:
            np.testing.assert_allclose(v.cpu().numpy(), g[f'post/{k}'], rtol=1e-4, atol=3e-6, err_msg=k)
    # and the file our own ModelSaver writes loads into the reference's optimizer class unchanged
    opt = torch.optim.Adam(FeedForward(**hp).parameters(), lr=1e-3)
    opt.load_state_dict(st.optimizer_state_dict())
    assert float(opt.state_dict()['state'][0]['step']) == 3.0


@pytest.mark.gpu
def test_validate_metrics_on_gpu_vs_reference():
    """a15: MaskedMSELoss / MaskedBadPix with the 15-px margin of validate/cli.py:271-280 on cuda tensors against the
    reference's values (G5), plus the L1 family for completeness."""
    g = load_golden('g5_losses.npz')
    t = lambda k: torch.from_numpy(g[k]).cuda()
    out = {'mean': t('mean'), 'logvar': t('logvar'), 'scores': t('scores')}
    np.testing.assert_allclose(loss.MaskedMSELoss()(out, t('gt'), t('mask')).item(), g['mse'], rtol=2e-6)
    np.testing.assert_allclose(loss.MaskedBadPix()(out, t('gt'), t('mask')).item(), g['badpix'], rtol=1e-6)
    np.testing.assert_allclose(loss.MaskedL1Loss()(out, t('gt'), t('mask')).item(), g['l1'], rtol=2e-6)
    np.testing.assert_array_equal(loss.create_mask_margin((2, 40, 44), 15).numpy(), g['margin_15'])
    zero = torch.zeros_like(t('mask'))
    assert loss.MaskedBadPix()(out, t('gt'), zero).item() == 0


@pytest.mark.gpu
def test_dataparallel_replicas_on_the_native_path():
    """reference train/cli.py:159 wraps the model in nn.DataParallel.  With more than one device id DataParallel runs
    REPLICAS (torch.nn.parallel.replicate) from one thread per device: a replica has no registered parameters (its
    copies hang off `_former_parameters`), shares the module's Python attributes, and is called concurrently.  Two
    replicas on the one GPU of the test box: the gathered output and the reduced gradients must equal the
    un-wrapped module run on the two half batches."""
    kw = dict(TINY_KW, model_uncert=True)
    state = synth.synth_state(synth.param_spec(**kw), seed=5)
    stacks, gt, mask = synth.synth_inputs(4, 16, seed=3)
    dev = 'cuda:0'
    data = [torch.from_numpy(s).to(dev) for s in stacks]
    tgt, tmask = torch.from_numpy(gt).to(dev), torch.from_numpy(mask).to(dev)
    m = _fresh(kw, state, dev)
    try:
        dp = torch.nn.DataParallel(m, device_ids=[0, 0])
        dp.train()
        out = dp(*data)
    except (RuntimeError, AssertionError, ValueError) as e:          # torch refusing duplicate device ids
        pytest.skip(f'DataParallel(device_ids=[0, 0]) not possible here: {e}')
    l = loss.ImprovedUncertaintyL1Loss()(out, tgt, tmask, None)
    l.backward()
    ref = _fresh(kw, state, dev)
    ref.train()
    outs = [ref(*[d[lo:lo + 2].contiguous() for d in data]) for lo in (0, 2)]
    mean = torch.cat([o['mean'] for o in outs])
    np.testing.assert_allclose(out['mean'].detach().cpu().numpy(), mean.detach().cpu().numpy(), rtol=1e-5, atol=1e-6)
    l2 = loss.ImprovedUncertaintyL1Loss()({'mean': mean, 'logvar': torch.cat([o['logvar'] for o in outs])}, tgt, tmask, None)
    l2.backward()
    np.testing.assert_allclose(float(l), float(l2), rtol=1e-6)
    for (n, p), (_, q) in zip(m.named_parameters(), ref.named_parameters()):
        scale = max(float(q.grad.abs().max()), 1e-6)
        assert float((p.grad - q.grad).abs().max()) <= 2e-4 * scale + 1e-7, n


@pytest.mark.gpu
def test_dataparallel_forwards_do_not_accumulate_workspaces():
    """DataParallel.parallel_apply starts fresh threads on every forward; the engine's per-thread workspaces
    (partial-sum buffer, side stream, weight-gradient scratch) must die with them: 50 forward/backward passes leave
    the caching allocator where the first few put it."""
    import gc
    import threading
    from mmlf_amd import engine
    kw = dict(TINY_KW, model_uncert=True)
    state = synth.synth_state(synth.param_spec(**kw), seed=5)
    stacks, gt, mask = synth.synth_inputs(4, 16, seed=3)
    dev = 'cuda:0'
    data = [torch.from_numpy(s).to(dev) for s in stacks]
    tgt, tmask = torch.from_numpy(gt).to(dev), torch.from_numpy(mask).to(dev)
    m = _fresh(kw, state, dev)
    try:
        dp = torch.nn.DataParallel(m, device_ids=[0, 0])
        dp.train()
        dp(*data)
    except (RuntimeError, AssertionError, ValueError) as e:
        pytest.skip(f'DataParallel(device_ids=[0, 0]) not possible here: {e}')
    seen = set()
    orig = engine._Workspace.__init__

    def counting(self, device):
        seen.add(threading.get_ident())
        orig(self, device)

    engine._Workspace.__init__ = counting
    try:
        def one():
            out = dp(*data)
            loss.ImprovedUncertaintyL1Loss()(out, tgt, tmask, None).backward()
            m.zero_grad(set_to_none=True)

        for _ in range(5):
            one()
        torch.cuda.synchronize(); gc.collect()
        reserved, allocated = torch.cuda.memory_reserved(), torch.cuda.memory_allocated()
        for _ in range(50):
            one()
        torch.cuda.synchronize(); gc.collect()
    finally:
        engine._Workspace.__init__ = orig
    assert len(seen) >= 1                                   # workspaces were (re)created by worker threads
    assert torch.cuda.memory_allocated() <= allocated + (1 << 20), (allocated, torch.cuda.memory_allocated())
    assert torch.cuda.memory_reserved() <= reserved + (32 << 20), (reserved, torch.cuda.memory_reserved())
    assert not hasattr(engine._Workspace, '_cache')         # no process-lifetime table of per-thread entries
