"""SURVEY section 8 'next' rows: f4 multimodal targets/losses, f3 checkpoint format / resume,
f2 validate-side distribution metrics -- against goldens generated from the reference."""
import numpy as np
import pytest
import torch

from conftest import TINY_KW, load_golden
from mmlf_amd import dl, loss, metrics, synth
from mmlf_amd.feed_forward import FeedForward
from mmlf_amd.train import TrainStep

DEVICES = ['cpu', pytest.param('cuda', marks=pytest.mark.gpu)]


@pytest.mark.parametrize('dev', DEVICES)
def test_multimodal_targets_and_losses(dev):
    g = load_golden('g6_multimodal.npz')
    t = lambda k: torch.from_numpy(g[k]).to(dev)
    np.testing.assert_allclose(dl.mpi_to_weights(t('mpi'), -3.5, 3.5, 108).cpu().numpy(), g['mpi_to_weights'],
                               rtol=1e-6, atol=1e-7)
    for name, fn, keys in (('multi_l1', loss.MultiMaskedL1Loss(), ['mean']),
                           ('multi_upr', loss.ImprovedMultiUncertaintyL1Loss(), ['mean', 'logvar'])):
        o = {'mean': t('mean').requires_grad_(True), 'logvar': t('logvar').requires_grad_(True)}
        val = fn(o, t('mpi'), t('mask'))
        val.backward()
        np.testing.assert_allclose(val.item(), g[name], rtol=2e-6)
        for k in keys:
            np.testing.assert_allclose(o[k].grad.cpu().numpy(), g[f'd{name}_d{k}'], rtol=2e-5, atol=1e-9)


@pytest.mark.parametrize('dev', DEVICES)
def test_validate_distribution_metrics(dev):
    g = load_golden('g6_multimodal.npz')
    t = lambda k: torch.from_numpy(g[k]).to(dev)
    mean, logvar, mpi = t('mean')[:1], t('logvar')[:1], t('mpi')[:1]
    l2d = metrics.laplace_to_discrete(108, -3.5, 3.5, mean, logvar)
    np.testing.assert_allclose(l2d.cpu().numpy(), g['laplace_to_discrete'], rtol=1e-6, atol=1e-12)
    lmm = metrics.lmm_to_discrete(108, -3.5, 3.5, t('lmm_means'), torch.exp(t('lmm_logvars')))
    np.testing.assert_allclose(lmm.cpu().numpy(), g['lmm_to_discrete'], rtol=1e-6, atol=1e-12)
    np.testing.assert_array_equal(metrics.mean_to_discrete(108, -3.5, 3.5, mean).cpu().numpy(), g['mean_to_discrete'])
    mm = metrics.multimodal_mask(mpi)
    np.testing.assert_array_equal(mm.cpu().numpy(), g['multimodal_mask'])
    gt = dl.mpi_to_weights(mpi, -3.5, 3.5, 108)      # float32 like the reference's dist_gt
    np.testing.assert_allclose(metrics.kl_divergence(l2d, gt).item(), g['kld'], rtol=1e-5)   # dist_gt is float32 in the reference
    np.testing.assert_allclose(metrics.kl_divergence(l2d, gt, mm).item(), g['kld_masked'], rtol=1e-5)
    np.testing.assert_allclose(metrics.nll_laplace(mpi, mean, logvar).item(), g['nll_laplace'], rtol=1e-5)


def test_checkpoint_format_and_resume(tmp_path):
    """f3: the file ModelSaver writes has the reference's keys, loads into torch.optim.Adam, and a
    resumed TrainStep continues exactly like an uninterrupted one."""
    kw = dict(TINY_KW, train_lr=1e-3, model_radius=5)
    state = synth.synth_state(synth.param_spec(**TINY_KW), 4)

    def fresh():
        m = FeedForward(**kw)
        m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in state.items()})
        return m

    stacks, gt, mask = synth.synth_inputs(2, 16, seed=2)
    data = [torch.from_numpy(s) for s in stacks] + [torch.from_numpy(gt), torch.from_numpy(mask)]
    a = TrainStep(fresh(), lr=1e-3, loss_margin=3)
    for it in (1, 2):
        a(*data, it)
    path = str(tmp_path / 'checkpoint.pt')
    dl.ModelSaver(only_best=False)(path, a.model, a, kw, None, 2, 0.5)
    ck = torch.load(path)
    assert set(ck) == {'model_state_dict', 'optimizer_state_dict', 'hyper_parameters', 'epoch', 'iteration', 'loss'}
    assert ck['iteration'] == 2 and ck['epoch'] is None and ck['hyper_parameters']['model_radius'] == 5
    assert list(ck['model_state_dict']) == list(fresh().state_dict())
    # the optimizer part is a valid torch.optim.Adam state dict
    m = fresh()
    opt = torch.optim.Adam(m.parameters(), lr=1e-5)
    it, _ = dl.load_checkpoint(path, m, opt, lr=1e-3)
    assert it == 2 and opt.param_groups[0]['lr'] == 1e-3
    assert float(opt.state_dict()['state'][0]['step']) == 2.0
    # resume == continue
    b = TrainStep(fresh(), lr=1e-5, loss_margin=3)
    it, _ = dl.load_checkpoint(path, b.model, b, lr=1e-3)
    a(*data, 3)
    b(*data, 3)
    for (k, va), (_, vb) in zip(a.model.state_dict().items(), b.model.state_dict().items()):
        torch.testing.assert_close(va, vb, rtol=0, atol=0, msg=k)
    # only_best keeps the better checkpoint
    saver = dl.ModelSaver(only_best=True)
    saver(path, a.model, None, kw, None, 3, 0.4)
    saver(path, a.model, None, kw, None, 4, 0.9)
    assert torch.load(path)['iteration'] == 3


@pytest.mark.parametrize('dev', DEVICES)
@pytest.mark.parametrize('variant', ['base', 'upr'])
def test_multimodal_train_step(dev, variant):
    """--train_loss_multimodal through TrainStep equals the same loss through plain autograd."""
    g = load_golden('g6_multimodal.npz')
    kw = dict(TINY_KW, model_uncert=(variant == 'upr'))
    state = synth.synth_state(synth.param_spec(**kw), 9)

    def fresh():
        m = FeedForward(**kw)
        m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in state.items()})
        return m.to(dev)

    stacks, _, _ = synth.synth_inputs(2, 14, seed=3, ps_w=18)
    data = [torch.from_numpy(s).to(dev) for s in stacks]
    mpi, mask = torch.from_numpy(g['mpi']).to(dev), torch.from_numpy(g['mask']).to(dev)
    st = TrainStep(fresh(), lr=1e-3, loss_margin=2, loss_multimodal=True)
    l1 = st(*data, mpi, mask, 1)
    m2 = fresh()
    m2.train()
    fn = loss.ImprovedMultiUncertaintyL1Loss() if variant == 'upr' else loss.MultiMaskedL1Loss()
    l2 = fn(m2(*data), mpi, st._mask(mask))
    np.testing.assert_allclose(float(l1), float(l2.detach()), rtol=1e-5)
    l2.backward()
    for (n, o, cnt), (_, p) in zip(st.layout, m2.named_parameters()):
        ref = p.grad.reshape(-1).cpu()
        got = st.grad[o:o + cnt].cpu()
        assert float((got - ref).norm()) <= 2e-3 * float(ref.norm()) + 1e-6, n


def test_validate_cli_model_setup_and_table(tmp_path):
    """validate/cli.py:213-247,350-351: the model comes from the checkpoint's own hyper-parameters plus the CLI's overrides --
    here from the checkpoint the REFERENCE's ModelSaver wrote (tests/golden/g9_checkpoint.pt) -- and the closing table rows"""
    import os
    from mmlf_amd import validate
    from mmlf_amd.ensamble import Ensamble
    path = os.path.join(os.path.dirname(__file__), 'golden', 'g9_checkpoint.pt')
    ck = torch.load(path, map_location='cpu')
    model, kw, n_params = validate.model_from_checkpoint(path, device='cpu')
    assert isinstance(model, FeedForward) and kw['model_chs'] == ck['hyper_parameters']['model_chs']
    assert kw['val_disp_min'] == -3.5 and kw['train_shift'] == 0.0 and kw['model_discrete'] is False
    for k, v in model.state_dict().items():
        assert torch.equal(v, ck['model_state_dict'][k]), k
    assert n_params == sum(v.numel() for k, v in ck['model_state_dict'].items() if 'running' not in k and 'num_batches' not in k)
    ens, _, n2 = validate.model_from_checkpoint(path, val_ensamble=True, val_disp_step=0.5, device='cpu')
    assert isinstance(ens, Ensamble) and len(ens.members()) == 14 and n2 == n_params
    head, row = validate.table_rows({'mse': 1.23456, 'badpix': 2.0, 'kld_um': 0.1, 'kld_mm': 0.25, 'kld': 0.3}, 0.4444)
    assert head == 'MSE & BadPix007 & KLD_UM & KLD_MM & KLD & - & TIME ' + 2 * chr(92)
    assert row == '1.235 & 2.000 & 0.100 & 0.250 & 0.300 & - & 0.444 ' + 2 * chr(92)


@pytest.mark.parametrize('kind', ['base', 'upr', 'dpp'])
@pytest.mark.parametrize('multimodal', [False, True])
def test_in_loop_validation_pass(kind, multimodal, tmp_path):
    """train.validation_pass == the validation block of the training loop (train/cli.py:265-318) written out with
    the loss modules: which loss each flag combination validates with, the margin mask, the averages, the files."""
    from mmlf_amd import pfm
    from mmlf_amd.train import LOG_HEADER, log_line, validation_pass
    g = load_golden('g6_multimodal.npz')
    mpi_all, rs = torch.from_numpy(g['mpi']), np.random.RandomState(4)
    B, H, W = mpi_all.shape[0], mpi_all.shape[-2], mpi_all.shape[-1]
    outs, batches = [], []
    for b in range(B):
        out = {'mean': torch.from_numpy(rs.uniform(-2, 2, (1, H, W)).astype(np.float32)),
               'logvar': torch.from_numpy(rs.uniform(-1, 1, (1, H, W)).astype(np.float32)) if kind == 'upr' else None}
        outs.append(out)
        gt = mpi_all[b:b + 1, 0, 4].contiguous()
        z = torch.zeros((1, 9, 3, H, W))
        batches.append((z, z, z, z, torch.rand(1, 3, H, W), gt, mpi_all[b:b + 1], None, torch.tensor([[b]])))

    class Seq(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.k = 0

        def forward(self, h, v, i, d):
            self.k += 1
            return dict(outs[self.k - 1])

    model = Seq()
    model.train()
    got = validation_pass(model, batches, uncert=(kind == 'upr'), loss_multimodal=multimodal, margin=3,
                          out_dir=str(tmp_path), scene_names=[f's{b}' for b in range(B)])
    assert not model.training                                     # model.eval(), :268
    # the block, literally (loss objects as train/cli.py:120-131 picks them)
    loss_fn = loss.MultiMaskedL1Loss() if multimodal else loss.MaskedL1Loss()
    loss_uncert_fn = loss.ImprovedMultiUncertaintyL1Loss() if multimodal else loss.ImprovedUncertaintyL1Loss()
    loss_val_avg = mse_avg = bad_pix_avg = 0.0
    for j, data in enumerate(batches):
        gt, mpi = data[5], data[6]
        mask = loss.create_mask_margin(gt.shape, 3)
        output = outs[j]
        if kind == 'upr':
            loss_val = loss_uncert_fn(output, mpi, mask) if multimodal else loss_uncert_fn(output, gt, mask)
        else:
            loss_val = loss_fn(output, mpi, mask) if multimodal else loss_fn(output, gt, mask)
        loss_val_avg += loss_val.item()
        mse_avg += loss.MaskedMSELoss()(output, gt, mask)
        bad_pix_avg += loss.MaskedBadPix()(output, gt, mask)
    j += 1
    np.testing.assert_allclose(got, (loss_val_avg / j, float(mse_avg) / j, float(bad_pix_avg) / j), rtol=1e-6)
    np.testing.assert_array_equal(pfm.load(str(tmp_path / 'ours' / 'disp_maps' / 's1.pfm')), outs[1]['mean'][0].numpy()[::-1])
    assert (tmp_path / 'scenes' / 's0' / 'uncert.pfm').exists() == (kind == 'upr')
    row = log_line(7, 0.5, *got, 1.25)
    assert row.startswith('      7, 0.50000000, ') and row.endswith(', 1.25000000') and len(row.split(', ')) == len(LOG_HEADER.split(','))
    with pytest.raises(ValueError):
        validation_pass(model, [], margin=3)


@pytest.mark.parametrize('dev', DEVICES)
@pytest.mark.parametrize('variant', ['base', 'dpp'])
def test_strongest_depth_train_step(dev, variant):
    """--train_loss_strongest (train/cli.py:190-192): the target is the depth of the plane with the largest alpha -- the
    step on the multi-plane tensor equals the plain step on the gathered target, and the CLI's exclusion is kept."""
    g = load_golden('g6_multimodal.npz')
    kw = dict(TINY_KW, model_discrete=(variant == 'dpp'))
    state = synth.synth_state(synth.param_spec(**kw), 9)

    def fresh():
        m = FeedForward(**kw)
        m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in state.items()})
        return m.to(dev)

    stacks, _, _ = synth.synth_inputs(2, 14, seed=3, ps_w=18)
    data = [torch.from_numpy(s).to(dev) for s in stacks]
    mpi, mask = torch.from_numpy(g['mpi']).to(dev), torch.from_numpy(g['mask']).to(dev)
    # the two lines of the reference loop, literally
    inds = torch.max(mpi[:, :, 3, :, :], dim=1)[1].unsqueeze(1)
    want_gt = torch.gather(mpi[:, :, 4, :, :], dim=1, index=inds).squeeze()
    torch.testing.assert_close(TrainStep.strongest_gt(mpi), want_gt, rtol=0, atol=0)
    assert float((want_gt - mpi[:, 0, 4]).abs().max()) > 0        # not simply the first plane
    a = TrainStep(fresh(), lr=1e-3, loss_margin=2, loss_strongest=True)
    b = TrainStep(fresh(), lr=1e-3, loss_margin=2)
    la, lb = a(*data, mpi, mask, 1), b(*data, want_gt.contiguous(), mask, 1)
    assert float(la) == float(lb)
    assert torch.equal(a.grad, b.grad) and torch.equal(a.flat, b.flat)
    with pytest.raises(AssertionError):
        TrainStep(fresh(), lr=1e-3, loss_strongest=True, loss_multimodal=True)


class _StubModel(torch.nn.Module):
    """returns fixed head outputs: pins the validation loop itself, not the network"""

    def __init__(self, out, disp_min=-3.5, disp_max=3.5, steps=108):
        super().__init__()
        self.out, self.disp_min, self.disp_max, self.steps = out, disp_min, disp_max, steps

    def forward(self, h, v, i, d):
        return dict(self.out)


@pytest.mark.parametrize('dev', DEVICES)
@pytest.mark.parametrize('compat', [True, False])
@pytest.mark.parametrize('kind', ['base', 'upr', 'dpp', 'ese'])
def test_validation_loop_vs_reference_helpers(kind, compat, dev, tmp_path):
    """reference validate/cli.py:249-351 per scene: MSE / BadPix (15-px margin), discretised predictive distribution,
    KL divergences over all / multimodal / unimodal pixels, NLL -- golden from the reference's own helper functions on
    fixed head outputs (tests/golden/g10_validate.npz) -- and the result files it writes.  compat=True: the numbers the
    reference's LOOP prints (its helpers rewrite their arguments in place and the loop passes the same arrays to
    nll_discrete and to three kl_divergence calls, validate/cli.py:313-337: golden keys `seq_*`, replayed without
    copies); compat=False: every metric from freshly discretised distributions (golden keys without `seq_`)."""
    from mmlf_amd import pfm, validate
    g = load_golden('g10_validate.npz')
    t = lambda k: torch.from_numpy(g[k]).to(dev)
    out = {'mean': t('mean'), 'logvar': None, 'scores': None, 'one_hot': None, 'posterior': None}
    if kind == 'upr':
        out['logvar'] = t('logvar')
    if kind == 'dpp':
        out.update(scores=t('scores'), posterior=t('posterior'), logvar=t('logvar'))
    if kind == 'ese':
        out = {'mean': t('mean'), 'logvar': t('logvar'), 'means': t('means'), 'logvars': t('logvars'), 'posterior': t('posterior')}
    H, W = g['gt'].shape[1:]
    views = [torch.rand(1, 3, 3, H, W) for _ in range(4)]
    scene = (*views, torch.rand(1, 3, H, W), t('gt'), t('mpi'), None, torch.tensor([[1]]))
    before = {k: v.clone() for k, v in out.items() if v is not None}
    rows, avg = validate.validate_scenes(_StubModel(out), [scene], out_dir=str(tmp_path), scene_names=['a', 'b'],
                                         reference_compat=compat)
    assert len(rows) == 1
    for k, v in before.items():                     # the in-place replay works on copies: the model's outputs are untouched
        assert torch.equal(out[k], v), k
    np.testing.assert_allclose(avg['mse'], g['mse'], rtol=1e-5)
    np.testing.assert_allclose(avg['badpix'], g['badpix'], rtol=1e-6)
    for key in ('kld', 'kld_mm', 'kld_um', 'nll'):
        np.testing.assert_allclose(avg[key], g[f'{kind}/{"seq_" if compat else ""}{key}'], rtol=2e-5, atol=1e-9, err_msg=key)
    if kind == 'dpp':                               # the files hold the ORIGINAL posterior (the reference saves before it evaluates)
        np.testing.assert_array_equal(np.load(str(tmp_path / 'scenes' / 'b' / 'posterior.npy')), g['posterior'][0])
    np.testing.assert_array_equal(pfm.load(str(tmp_path / 'scenes' / 'b' / 'result.pfm')), g['mean'][0][::-1])
    assert (tmp_path / 'ours' / 'runtimes' / 'b.txt').exists()
    if kind == 'ese':
        lmm = np.load(str(tmp_path / 'scenes' / 'b' / 'gmm.npy'))
        np.testing.assert_allclose(lmm[0], g['means'][:, 0], rtol=0, atol=0)
        np.testing.assert_allclose(lmm[1], np.exp(g['logvars'][:, 0]), rtol=1e-6)


@pytest.mark.gpu
def test_validation_loop_runs_the_ensemble_on_the_gpu(tmp_path):
    """--val_ensamble end to end on the HIP path: 70 members, fused reduce, Laplace-mixture distribution metrics, files"""
    from mmlf_amd import pfm, validate
    from mmlf_amd.ensamble import Ensamble
    kw = dict(TINY_KW, model_uncert=True)
    model = FeedForward(**kw)
    model.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in synth.synth_state(synth.param_spec(**kw), 4).items()})
    model.cuda().eval()
    ens = Ensamble(model, -3.5, 3.5, 0.1)
    scenes = []
    for k in range(2):
        s = synth.synth_scene(k, 24, 24)
        scenes.append(tuple(torch.from_numpy(np.asarray(a)).unsqueeze(0).cuda() for a in s[:8]) + (torch.tensor([[k]]),))
    rows, avg = validate.validate_scenes(ens, scenes, out_dir=str(tmp_path), scene_names=['s0', 's1'], margin=4)
    assert len(rows) == 2 and all(np.isfinite(list(r.values())).all() for r in rows)
    assert avg['kld'] > 0 and avg['mse'] > 0
    assert pfm.load(str(tmp_path / 'ours' / 'disp_maps' / 's1.pfm')).shape == (24, 24)
    assert np.load(str(tmp_path / 'scenes' / 's0' / 'gmm.npy')).shape == (2, 70, 24, 24)
    assert np.load(str(tmp_path / 'scenes' / 's0' / 'posterior.npy')).shape == (70, 24, 24)


def test_nll_discrete_modes_differ_by_the_in_place_smoothing_alone():
    """round 4's advisor finding: the two modes of metrics.nll_discrete used different dtypes (float64 weights without
    in-place, the array's own float32 with), so validate_scenes(reference_compat=True / False) returned NLLs that
    differed for a reason unrelated to the documented smoothing sequence.  Both work in the array's own dtype now, as the
    reference's numpy helper does (reference validate/cli.py:52-73): on fresh copies the two modes give the SAME number."""
    g = torch.Generator().manual_seed(3)
    weights = torch.rand((1, 108, 9, 11), generator=g)
    weights /= weights.sum(1, keepdim=True)
    post = torch.softmax(torch.randn((1, 108, 9, 11), generator=g), 1)
    w0, p0 = weights.clone(), post.clone()
    a = metrics.nll_discrete(weights, post, inplace=False)
    assert torch.equal(weights, w0) and torch.equal(post, p0) and a.dtype == torch.float32
    b = metrics.nll_discrete(weights.clone(), post.clone(), inplace=True)
    assert torch.equal(a, b), (float(a), float(b))
