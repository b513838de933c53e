"""GPU parity tests, model level: FeedForward on the HIP path against the golden vectors produced
by the reference (tests/golden) and against the CPU oracle.  Tolerances are float32 tolerances
(BASELINE.json north_star: per-pixel depth MAE <= 1e-4); they are written next to each check."""
import numpy as np
import pytest
import torch

from conftest import BASE_KW, TINY_KW, VARIANTS, load_golden
from mmlf_amd import synth

pytestmark = pytest.mark.gpu
# G2 gradient bars, relative L2 per parameter tensor against the reference's float32 run.  End-to-end gradients are
# ill-conditioned: the reference's own float32 and float64 runs differ by 0.6-0.9 % per tensor.  The BASE
# input additionally has ONE head unit whose ReLU flips between implementations (tools/mode_diverge.py): +1.6-1.9 % on every
# tensor below the head; measured worst tensors 2.44 % (in_net_id.1.3.weight), median 1.7 %.  UPR / DPP have no such unit:
# worst 0.98 % / 1.08 % (in_net_id.2.3.bias / in_net_hv.1.0.bias), medians 0.7 % / 0.9 %.  (Round 3 used 3 % for all.)
G2_GRAD_BAR = {'base': 2.6e-2, 'upr': 1.2e-2, 'dpp': 1.3e-2}
DEPTH_MAE_TOL = 1e-4


def _dev():
    assert torch.cuda.is_available()
    return torch.device('cuda:0')


def _model(kw, state):
    from mmlf_amd.feed_forward import FeedForward
    m = FeedForward(**kw)
    m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in state.items()})
    return m.to(_dev())


def _state(g):
    return {k[len('state/'):]: v for k, v in g.items() if k.startswith('state/')}


def _loss_fn(variant):
    from mmlf_amd import loss, dl
    if variant == 'upr':
        return lambda out, gt, mask: loss.ImprovedUncertaintyL1Loss()(out, gt, mask, None)
    if variant == 'dpp':
        return lambda out, gt, mask: loss.MaskedCrossEntropy()(out, dl.reg_to_class(gt, -3.5, 3.5, 108), mask)
    return lambda out, gt, mask: loss.MaskedL1Loss()(out, gt, mask)


def test_native_library_is_loaded_and_required(monkeypatch):
    from mmlf_amd import _lib
    _lib.load()
    maps = open('/proc/self/maps').read()
    assert 'libmmlf_hip.so' in maps
    # the GPU path must fail loudly without the library
    monkeypatch.setattr(_lib, '_lib', None)
    monkeypatch.setattr(_lib, 'LIB_PATH', '/nonexistent/libmmlf_hip.so')
    g = load_golden('g1_tiny_base.npz')
    m = _model(TINY_KW, _state(g))
    with pytest.raises(RuntimeError, match='no CPU fallback'):
        m(*[torch.from_numpy(g[f'in{i}']).to(_dev()) for i in range(4)])


@pytest.mark.parametrize('variant', list(VARIANTS))
def test_g1_tiny_forward_backward_vs_reference(variant):
    g = load_golden(f'g1_tiny_{variant}.npz')
    kw = dict(TINY_KW, **VARIANTS[variant])
    dev = _dev()
    m = _model(kw, _state(g))
    stacks = [torch.from_numpy(g[f'in{i}']).to(dev) for i in range(4)]
    m.eval()
    with torch.no_grad():
        out = m(*stacks)
    for k, v in out.items():
        if v is None:
            assert f'eval_{k}' not in g
        else:
            np.testing.assert_allclose(v.cpu().numpy(), g[f'eval_{k}'], rtol=5e-5, atol=5e-6, err_msg=f'eval {k}')
    m.train()
    out = m(*stacks)
    for k, v in out.items():
        if v is not None and k != 'one_hot':
            np.testing.assert_allclose(v.detach().cpu().numpy(), g[f'train_{k}'], rtol=1e-4, atol=1e-5, err_msg=f'train {k}')
    loss = _loss_fn(variant)(out, torch.from_numpy(g['gt']).to(dev), torch.from_numpy(g['mask']).to(dev))
    np.testing.assert_allclose(loss.item(), g['loss'], rtol=2e-5)
    loss.backward()
    for n, p in m.named_parameters():
        ref = g[f'grad/{n}']
        scale = max(np.abs(ref).max(), 1e-6)
        err = np.abs(p.grad.cpu().numpy() - ref).max()
        assert err <= 5e-4 * scale + 5e-7, (n, err, scale)
    sd = m.state_dict()
    for k, v in sd.items():
        if 'running' in k:
            np.testing.assert_allclose(v.cpu().numpy(), g[f'post/{k}'], rtol=1e-5, atol=1e-6, err_msg=k)
        if 'num_batches' in k:   # stream nets are called twice per forward (H then V, I then D)
            assert int(v) == int(g[f'post/{k}']), k


@pytest.mark.parametrize('variant', list(VARIANTS))
def test_g2_full_size_eval_depth_mae(variant):
    g = load_golden(f'g2_full_{variant}.npz')
    kw = dict(BASE_KW, **VARIANTS[variant])
    m = _model(kw, synth.synth_state(synth.param_spec(**kw), seed=21))
    stacks, _, _ = synth.synth_inputs(1, 96, seed=7)
    m.eval()
    with torch.no_grad():
        out = m(*[torch.from_numpy(s).to(_dev()) for s in stacks])
    mean = out['mean'].cpu().numpy()
    if variant != 'dpp':
        mae = np.abs(mean - g['eval_mean']).mean()
        assert mae <= DEPTH_MAE_TOL, mae
        assert np.abs(mean - g['eval_mean']).max() <= 1e-3
    if variant == 'upr':
        assert np.abs(out['logvar'].cpu().numpy() - g['eval_logvar']).mean() <= DEPTH_MAE_TOL
        np.testing.assert_allclose(out['posterior'].cpu().numpy()[:, :, ::8, ::8], g['eval_posterior_s'], rtol=5e-3, atol=1e-6)
    if variant == 'dpp':
        sc = out['scores'].cpu().numpy()
        np.testing.assert_allclose(sc[:, :, ::8, ::8], g['eval_scores_s'], rtol=2e-4, atol=5e-5)
        flips = (sc.argmax(1) != g['eval_argmax']).mean()
        assert flips <= 0.0015, flips                      # one flipped bin moves a pixel by 7/107
        assert np.abs(mean - g['eval_mean']).mean() <= DEPTH_MAE_TOL
        ok = np.isfinite(g['eval_logvar'])
        assert np.abs(out['logvar'].cpu().numpy() - g['eval_logvar'])[ok].mean() <= 1e-3


@pytest.mark.parametrize('variant', ['base', 'upr', 'dpp'])
def test_g2_full_size_train_step_vs_reference(variant):
    from mmlf_amd.loss import create_mask_margin
    g = load_golden('g2_full_dpp_train.npz' if variant == 'dpp' else f'g2_full_{variant}.npz')
    kw = dict(BASE_KW, **VARIANTS[variant])
    dev = _dev()
    m = _model(kw, synth.synth_state(synth.param_spec(**kw), seed=21))
    stacks, gt, mask = synth.synth_inputs(2, 96, seed=8)
    mask = torch.from_numpy(mask).int() * create_mask_margin(mask.shape, 11)
    m.train()
    out = m(*[torch.from_numpy(s).to(dev) for s in stacks])
    if variant == 'dpp':     # the 280->108 and 108->108 head convolutions inside the real net, forward and backward
        sc = out['scores'].detach().cpu().numpy()
        np.testing.assert_allclose(sc[:, :, ::8, ::8], g['train_scores_s'], rtol=2e-4, atol=5e-5)
        assert (sc.argmax(1) != g['train_argmax']).mean() <= 0.0015
    else:
        assert np.abs(out['mean'].detach().cpu().numpy() - g['train_mean']).mean() <= DEPTH_MAE_TOL
    loss = _loss_fn(variant)(out, torch.from_numpy(gt).to(dev), mask.to(dev))
    np.testing.assert_allclose(loss.item(), g['loss'], rtol=1e-4)
    loss.backward()
    # end-to-end gradients are ill-conditioned (the reference's own fp32 vs fp64 runs differ by 0.6 % here, and this
    # input has a head unit whose ReLU flips between implementations: +1.6 %): G2_GRAD_BAR per variant is the bar
    # HERE; test_conditioned_train_step_gradients_at_float32_level holds the tight one
    ratios = {}
    for n, p in m.named_parameters():
        ref = g[f'grad_s/{n}']
        got = p.grad.cpu().numpy()
        got = got.reshape(-1)[::97] if got.size > 4096 else got
        ratios[n] = np.linalg.norm(got - ref) / (np.linalg.norm(ref) + 5e-6 / 3e-2)   # floor: biases feeding BN have zero true gradient
    worst = sorted(ratios.items(), key=lambda kv: -kv[1])[:3]
    print(f'G2 {variant}: gradient relative L2 per tensor: median {np.median(list(ratios.values())):.4f}, worst {worst}')
    for n, r in ratios.items():
        assert r <= G2_GRAD_BAR[variant], (n, r)
    for k, v in m.state_dict().items():
        if 'running' in k:
            np.testing.assert_allclose(v.cpu().numpy(), g[f'post/{k}'], rtol=5e-5, atol=2e-6, err_msg=k)


@pytest.mark.parametrize('variant', ['base', 'upr'])
@pytest.mark.parametrize('mode', ['f16x3', 'bf16x6', 'f32'])
def test_conditioned_train_step_gradients_at_float32_level(variant, mode, monkeypatch):
    """How far may a float32 implementation's gradients be from the truth?  The gradient of this net is discontinuous
    where few units carry much of it: ONE ReLU flip in the head's one-channel first convolution (a pre-activation
    within rounding noise of zero) moves every gradient below it by 1.6 % -- that, not arithmetic, is what the 2.6 %
    BASE bar of the G2 test above absorbs (measured: tools/mode_diverge.py finds exactly one such unit in G2's input).
    tests/golden/g11_conditioned_*.npz is a train step whose head pre-activations and L1 signs all stay 1e-4 (20 x the
    noise) away from flipping, with the reference's float32 AND float64 runs.  Yardstick = the reference's own float32
    distance from its float64 run, per parameter tensor.  On this fixture torch's own GPU kernels (MIOpen / ATen, the
    ops the reference executes on a GPU) measure median 1.25-1.49, 90th percentile 1.6-1.7, worst tensor 2.0-2.9 of
    that yardstick (tools/grad_yardstick.py); every arithmetic mode of this library must stay inside the same band,
    and within 1.5 % of the float64 gradient per tensor (the float32 runs themselves sit at 0.6-0.9 %: the many small
    ReLU flips of the wide layers)."""
    from mmlf_amd import engine
    from mmlf_amd.loss import create_mask_margin
    monkeypatch.setattr(engine, 'CONV_MODE', mode)
    g = load_golden(f'g11_conditioned_{variant}.npz')
    g32 = {k[4:]: v for k, v in g.items() if k.startswith('f32/')}
    g64 = {k[4:]: v for k, v in g.items() if k.startswith('f64/')}
    kw = dict(BASE_KW, **VARIANTS[variant])
    dev = _dev()
    m = _model(kw, synth.synth_state(synth.param_spec(**kw), seed=21))
    stacks, gt, mask = synth.synth_inputs(2, 96, seed=int(g['seed']))
    mask = torch.from_numpy(mask).int() * create_mask_margin(mask.shape, 11)
    m.train()
    out = m(*[torch.from_numpy(s).to(dev) for s in stacks])
    loss = _loss_fn(variant)(out, torch.from_numpy(gt).to(dev), mask.to(dev))
    loss.backward()
    # forward: depth (and log-variance) against float64, no farther than 2 x the reference's float32 run
    for key in ('mean', 'logvar') if variant == 'upr' else ('mean',):
        e_hip = np.abs(out[key].detach().cpu().numpy().astype(np.float64) - g64[f'train_{key}']).mean()
        e_ref = np.abs(g32[f'train_{key}'].astype(np.float64) - g64[f'train_{key}']).mean()
        assert e_hip <= 2.0 * e_ref + 1e-7, (key, e_hip, e_ref)
    assert abs(loss.item() - float(g64['loss'])) <= 2e-6 * abs(float(g64['loss']))
    ratios, gnorm = [], max(np.linalg.norm(g64[k]) for k in g64 if k.startswith('grad_s/'))
    for n, p in m.named_parameters():
        ref64 = g64[f'grad_s/{n}']
        got = p.grad.cpu().numpy().astype(np.float64)
        got = got.reshape(-1)[::97] if got.size > 4096 else got
        d_hip = np.linalg.norm(got - ref64)
        d_ref = np.linalg.norm(g32[f'grad_s/{n}'].astype(np.float64) - ref64)
        if n.endswith('.2.bias') and '.7.' not in n:
            # a conv bias in front of BatchNorm: the true gradient is exactly 0; every float32 run holds noise there
            assert np.linalg.norm(ref64) <= 1e-9 * gnorm and d_hip <= 4.0 * d_ref + 1e-6 * gnorm, (n, d_hip, d_ref)
            continue
        assert d_hip <= 4.0 * d_ref + 1e-7 * gnorm, (n, d_hip / max(d_ref, 1e-30))
        assert d_hip <= 1.5e-2 * np.linalg.norm(ref64) + 1e-7 * gnorm, (n, d_hip / np.linalg.norm(ref64))
        ratios.append(d_hip / max(d_ref, 1e-30))
    assert np.median(ratios) <= 1.6 and np.percentile(ratios, 90) <= 2.0, (np.median(ratios), np.percentile(ratios, 90))


@pytest.mark.parametrize('W', [125, 126, 200])
def test_train_and_eval_across_the_narrow_kernel_switch(W):
    """The 27 -> 70 / 70 -> 70 layers run the tiled sixteen-wave kernel up to a pitch of 127 positions (W <= 125) and the
    register-streamed kernel above it (full frames): the full-width net on both sides of the switch -- eval forward, and a
    train-mode forward + backward with every parameter gradient -- against the stock torch ops of the same module."""
    kw = dict(BASE_KW, model_uncert=True)
    state = synth.synth_state(synth.param_spec(**kw), seed=31)
    dev = _dev()
    gen = torch.Generator(device=dev).manual_seed(W)
    B, H = 2, 12
    stacks = [torch.rand((B, 9, 3, H, W), device=dev, generator=gen) for _ in range(4)]
    gt = 4.0 * torch.rand((B, H, W), device=dev, generator=gen) - 2.0
    res = {}
    for path in ('native', 'torch'):
        m = _model(kw, state)
        if path == 'torch':
            m._native_ok = False
        m.eval()
        with torch.no_grad():
            ev = m(*stacks)
        m.train()
        out = m(*stacks)
        lossv = (torch.exp(-out['logvar']) * torch.abs(out['mean'] - gt) + out['logvar']).mean()
        lossv.backward()
        res[path] = (ev['mean'].cpu(), ev['logvar'].cpu(), float(lossv), {n: p.grad.cpu() for n, p in m.named_parameters()})
    a, b = res['native'], res['torch']
    assert float((a[0] - b[0]).abs().mean()) <= DEPTH_MAE_TOL and float((a[1] - b[1]).abs().mean()) <= DEPTH_MAE_TOL
    np.testing.assert_allclose(a[2], b[2], rtol=1e-4)
    floor = 1e-4 * max(float(g.norm()) for g in b[3].values())
    for n, ref in b[3].items():
        if n.endswith('.2.bias') and not n.startswith('out_net.7.'):
            continue        # conv bias in front of BatchNorm: true-zero gradient
        assert float((a[3][n] - ref).norm()) <= 3e-2 * float(ref.norm()) + floor, n


def test_dpp_with_eleven_views_runs_natively():
    """--model_views 11 gives the DPP head 4*11*3 = 132 channels (channel stride 136): forward, the NCHW pack of its
    gradient (a transpose tile that must not depend on the channel count) and backward against the stock torch ops"""
    from mmlf_amd import dl, loss
    kw = dict(TINY_KW, model_views=11, model_discrete=True)
    state = synth.synth_state(synth.param_spec(**kw), seed=6)
    stacks, gt, mask = synth.synth_inputs(2, 16, views=11, seed=6)
    dev = _dev()
    t = [torch.from_numpy(s).to(dev) for s in stacks]
    res = {}
    for path in ('native', 'torch'):
        m = _model(kw, state)
        assert m._native_ok and m.steps == 132
        m.train()
        if path == 'torch':
            m._native_ok = False
        out = m(*t)
        l = loss.MaskedCrossEntropy()(out, dl.reg_to_class(torch.from_numpy(gt).to(dev), -3.5, 3.5, 132),
                                      torch.from_numpy(mask).to(dev))
        l.backward()
        res[path] = (out['scores'].detach().cpu(), float(l), {n: p.grad.cpu() for n, p in m.named_parameters()})
    a, b = res['native'], res['torch']
    torch.testing.assert_close(a[0], b[0], rtol=2e-4, atol=2e-5)
    np.testing.assert_allclose(a[1], b[1], rtol=1e-5)
    floor = 1e-4 * max(float(g.norm()) for g in b[2].values())
    for n, ref in b[2].items():
        assert float((a[2][n] - ref).norm()) <= 2e-3 * float(ref.norm()) + floor, n


def test_full_size_vs_oracle_other_seed(oracle):
    """Same check against the CPU oracle itself on inputs no golden covers (B=3, 40x56 patch)."""
    kw = dict(BASE_KW, model_uncert=True)
    state = synth.synth_state(synth.param_spec(**kw), seed=5)
    stacks, gt, mask = synth.synth_inputs(3, 40, seed=2, ps_w=56)
    net = oracle.OracleNet(kw, state)
    ref = net.forward(*stacks, train=True)
    m = _model(kw, state)
    m.train()
    out = m(*[torch.from_numpy(s).to(_dev()) for s in stacks])
    assert np.abs(out['mean'].detach().cpu().numpy() - ref['mean']).mean() <= DEPTH_MAE_TOL
    assert np.abs(out['logvar'].detach().cpu().numpy() - ref['logvar']).mean() <= DEPTH_MAE_TOL


def test_halo_11_tile_invariance_eval():
    """Whole-net receptive radius is 11 px: an interior crop with an 11-px halo reproduces the
    full-frame result."""
    m = _model(BASE_KW, synth.synth_state(synth.param_spec(**BASE_KW), seed=3))
    m.eval()
    stacks, _, _ = synth.synth_inputs(1, 64, seed=4)
    ts = [torch.from_numpy(s).to(_dev()) for s in stacks]
    with torch.no_grad():
        full = m(*ts)['mean']
        crop = m(*[t[..., 8:56, 8:56].contiguous() for t in ts])['mean']
    np.testing.assert_allclose(crop[:, 11:-11, 11:-11].cpu().numpy(), full[:, 19:45, 19:45].cpu().numpy(),
                               rtol=1e-5, atol=1e-6)


def test_reference_train_loop_shape_dataparallel_adam_autograd():
    """The reference's own loop body (train/cli.py:159,243-258): DataParallel wrap, torch.optim.Adam,
    loss.backward() through autograd -- must give the same update as the native TrainStep."""
    from mmlf_amd import loss
    from mmlf_amd.train import TrainStep
    g = load_golden('g1_tiny_upr.npz')
    kw = dict(TINY_KW, model_uncert=True)
    dev = _dev()
    stacks = [torch.from_numpy(g[f'in{i}']).to(dev) for i in range(4)]
    gt, mask = torch.from_numpy(g['gt']).to(dev), torch.from_numpy(g['mask']).to(dev)
    m1 = _model(kw, _state(g))
    dp = torch.nn.DataParallel(m1)
    opt = torch.optim.Adam(dp.parameters(), lr=1e-3)
    dp.train()
    opt.zero_grad()
    out = dp(*stacks)
    l1 = loss.ImprovedUncertaintyL1Loss()(out, gt, mask, None)
    l1.backward()
    opt.step()
    m2 = _model(kw, _state(g))
    st = TrainStep(m2, lr=1e-3, loss_margin=0)
    l2 = st(*stacks, gt, mask, 1)
    np.testing.assert_allclose(float(l1), float(l2), rtol=1e-6)
    np.testing.assert_allclose(float(l1), g['loss'], rtol=2e-5)
    for (k, a), (_, b) in zip(dp.module.state_dict().items(), m2.state_dict().items()):
        ref_g = g.get(f'grad/{k}')
        if ref_g is not None:      # compare where the gradient is far above rounding noise
            solid = torch.from_numpy(np.abs(ref_g) > 1e-5)
            torch.testing.assert_close(a.cpu()[solid], b.cpu()[solid], rtol=0, atol=3e-6, msg=k)
            np.testing.assert_allclose(a.cpu().numpy()[solid.numpy()], g[f'post/{k}'][solid.numpy()], rtol=0, atol=3e-6, err_msg=k)
        else:
            torch.testing.assert_close(a, b, rtol=1e-5, atol=1e-6, msg=k)


def test_full_frame_512_equals_halo_tiles():
    """BASELINE.json configs[4] size: a 512x512 frame through the un-tiled eval forward equals 11-px-halo
    tiles of it (size-independent property; the two runs use different grid pitches / window modes)."""
    kw = dict(BASE_KW, model_uncert=True)
    m = _model(kw, synth.synth_state(synth.param_spec(**kw), seed=13))
    m.eval()
    stacks, _, _ = synth.synth_inputs(1, 512, seed=14)
    ts = [torch.from_numpy(s).to(_dev()) for s in stacks]
    with torch.no_grad():
        full = m(*ts)
        y0, x0, sz = 120, 300, 150                       # interior tile, odd size
        tile = m(*[t[..., y0 - 11:y0 + sz + 11, x0 - 11:x0 + sz + 11].contiguous() for t in ts])
    for k in ('mean', 'logvar'):
        a = tile[k][:, 11:-11, 11:-11].cpu().numpy()
        b = full[k][:, y0:y0 + sz, x0:x0 + sz].cpu().numpy()
        np.testing.assert_allclose(a, b, rtol=2e-5, atol=2e-6, err_msg=k)
    assert np.isfinite(full['posterior'].cpu().numpy()).all()


def test_train_step_is_permutation_invariant_over_the_batch():
    """BN statistics, the masked-mean loss and every gradient are sums over the batch: shuffling the
    patches of a 96-patch ps=96 batch must not change the loss or the gradients (beyond summation order)."""
    from mmlf_amd.train import TrainStep
    dev = _dev()
    state = synth.synth_state(synth.param_spec(**BASE_KW), seed=17)
    stacks, gt, mask = synth.synth_inputs(96, 96, seed=18)
    perm = np.random.RandomState(0).permutation(96)
    res = []
    for order in (np.arange(96), perm):
        st = TrainStep(_model(BASE_KW, state), lr=1e-3)
        data = [torch.from_numpy(s[order]).to(dev) for s in stacks]
        loss = st(*data, torch.from_numpy(gt[order]).to(dev), torch.from_numpy(mask[order]).to(dev), 1)
        res.append((float(loss), st.grad.clone()))
    np.testing.assert_allclose(res[0][0], res[1][0], rtol=1e-5)
    g0, g1 = res[0][1], res[1][1]
    assert float((g0 - g1).norm()) <= 2e-2 * float(g0.norm())     # ReLU/sign flips: see DESIGN.md section 2


@pytest.mark.parametrize('variant', ['base', 'dpp'])
def test_producers_keep_every_conv_operand_absmax_exact(variant, monkeypatch):
    """f16x3 scales every conv / wgrad operand by a power of two derived from the tensor's max |x|, which
    the PRODUCING kernel maintains (conv epilogue, BN apply, BN backward, NCHW pack).  With the engine's
    check hook on, a full train step and an eval forward verify at every consumer that the slot equals the
    tensor's true maximum (a stale or missing slot would mean a wrong scale: overflow or lost bits)."""
    from mmlf_amd import engine
    from mmlf_amd.train import TrainStep
    monkeypatch.setattr(engine, 'CONV_MODE', 'f16x3')
    monkeypatch.setattr(engine, 'CHECK_ABSMAX', True)
    dev = _dev()
    kw = dict(BASE_KW, **VARIANTS[variant])
    m = _model(kw, synth.synth_state(synth.param_spec(**kw), seed=3))
    stacks, gt, mask = synth.synth_inputs(2, 32, seed=4)
    step = TrainStep(m, lr=1e-3, loss_margin=11)
    loss = step(*[torch.from_numpy(s).to(dev) for s in stacks], torch.from_numpy(gt).to(dev),
                torch.from_numpy(mask).to(dev), 1)
    assert torch.isfinite(loss)
    m.eval()
    with torch.no_grad():
        out = m(*[torch.from_numpy(s).to(dev) for s in stacks])
    assert torch.isfinite(out['mean']).all()


@pytest.mark.parametrize('scale', [0.0, 1e-2, 1.0, 3e3])
def test_f16_split_is_scale_invariant(scale, monkeypatch):
    """Inputs from all-zero to thousands: the f16-split path (power-of-two operand scaling) must track the
    exact-f32 kernels at every magnitude -- no overflow, no lost bits -- in train mode (BatchNorm renormalises)
    and through the backward pass."""
    from mmlf_amd import engine
    from mmlf_amd.loss import MaskedL1Loss
    dev = _dev()
    kw = dict(BASE_KW, model_in_blocks=2, model_out_blocks=3, model_chs=8)
    stacks, gt, mask = synth.synth_inputs(2, 24, seed=5)
    res = {}
    for mode in ('f32', 'f16x3'):
        monkeypatch.setattr(engine, 'CONV_MODE', mode)
        m = _model(kw, synth.synth_state(synth.param_spec(**kw), seed=9))
        m.train()
        out = m(*[torch.from_numpy(s * np.float32(scale)).to(dev) for s in stacks])
        loss = MaskedL1Loss()(out, torch.from_numpy(gt).to(dev), torch.from_numpy(mask).to(dev))
        loss.backward()
        res[mode] = (out['mean'].detach().cpu().numpy(), {n: p.grad.cpu().numpy() for n, p in m.named_parameters()})
    a, b = res['f32'], res['f16x3']
    assert np.isfinite(b[0]).all() and all(np.isfinite(g).all() for g in b[1].values())
    if scale == 0.0:
        return      # all-zero images: BatchNorm of a constant is degenerate, only finiteness is meaningful
    np.testing.assert_allclose(b[0], a[0], rtol=1e-3, atol=1e-4)
    floor = 1e-4 * max(np.linalg.norm(g) for g in a[1].values())
    for n in a[1]:
        if n.endswith('.2.bias') and not n.startswith('out_net.2.'):
            continue        # a conv bias in front of BatchNorm has a true-zero gradient: both modes return noise
        # end-to-end gradients are ill-conditioned: 5 % relative L2 per tensor
        assert np.linalg.norm(b[1][n] - a[1][n]) <= 5e-2 * np.linalg.norm(a[1][n]) + floor, n


def test_side_stream_weight_gradient_gives_the_same_bits(monkeypatch):
    """MMLF_OVERLAP_WGRAD=1 (the default since round 6) runs the wide blocks' first weight gradient on a side stream with its own
    workspace, beside the BatchNorm-backward kernels of the block underneath: the gradients must be bit-identical to the
    single-stream order, step after step."""
    from mmlf_amd import engine
    from mmlf_amd.train import TrainStep
    state = synth.synth_state(synth.param_spec(**BASE_KW), seed=9)
    stacks, gt, mask = synth.synth_inputs(2, 32, seed=5)
    dev = _dev()
    t = [torch.from_numpy(s).to(dev) for s in stacks]
    res = {}
    for overlap in (False, True):
        monkeypatch.setattr(engine, 'OVERLAP_WGRAD', overlap)
        step = TrainStep(_model(BASE_KW, state), lr=1e-3)
        for it in range(3):
            step(*t, torch.from_numpy(gt).to(dev), torch.from_numpy(mask).to(dev), it + 1)
        torch.cuda.synchronize()
        res[overlap] = (step.grad.clone(), step.flat.clone())
    assert torch.equal(res[False][0], res[True][0]) and torch.equal(res[False][1], res[True][1])
    assert torch.isfinite(res[True][0]).all() and float(res[True][0].abs().max()) > 0
