#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by importing the REFERENCE
(/root/reference, read-only) on CPU in the build container.

Run (build container only -- /root/reference does not exist on the GPU box):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py

Only data is written (inputs, expected outputs); no reference source is copied.
Weights and inputs come from mmlf_amd.synth (numpy RandomState, stream-frozen), so
the large full-size cases store outputs only.

Groups:
  G1  tiny net (chs=8, 2 in-blocks, 3 out-blocks, 16x16, B=3): every tensor, BASE/UPR/DPP,
      train forward, loss, all parameter grads, BN buffers, params after one Adam step,
      eval forward.
  G2  full-size net (defaults), formula weights: train/eval outputs, loss, sampled grads.
  G4  hci4d.Shift alone and the 70-member Ensamble on a tiny UPR net.
  G5  losses and dl helpers.
  G6  multimodal losses.
  G7  the training patch pipeline (hci4d transforms as train/cli.py composes them).
  G8  round-2 additions: padded / multimodal loss gradients, DPP full-size train step, eval-mode training
      (--train_eval_mode), and a checkpoint.pt written by the reference's own ModelSaver + torch.optim.Adam
      with the state one resumed step later (g9_checkpoint.pt is a data file: tensors and hyper-parameters).
  G10 round 3: the bytes the reference's pfm.save writes (grey, colour, flipped views as HCI4D.save_batch passes
      them) and what its pfm.load reads back; --model_unet outputs.
  G11 a well-conditioned train step (no head ReLU / L1 sign within 1e-4 of flipping), float32 and float64 runs of the
      reference: the fixture for TIGHT gradient parity (see g11_conditioned_f64).
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, '/root/reference')

from mmlf.model.feed_forward import FeedForward as RefFeedForward  # noqa: E402
from mmlf.model.ensamble import Ensamble as RefEnsamble  # noqa: E402
from mmlf.model import loss as ref_loss  # noqa: E402
from mmlf.utils import dl as ref_dl  # noqa: E402
from mmlf.data.hci4d import Shift as RefShift  # noqa: E402

from mmlf_amd import synth  # noqa: E402

torch.manual_seed(0)
torch.set_num_threads(8)

BASE_KW = dict(model_ksize=2, model_in_blocks=3, model_out_blocks=8, model_chs=70,
               model_views=9, model_cross=False, model_uncert=False, model_unet=False,
               model_discrete=False, model_no_batchnorm=False,
               model_batchnorm_momentum=0.1, val_disp_min=-3.5, val_disp_max=3.5)
TINY_KW = dict(BASE_KW, model_in_blocks=2, model_out_blocks=3, model_chs=8)
VARIANTS = {
    'base': {},
    'upr': {'model_uncert': True},
    'dpp': {'model_discrete': True},
}


def build_ref(kw, seed):
    spec = synth.param_spec(**kw)
    state = synth.synth_state(spec, seed)
    model = RefFeedForward(**kw)
    ref_keys = list(model.state_dict().keys())
    assert ref_keys == [n for n, _, _ in spec], 'param_spec drifted from the reference key set'
    model.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in state.items()})
    return model, state


def train_mask(mask):
    # reference train/cli.py:194
    return torch.from_numpy(mask).int() * ref_loss.create_mask_margin(mask.shape, 11)


def loss_for(variant, out, gt, mask, kw):
    if variant == 'upr':
        return ref_loss.ImprovedUncertaintyL1Loss()(out, gt, mask, None)
    if variant == 'dpp':
        tgt = ref_dl.reg_to_class(gt, kw['val_disp_min'], kw['val_disp_max'], 108)
        return ref_loss.MaskedCrossEntropy()(out, tgt, mask)
    return ref_loss.MaskedL1Loss()(out, gt, mask)


def out_arrays(out, prefix):
    return {f'{prefix}{k}': v.detach().numpy() for k, v in out.items() if v is not None}


def g1_tiny():
    for variant, extra in VARIANTS.items():
        kw = dict(TINY_KW, **extra)
        B, ps = (2 if variant == 'dpp' else 3), 16
        model, state = build_ref(kw, seed=11)
        stacks, gt, mask = synth.synth_inputs(B, ps, seed=5)
        # margin 11 would leave nothing of a 16x16 patch: use margin 3 for the tiny case
        m = torch.from_numpy(mask).int() * ref_loss.create_mask_margin(mask.shape, 3)
        tstacks = [torch.from_numpy(s) for s in stacks]
        tgt = torch.from_numpy(gt)
        rec = {}
        # eval forward with the pristine state
        model.eval()
        with torch.no_grad():
            rec.update(out_arrays(model(*tstacks), 'eval_'))
        # train forward/backward/Adam
        model.train()
        opt = torch.optim.Adam(model.parameters(), lr=1e-3)
        opt.zero_grad()
        out = model(*tstacks)
        rec.update(out_arrays(out, 'train_'))
        loss = loss_for(variant, out, tgt, m, kw)
        loss.backward()
        rec['loss'] = loss.detach().numpy()
        for n, p in model.named_parameters():
            rec[f'grad/{n}'] = p.grad.numpy().copy()
        opt.step()
        for n, v in model.state_dict().items():
            rec[f'post/{n}'] = v.numpy().copy()
        for n, v in state.items():
            rec[f'state/{n}'] = v
        for i, s in enumerate(stacks):
            rec[f'in{i}'] = s
        rec['gt'] = gt
        rec['mask'] = m.numpy()
        np.savez_compressed(os.path.join(HERE, f'g1_tiny_{variant}.npz'), **rec)
        print('G1', variant, 'loss', float(loss), 'arrays', len(rec))


def sample(a, stride=97):
    return a.reshape(-1)[::stride].copy()


def g2_full():
    for variant, extra in VARIANTS.items():
        kw = dict(BASE_KW, **extra)
        model, state = build_ref(kw, seed=21)
        rec = {}
        # eval, B=1
        stacks, gt, mask = synth.synth_inputs(1, 96, seed=7)
        model.eval()
        with torch.no_grad():
            out = model(*[torch.from_numpy(s) for s in stacks])
        rec['eval_mean'] = out['mean'].numpy()
        if variant == 'upr':
            rec['eval_logvar'] = out['logvar'].numpy()
            rec['eval_posterior_s'] = out['posterior'].numpy()[:, :, ::8, ::8].copy()
        if variant == 'dpp':
            sc = out['scores'].numpy()
            rec['eval_scores_s'] = sc[:, :, ::8, ::8].copy()
            rec['eval_argmax'] = sc.argmax(1).astype(np.int16)
            rec['eval_logvar'] = out['logvar'].numpy()
            rec['eval_posterior_s'] = out['posterior'].numpy()[:, :, ::8, ::8].copy()
        # train, B=2 (BASE and UPR: forward + loss + sampled grads + BN buffers)
        if variant in ('base', 'upr'):
            stacks, gt, mask = synth.synth_inputs(2, 96, seed=8)
            m = train_mask(mask)
            model.train()
            model.zero_grad()
            out = model(*[torch.from_numpy(s) for s in stacks])
            loss = loss_for(variant, out, torch.from_numpy(gt), m, kw)
            loss.backward()
            rec['train_mean'] = out['mean'].detach().numpy()
            if variant == 'upr':
                rec['train_logvar'] = out['logvar'].detach().numpy()
            rec['loss'] = loss.detach().numpy()
            for n, p in model.named_parameters():
                g = p.grad.numpy()
                rec[f'grad_s/{n}'] = sample(g) if g.size > 4096 else g.copy()
            for n, v in model.state_dict().items():
                if 'running' in n or 'num_batches' in n:
                    rec[f'post/{n}'] = v.numpy().copy()
        np.savez_compressed(os.path.join(HERE, f'g2_full_{variant}.npz'), **rec)
        print('G2', variant, {k: v.shape for k, v in rec.items() if not k.startswith(('grad', 'post'))})


def g4_shift_ensamble():
    rec = {}
    rs = np.random.RandomState(3)
    stacks = [rs.uniform(size=(1, 9, 3, 12, 14)).astype(np.float32) for _ in range(4)]
    for i, s in enumerate(stacks):
        rec[f'in{i}'] = s
    for d in (-3.5, -0.3, 0.0, 0.3, 2.5, 1.0):
        data = tuple(torch.from_numpy(s.copy()) for s in stacks)
        data = RefShift(float(d))(data)
        for i in range(4):
            rec[f'shift_{d}_{i}'] = data[i].numpy()
    np.savez_compressed(os.path.join(HERE, 'g4_shift.npz'), **rec)
    print('G4 shift ok')

    kw = dict(TINY_KW, model_uncert=True)
    model, state = build_ref(kw, seed=31)
    stacks, _, _ = synth.synth_inputs(1, 24, seed=9)
    ens = RefEnsamble(model, -3.5, 3.5, 0.1)
    ens.eval()
    with torch.no_grad():
        out = ens(*[torch.from_numpy(s) for s in stacks])
    rec = {k: v.numpy() for k, v in out.items()}
    for n, v in state.items():
        rec[f'state/{n}'] = v
    for i, s in enumerate(stacks):
        rec[f'in{i}'] = s
    np.savez_compressed(os.path.join(HERE, 'g4_ensamble.npz'), **rec)
    print('G4 ensamble', {k: v.shape for k, v in out.items()})


def g5_losses():
    rs = np.random.RandomState(4)
    rec = {}
    B, H, W = 2, 20, 22
    gt = (7.4 * rs.uniform(size=(B, H, W)) - 3.7).astype(np.float32)
    gt[0, 0, 0] = -3.5 + 0.5 * 7.0 / 107.0  # exactly between two bins: hits no class
    gt[0, 0, 1] = 3.5
    gt[0, 0, 2] = -3.5
    mean = (gt + rs.normal(scale=0.1, size=gt.shape)).astype(np.float32)
    logvar = rs.normal(scale=0.5, size=gt.shape).astype(np.float32)
    scores = rs.normal(scale=2.0, size=(B, 108, H, W)).astype(np.float32)
    mask = (rs.uniform(size=gt.shape) > 0.3).astype(np.int32)
    mask_padding = (np.abs(gt) < 3.0).astype(np.int32)
    rec.update(gt=gt, mean=mean, logvar=logvar, scores=scores, mask=mask,
               mask_padding=mask_padding)
    tgt, tmean, tlv, tsc = map(torch.from_numpy, (gt, mean, logvar, scores))
    tmask, tmp = torch.from_numpy(mask), torch.from_numpy(mask_padding)
    cls = ref_dl.reg_to_class(tgt, -3.5, 3.5, 108)
    rec['reg_to_class'] = cls.numpy().astype(np.uint8)
    rec['class_to_reg'] = ref_dl.class_to_reg(cls, -3.5, 3.5, 108).numpy()
    out = {'mean': tmean, 'logvar': tlv, 'scores': tsc}
    rec['l1'] = ref_loss.MaskedL1Loss()(out, tgt, tmask).numpy()
    rec['mse'] = ref_loss.MaskedMSELoss()(out, tgt, tmask).numpy()
    rec['badpix'] = ref_loss.MaskedBadPix()(out, tgt, tmask).numpy()
    rec['upr'] = ref_loss.ImprovedUncertaintyL1Loss()(out, tgt, tmask, None).numpy()
    rec['upr_padding'] = ref_loss.ImprovedUncertaintyL1Loss()(
        {'mean': tmean.clone(), 'logvar': tlv.clone()}, tgt, tmask, tmp).numpy()
    rec['ce'] = ref_loss.MaskedCrossEntropy()(out, cls, tmask).numpy()
    zero = torch.zeros_like(tmask)
    rec['l1_zero_mask'] = ref_loss.MaskedL1Loss()(out, tgt, zero).numpy()
    for mg in (0, 11, 15):
        rec[f'margin_{mg}'] = ref_loss.create_mask_margin((2, 40, 44), mg).numpy()
    # gradients of the three training losses w.r.t. the head outputs
    for name, fn, keys in (('l1', lambda o: ref_loss.MaskedL1Loss()(o, tgt, tmask), ['mean']),
                           ('upr', lambda o: ref_loss.ImprovedUncertaintyL1Loss()(o, tgt, tmask, None),
                            ['mean', 'logvar']),
                           ('ce', lambda o: ref_loss.MaskedCrossEntropy()(o, cls, tmask), ['scores'])):
        o = {k: v.clone().requires_grad_(True) for k, v in out.items()}
        fn(o).backward()
        for k in keys:
            rec[f'd{name}_d{k}'] = o[k].grad.numpy()
    # UPR / DPP heads given a raw trunk output
    m_upr = RefFeedForward(**dict(TINY_KW, model_uncert=True))
    rec['grid_np'] = np.linspace(-3.5, 3.5, 108)
    rec['grid_torch'] = torch.linspace(-3.5, 3.5, 108).numpy()
    from mmlf.model.feed_forward import laplacian
    post = torch.zeros((B, 108, H, W))
    post[:, :, :, :] = torch.from_numpy(np.linspace(-3.5, 3.5, 108)).view(1, -1, 1, 1)
    rec['upr_posterior'] = laplacian(post, tmean, torch.exp(tlv)).numpy()
    del m_upr
    np.savez_compressed(os.path.join(HERE, 'g5_losses.npz'), **rec)
    print('G5 ok', len(rec))


def g6_multimodal():
    """multimodal targets and losses (loss.py:80-103,336-372; dl.py:134-157)."""
    rs = np.random.RandomState(6)
    B, P, H, W = 2, 3, 14, 18
    mpi = rs.uniform(size=(B, P, 5, H, W)).astype(np.float32)
    mpi[:, :, 4] = (7.4 * rs.uniform(size=(B, P, H, W)) - 3.7).astype(np.float32)
    wts = rs.uniform(size=(B, P, H, W)).astype(np.float32)
    wts[rs.uniform(size=wts.shape) < 0.35] = 0.0
    wts[:, :, :3, :4] = 0.0                     # out-of-range pixels: no surface at all
    mpi[:, :, 3] = wts
    mean = (mpi[:, 0, 4] + rs.normal(scale=0.2, size=(B, H, W))).astype(np.float32)
    logvar = rs.normal(scale=0.5, size=(B, H, W)).astype(np.float32)
    mask = (rs.uniform(size=(B, H, W)) > 0.25).astype(np.int32)
    rec = dict(mpi=mpi, mean=mean, logvar=logvar, mask=mask)
    tm, tl, tmpi, tmask = map(torch.from_numpy, (mean, logvar, mpi, mask))
    rec['mpi_to_weights'] = ref_dl.mpi_to_weights(tmpi, -3.5, 3.5, 108).numpy()
    for name, fn, keys in (('multi_l1', ref_loss.MultiMaskedL1Loss(), ['mean']),
                           ('multi_upr', ref_loss.ImprovedMultiUncertaintyL1Loss(), ['mean', 'logvar'])):
        o = {'mean': tm.clone().requires_grad_(True), 'logvar': tl.clone().requires_grad_(True)}
        val = fn(o, tmpi, tmask)
        val.backward()
        rec[name] = val.detach().numpy()
        for k in keys:
            rec[f'd{name}_d{k}'] = o[k].grad.numpy()
    # validate-side metric inputs/outputs (row f2), numpy like the reference; the reference's
    # kl_divergence only broadcasts for batch size 1 (validate/cli.py:179), which is what validate uses
    from mmlf.validate import cli as vcli
    mpi1, mean1, logvar1 = mpi[:1], mean[:1], logvar[:1]
    rec['multimodal_mask'] = vcli.multimodal_mask(mpi1)
    rec['laplace_to_discrete'] = vcli.laplace_to_discrete(108, -3.5, 3.5, mean1, logvar1)
    means = np.stack([mean1 + 0.1 * k for k in range(5)]).astype(np.float32)
    logvars = np.stack([logvar1 - 0.05 * k for k in range(5)]).astype(np.float32)
    rec['lmm_means'], rec['lmm_logvars'] = means, logvars
    import io, contextlib
    with contextlib.redirect_stdout(io.StringIO()):
        # validate/cli.py:302,318 passes exp(logvars) under the name `logvars`; mirror the call
        rec['lmm_to_discrete'] = vcli.lmm_to_discrete(108, -3.5, 3.5, means, np.exp(logvars))
        dist_gt = ref_dl.mpi_to_weights(torch.from_numpy(mpi1), -3.5, 3.5, 108).numpy()
        rec['kld'] = vcli.kl_divergence(rec['laplace_to_discrete'].copy(), dist_gt.copy())
        rec['kld_masked'] = vcli.kl_divergence(rec['laplace_to_discrete'].copy(), dist_gt.copy(),
                                               rec['multimodal_mask'])
        rec['nll_laplace'] = vcli.nll_laplace(mpi1, mean1, logvar1, None)
        rec['mean_to_discrete'] = vcli.mean_to_discrete(108, -3.5, 3.5, mean1)
    np.savez_compressed(os.path.join(HERE, 'g6_multimodal.npz'), **rec)
    print('G6 ok', {k: np.asarray(v).shape for k, v in rec.items()})


def g7_patch_pipeline():
    """The training transform chain of reference train/cli.py:72-91 on synthetic cached scenes; Python's
    `random` is seeded per sample, so a test that seeds it the same way draws the same parameters."""
    import copy
    import random
    from mmlf.data import hci4d as ref_hci4d
    rec = {}
    cases = [('a', 0, 72, 80, 8, 2, True), ('b', 1, 128, 136, 8, 4, True), ('c', 2, 40, 44, 6, 1, False),
             ('d', 3, 96, 100, 16, 2, True)]
    for tag, seed, H, W, ps, max_down, augment in cases:
        scene = synth.synth_scene(seed, H, W)
        if augment:
            chain = [ref_hci4d.RandomDownSampling(max_down), ref_hci4d.RandomShift(1.0),
                     ref_hci4d.RandomCrop(ps + 2 * 4 * 2), ref_hci4d.CenterCrop(ps), ref_hci4d.RandomRotate(),
                     ref_hci4d.RedistColor(), ref_hci4d.Brightness(), ref_hci4d.Contrast()]
        else:
            chain = [ref_hci4d.RandomCrop(ps + 2 * 4 * 2), ref_hci4d.CenterCrop(ps)]
        rec[f'{tag}.cfg'] = np.array([seed, H, W, ps, max_down, int(augment)])
        for k in range(6):
            random.seed(100 * seed + k)
            data = copy.deepcopy(scene)            # hci4d.py:289-291
            for tf in chain:
                data = tf(data)
            for name, arr in zip(('h', 'v', 'i', 'd', 'center', 'gt', 'mpi', 'mask'), data):
                rec[f'{tag}.{k}.{name}'] = np.ascontiguousarray(arr)
    # a fixed pre-shift (--train_shift) in front of the chain, train/cli.py:93-94
    scene = synth.synth_scene(4, 64, 72)
    chain = [ref_hci4d.Shift(0.37), ref_hci4d.RandomDownSampling(2), ref_hci4d.RandomShift(1.0),
             ref_hci4d.RandomCrop(8 + 16), ref_hci4d.CenterCrop(8), ref_hci4d.RandomRotate(),
             ref_hci4d.RedistColor(), ref_hci4d.Brightness(), ref_hci4d.Contrast()]
    for k in range(3):
        random.seed(900 + k)
        data = copy.deepcopy(scene)
        for tf in chain:
            data = tf(data)
        for name, arr in zip(('h', 'v', 'i', 'd', 'center', 'gt', 'mpi', 'mask'), data):
            rec[f'e.{k}.{name}'] = np.ascontiguousarray(arr)
    np.savez_compressed(os.path.join(HERE, 'g7_patch_pipeline.npz'), **rec)
    print('G7 ok', len(rec), 'arrays')


def g8_extras():
    rec = {}
    # (a) UPR loss with mask_padding: gradients (inputs are G5's)
    g5 = dict(np.load(os.path.join(HERE, 'g5_losses.npz')))
    tgt, tmask, tmp = (torch.from_numpy(g5[k]) for k in ('gt', 'mask', 'mask_padding'))
    o = {'mean': torch.from_numpy(g5['mean']).requires_grad_(True), 'logvar': torch.from_numpy(g5['logvar']).requires_grad_(True)}
    val = ref_loss.ImprovedUncertaintyL1Loss()(o, tgt, tmask, tmp)
    val.backward()
    rec['upr_padding'] = val.detach().numpy()
    rec['dupr_padding_dmean'], rec['dupr_padding_dlogvar'] = o['mean'].grad.numpy(), o['logvar'].grad.numpy()
    # (b) multimodal: cross entropy on mpi_to_weights(mpi) and the padded multimodal L1 / UPR losses (inputs are G6's)
    g6 = dict(np.load(os.path.join(HERE, 'g6_multimodal.npz')))
    rs = np.random.RandomState(8)
    B, P, _, H, W = g6['mpi'].shape
    scores = rs.normal(scale=1.5, size=(B, 108, H, W)).astype(np.float32)
    rec['scores'] = scores
    tmpi, tmask6 = torch.from_numpy(g6['mpi']), torch.from_numpy(g6['mask'])
    sc = torch.from_numpy(scores).requires_grad_(True)
    val = ref_loss.MaskedCrossEntropy()({'scores': sc}, ref_dl.mpi_to_weights(tmpi, -3.5, 3.5, 108), tmask6)
    val.backward()
    rec['multi_ce'], rec['dmulti_ce_dscores'] = val.detach().numpy(), sc.grad.numpy()
    pad = 3.0
    mpi_p = tmpi.clone()                                   # train/cli.py:219-220
    mpi_p[:, :, 3, :, :] *= (torch.abs(mpi_p[:, :, 4, :, :]) < pad).float()
    rec['pad'] = np.float32(pad)
    for name, fn, keys in (('multi_l1_pad', ref_loss.MultiMaskedL1Loss(), ['mean']),
                           ('multi_upr_pad', ref_loss.ImprovedMultiUncertaintyL1Loss(), ['mean', 'logvar'])):
        o = {'mean': torch.from_numpy(g6['mean']).requires_grad_(True), 'logvar': torch.from_numpy(g6['logvar']).requires_grad_(True)}
        val = fn(o, mpi_p, tmask6)
        val.backward()
        rec[name] = val.detach().numpy()
        for k in keys:
            rec[f'd{name}_d{k}'] = o[k].grad.numpy()
    np.savez_compressed(os.path.join(HERE, 'g8_losses.npz'), **rec)
    print('G8 losses', {k: np.asarray(v).shape for k, v in rec.items()})

    # (c) DPP full-size train step (B=2): loss, sampled gradients, BN buffers
    kw = dict(BASE_KW, model_discrete=True)
    model, state = build_ref(kw, seed=21)
    stacks, gt, mask = synth.synth_inputs(2, 96, seed=8)
    m = train_mask(mask)
    model.train()
    model.zero_grad()
    out = model(*[torch.from_numpy(s) for s in stacks])
    loss = loss_for('dpp', out, torch.from_numpy(gt), m, kw)
    loss.backward()
    rec = {'loss': loss.detach().numpy(), 'train_scores_s': out['scores'].detach().numpy()[:, :, ::8, ::8].copy(),
           'train_argmax': out['scores'].detach().numpy().argmax(1).astype(np.int16)}
    for n, p in model.named_parameters():
        g = p.grad.numpy()
        rec[f'grad_s/{n}'] = sample(g) if g.size > 4096 else g.copy()
    for n, v in model.state_dict().items():
        if 'running' in n or 'num_batches' in n:
            rec[f'post/{n}'] = v.numpy().copy()
    np.savez_compressed(os.path.join(HERE, 'g2_full_dpp_train.npz'), **rec)
    print('G8 dpp train loss', float(loss))

    # (d) --train_eval_mode (train/cli.py:227-230): model.eval() during the optimisation step -- BatchNorm uses
    # and does not update its running statistics, gradients flow through them as constants
    kw = dict(TINY_KW, model_uncert=True)
    model, state = build_ref(kw, seed=12)
    stacks, gt, mask = synth.synth_inputs(3, 16, seed=6)
    m = torch.from_numpy(mask).int() * ref_loss.create_mask_margin(mask.shape, 3)
    model.eval()
    opt = torch.optim.Adam(model.parameters(), lr=1e-3)
    opt.zero_grad()
    out = model(*[torch.from_numpy(s) for s in stacks])
    loss = loss_for('upr', out, torch.from_numpy(gt), m, kw)
    loss.backward()
    rec = {'loss': loss.detach().numpy(), 'mask': m.numpy(), 'gt': gt}
    rec.update(out_arrays({k: out[k] for k in ('mean', 'logvar')}, 'out_'))
    for n, p in model.named_parameters():
        rec[f'grad/{n}'] = p.grad.numpy().copy()
    opt.step()
    for n, v in model.state_dict().items():
        rec[f'post/{n}'] = v.numpy().copy()
    for i, s in enumerate(stacks):
        rec[f'in{i}'] = s
    np.savez_compressed(os.path.join(HERE, 'g8_evalmode_upr.npz'), **rec)
    print('G8 eval-mode loss', float(loss))

    # (e) a checkpoint written by the REFERENCE's ModelSaver after two Adam steps, and the model one resumed
    # step later (reference resume: train/cli.py:137-157)
    kw = dict(TINY_KW, train_lr=1e-3, model_radius=5, val_disp_step=0.1)
    model, state = build_ref(TINY_KW, seed=4)
    stacks, gt, mask = synth.synth_inputs(2, 16, seed=2)
    m = torch.from_numpy(mask).int() * ref_loss.create_mask_margin(mask.shape, 3)
    tst = [torch.from_numpy(s) for s in stacks]

    def one_step(model, opt):
        model.train()
        opt.zero_grad()
        l = ref_loss.MaskedL1Loss()(model(*tst), torch.from_numpy(gt), m)
        l.backward()
        opt.step()
        return float(l)

    opt = torch.optim.Adam(model.parameters(), lr=1e-3)
    for _ in range(2):
        one_step(model, opt)
    ref_dl.ModelSaver(only_best=False)(os.path.join(HERE, 'g9_checkpoint.pt'), torch.nn.DataParallel(model), opt, kw,
                                       None, 2, 0.5)
    # resume in a fresh model + optimizer, exactly as the reference does, and take one more step
    model2 = RefFeedForward(**TINY_KW)
    opt2 = torch.optim.Adam(model2.parameters(), lr=1e-5)
    ck = torch.load(os.path.join(HERE, 'g9_checkpoint.pt'))
    model2.load_state_dict(ck['model_state_dict'])
    opt2.load_state_dict(ck['optimizer_state_dict'])
    for gparam in opt2.param_groups:
        gparam['lr'] = 1e-3
    l3 = one_step(model2, opt2)
    rec = {'loss3': np.float32(l3), 'iteration': np.int64(ck['iteration'])}
    for n, v in model2.state_dict().items():
        rec[f'post/{n}'] = v.numpy().copy()
    for i, s in enumerate(stacks):
        rec[f'in{i}'] = s
    rec['gt'], rec['mask'] = gt, m.numpy()
    np.savez_compressed(os.path.join(HERE, 'g9_checkpoint_next.npz'), **rec)
    print('G9 checkpoint', os.path.getsize(os.path.join(HERE, 'g9_checkpoint.pt')), 'bytes; resumed loss', l3)


def g10_pfm():
    import tempfile
    from mmlf.utils import pfm as ref_pfm
    rs = np.random.RandomState(31)
    rec = {}
    with tempfile.TemporaryDirectory() as tmp:
        cases = {'grey': rs.randn(7, 5).astype(np.float32),
                 'grey1': rs.randn(4, 6, 1).astype(np.float32),
                 'colour': rs.rand(6, 9, 3).astype(np.float32),
                 # hci4d.py:358-366: save_batch writes np.flip(x.copy(), 0), a negatively strided view
                 'flipped': np.flip(rs.randn(8, 8).astype(np.float32).copy(), 0),
                 'scaled': (100.0 * rs.randn(3, 4)).astype(np.float32)}
        for name, arr in cases.items():
            fn = os.path.join(tmp, name + '.pfm')
            if name == 'scaled':
                ref_pfm.save(fn, arr, scale=2.5)
            else:
                ref_pfm.save(fn, arr)
            rec[f'{name}/array'] = np.ascontiguousarray(arr)
            rec[f'{name}/bytes'] = np.frombuffer(open(fn, 'rb').read(), dtype=np.uint8).copy()
            rec[f'{name}/loaded'] = ref_pfm.load(fn)
        # a big-endian file (positive scale), which only load has to understand
        be = rs.randn(5, 3).astype('>f4')
        fn = os.path.join(tmp, 'be.pfm')
        with open(fn, 'wb') as f:
            f.write(b'Pf\n3 5\n1.000000\n')
            be.tofile(f)
        rec['bigendian/bytes'] = np.frombuffer(open(fn, 'rb').read(), dtype=np.uint8).copy()
        rec['bigendian/loaded'] = ref_pfm.load(fn)
    np.savez_compressed(os.path.join(HERE, 'g10_pfm.npz'), **rec)
    print('G10 pfm', {k: v.shape for k, v in rec.items() if k.endswith('bytes')})


def g10_validate():
    """the per-scene body of validate/cli.py:258-345 on FIXED model outputs (a stub model), for the three head kinds and
    the ensemble: MSE / BadPix with the 15-px margin, the discretised predictive distribution, KL divergences (all,
    multimodal, unimodal pixels), NLL -- computed with the reference's own helpers"""
    import contextlib
    import io
    from mmlf.validate import cli as vcli
    rs = np.random.RandomState(41)
    H = W = 40
    P, S = 3, 5
    mpi = rs.uniform(0.0, 1.0, (1, P, 5, H, W)).astype(np.float32)
    mpi[:, :, 4] = (6.0 * mpi[:, :, 4] - 3.0).astype(np.float32)
    gt = mpi[:, 0, 4].copy()
    mean = (gt + 0.05 * rs.randn(1, H, W)).astype(np.float32)
    logvar = (-2.0 + 0.5 * rs.randn(1, H, W)).astype(np.float32)
    scores = rs.randn(1, 108, H, W).astype(np.float32)
    e = np.exp(scores)
    posterior = (e / e.sum(1, keepdims=True)).astype(np.float32)
    means = (gt[None] + 0.3 * rs.randn(S, 1, H, W)).astype(np.float32)
    logvars = (-2.0 + 0.5 * rs.randn(S, 1, H, W)).astype(np.float32)
    rec = dict(mpi=mpi, gt=gt, mean=mean, logvar=logvar, scores=scores, posterior=posterior, means=means, logvars=logvars)
    mask = ref_loss.create_mask_margin(gt.shape, 15)
    out = {'mean': torch.from_numpy(mean)}
    rec['mse'] = ref_loss.MaskedMSELoss()(out, torch.from_numpy(gt), mask).numpy()
    rec['badpix'] = ref_loss.MaskedBadPix()(out, torch.from_numpy(gt), mask).numpy()
    dist_gt = ref_dl.mpi_to_weights(torch.from_numpy(mpi), -3.5, 3.5, 108).numpy()
    mm = vcli.multimodal_mask(mpi)
    with contextlib.redirect_stdout(io.StringIO()):
        kinds = {
            'base': (vcli.mean_to_discrete(108, -3.5, 3.5, mean), vcli.nll_laplace(mpi, mean, np.zeros_like(mean), None)),
            'upr': (vcli.laplace_to_discrete(108, -3.5, 3.5, mean, logvar), vcli.nll_laplace(mpi, mean, logvar, None)),
            'dpp': (posterior.copy(), vcli.nll_discrete(dist_gt.copy(), posterior.copy(), -3.5, 3.5, None)),
            # validate/cli.py:302,318: exp(logvars) goes in under the name `logvars`
            'ese': (vcli.lmm_to_discrete(108, -3.5, 3.5, means, np.exp(logvars)), 0.0),
        }
        for k, (dist, nll) in kinds.items():
            rec[f'{k}/nll'] = np.float64(nll)
            rec[f'{k}/kld'] = np.float64(vcli.kl_divergence(dist.copy(), dist_gt.copy()))
            rec[f'{k}/kld_mm'] = np.float64(vcli.kl_divergence(dist.copy(), dist_gt.copy(), mm))
            rec[f'{k}/kld_um'] = np.float64(vcli.kl_divergence(dist.copy(), dist_gt.copy(), 1.0 - mm))
        # what the LOOP prints (validate/cli.py:313-337): the helpers rewrite their arguments in place and the loop hands
        # the same arrays to nll_discrete and to three kl_divergence calls in a row -- replayed here without copies
        for k in ('base', 'upr', 'dpp', 'ese'):
            d_gt = ref_dl.mpi_to_weights(torch.from_numpy(mpi), -3.5, 3.5, 108).cpu().numpy()
            if k == 'ese':
                dist, nll = vcli.lmm_to_discrete(108, -3.5, 3.5, means, np.exp(logvars)), 0.0
            elif k == 'dpp':
                post = posterior.copy()                       # the loop's `posterior`: the model output on the host
                weights = ref_dl.mpi_to_weights(torch.from_numpy(mpi), -3.5, 3.5, 108).cpu().numpy()
                dist = post
                nll = vcli.nll_discrete(weights, post, -3.5, 3.5, None)
            elif k == 'upr':
                dist = vcli.laplace_to_discrete(108, -3.5, 3.5, mean, logvar)
                nll = vcli.nll_laplace(mpi, mean, logvar, None)
            else:
                nll = vcli.nll_laplace(mpi, mean, np.zeros_like(mean), None)
                dist = vcli.mean_to_discrete(108, -3.5, 3.5, mean)
            rec[f'{k}/seq_nll'] = np.float64(nll)
            rec[f'{k}/seq_kld'] = np.float64(vcli.kl_divergence(dist, d_gt))
            rec[f'{k}/seq_kld_mm'] = np.float64(vcli.kl_divergence(dist, d_gt, mm))
            rec[f'{k}/seq_kld_um'] = np.float64(vcli.kl_divergence(dist, d_gt, 1.0 - mm))
    np.savez_compressed(os.path.join(HERE, 'g10_validate.npz'), **rec)
    print('G10 validate', {k: float(v) for k, v in rec.items() if '/' in k})


def g10_unet():
    """--model_unet (reference feed_forward.py:189-204, unet.py): outputs only, weights by synth.formula_state"""
    kw = dict(TINY_KW, model_unet=True, model_uncert=True)
    model = RefFeedForward(**kw)
    state = synth.formula_state([(k, v.shape) for k, v in model.state_dict().items()], seed=3)
    model.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in state.items()})
    stacks, gt, mask = synth.synth_inputs(2, 32, seed=9)
    rec = {'keys': np.array(list(model.state_dict().keys()))}
    model.train()
    out = model(*[torch.from_numpy(s) for s in stacks])
    loss = ref_loss.ImprovedUncertaintyL1Loss()(out, torch.from_numpy(gt), train_mask(mask), None)
    loss.backward()
    rec['train_mean'], rec['train_logvar'] = out['mean'].detach().numpy(), out['logvar'].detach().numpy()
    rec['loss'] = loss.detach().numpy()
    rec['grad/out_net.last.weight'] = model.out_net.last.weight.grad.numpy().copy()
    rec['grad/in_net_hv.0.0.weight'] = model.in_net_hv[0][0].weight.grad.numpy().copy()
    model.eval()
    with torch.no_grad():
        out = model(*[torch.from_numpy(s) for s in stacks])
    rec['eval_mean'], rec['eval_logvar'] = out['mean'].numpy(), out['logvar'].numpy()
    np.savez_compressed(os.path.join(HERE, 'g10_unet.npz'), **rec)
    print('G10 unet', float(loss), len(rec['keys']), 'keys')


def g11_conditioned_f64():
    """A WELL-CONDITIONED train step for tight gradient parity.  The gradient of this net is a discontinuous function
    of its activations where few units carry much of it: the head's first convolution has ONE (UPR: two) channel(s),
    and a single ReLU flip there -- a pre-activation within float32 rounding noise (~5e-6) of zero -- moves every
    gradient below it by more than 1 % (measured: g2's input has such a unit; implementations that round differently
    land on either side of it).  The L1 loss adds sign(mean - gt).  So: search input seeds until the reference's
    float32 run keeps every head pre-activation and every masked |mean - gt| at least 1e-4 away from zero (20 x the
    noise), then record the reference's float32 AND float64 runs of that step.  Against this fixture the distance of
    an implementation's gradients from the float64 run is rounding noise alone, and the reference's own float32
    distance is the yardstick."""
    for variant in ('base', 'upr'):
        kw = dict(BASE_KW, **VARIANTS[variant])
        model, state = build_ref(kw, seed=21)
        model.train()
        pre = {}
        model.out_net[7][0].register_forward_hook(lambda mod, inp, out: pre.__setitem__('y', out.detach()))
        chosen = None
        for seed in range(100, 400):
            stacks, gt, mask = synth.synth_inputs(2, 96, seed=seed)
            m = train_mask(mask)
            with torch.no_grad():
                # (train-mode forward: running statistics move, which the recorded run below does not depend on)
                out = model(*[torch.from_numpy(s) for s in stacks])
            margin_y = float(pre['y'].abs().min())
            d = (out['mean'] - torch.from_numpy(gt)).abs()
            margin_l = float(d[m.bool()].min())
            print('G11', variant, 'seed', seed, 'min |head pre-activation|', margin_y, 'min |mean - gt|', margin_l, flush=True)
            if margin_y > 1e-4 and margin_l > 1e-4:
                chosen = seed
                break
        assert chosen is not None
        rec = {'seed': np.int64(chosen)}
        for prec in ('f32', 'f64'):
            model, state = build_ref(kw, seed=21)
            stacks, gt, mask = synth.synth_inputs(2, 96, seed=chosen)
            m = train_mask(mask)
            if prec == 'f64':
                model = model.double()
            cast = (lambda a: torch.from_numpy(a).double()) if prec == 'f64' else torch.from_numpy
            model.train()
            model.zero_grad()
            out = model(*[cast(s) for s in stacks])
            loss = loss_for(variant, out, cast(gt), m, kw)
            loss.backward()
            rec[f'{prec}/train_mean'] = out['mean'].detach().numpy()
            if variant == 'upr':
                rec[f'{prec}/train_logvar'] = out['logvar'].detach().numpy()
            rec[f'{prec}/loss'] = loss.detach().numpy()
            for n, p in model.named_parameters():
                g = p.grad.numpy()
                rec[f'{prec}/grad_s/{n}'] = sample(g) if g.size > 4096 else g.copy()
        np.savez_compressed(os.path.join(HERE, f'g11_conditioned_{variant}.npz'), **rec)
        print('G11', variant, 'seed', chosen, float(rec['f32/loss']), float(rec['f64/loss']))


if __name__ == '__main__':
    if len(sys.argv) > 1 and sys.argv[1] == 'g11':
        g11_conditioned_f64()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == 'g10u':
        g10_unet()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == 'g10v':
        g10_validate()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == 'g10':
        g10_pfm()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == 'g8':
        g8_extras()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == 'g6':
        g6_multimodal()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == 'g7':
        g7_patch_pipeline()
        sys.exit(0)
    g1_tiny()
    g4_shift_ensamble()
    g5_losses()
    g2_full()
    g6_multimodal()
    g7_patch_pipeline()
    g8_extras()
    g10_pfm()
    g10_unet()
    g10_validate()
    g11_conditioned_f64()
    sizes = {f: os.path.getsize(os.path.join(HERE, f)) for f in sorted(os.listdir(HERE))
             if f.endswith('.npz')}
    print(sizes)
