"""The pieces of the path composed the way the reference's two drivers compose theirs (mmlf/train/cli.py:185-306,
mmlf/validate/cli.py:249-351), on one MI355X: scenes cached in HBM -> on-device patch batches -> native train steps ->
checkpoint in the reference's format -> resume in a fresh process state -> validation loop with result files.
Not a parity test (each piece has its own, against goldens): this one checks that they fit together and that training
actually trains."""
import random

import numpy as np
import pytest
import torch

from conftest import TINY_KW
from mmlf_amd import dl, patches, pfm, synth, validate
from mmlf_amd.feed_forward import FeedForward
from mmlf_amd.train import TrainStep

pytestmark = pytest.mark.gpu


def _scene(seed, H=72, W=72):
    """a scene whose disparity is a smooth function of the images, so that a few dozen steps can learn something"""
    s = list(synth.synth_scene(seed, H, W))
    yy, xx = np.meshgrid(np.linspace(-1, 1, H), np.linspace(-1, 1, W), indexing='ij')
    gt = (0.8 * np.sin(2.0 * xx + seed) * np.cos(1.5 * yy)).astype(np.float32)
    s[5] = gt
    for k in range(4):                           # every view carries the disparity in its red channel
        s[k] = s[k].copy()
        s[k][:, 0] = (0.5 + 0.4 * gt)[None]
    s[6] = s[6].copy()
    s[6][:, 4] = gt[None]                         # planes agree with gt
    s[7] = np.ones((H, W), np.int64)
    return tuple(s)


@pytest.mark.parametrize('variant', ['base', 'upr'])
def test_train_checkpoint_resume_validate(variant, tmp_path):
    kw = dict(TINY_KW, model_uncert=(variant == 'upr'), train_lr=2e-3)
    scenes = [_scene(k) for k in range(3)]
    pipe = patches.PatchPipeline(scenes, 16, 2, augment=False)
    torch.manual_seed(0)
    model = FeedForward(**kw).cuda()
    step = TrainStep(model, lr=kw['train_lr'], warm_start=False, loss_margin=2)
    random.seed(5)
    losses = []
    for it in range(1, 61):
        h, v, i_, d, center, gt, mpi, mask, index = pipe.sample([random.randrange(3) for _ in range(8)])
        losses.append(float(step(h, v, i_, d, gt, mask, it)))
    assert np.isfinite(losses).all()
    assert np.mean(losses[-10:]) < 0.7 * np.mean(losses[:10]), (np.mean(losses[:10]), np.mean(losses[-10:]))
    # checkpoint in the reference's format, resume into fresh objects, continue identically
    ck = str(tmp_path / 'checkpoint.pt')
    dl.ModelSaver()(ck, model, step, kw, 1, 60, losses[-1])
    batch = pipe.sample([0, 1, 2, 0])
    cont = float(step(*batch[:4], batch[5], batch[7], 61))
    model2 = FeedForward(**kw).cuda()
    step2 = TrainStep(model2, lr=1e-9, warm_start=False, loss_margin=2)
    it0, state = dl.load_checkpoint(ck, model2, step2, lr=kw['train_lr'], map_location='cuda')
    assert it0 == 60 and state['hyper_parameters']['model_chs'] == kw['model_chs']
    cont2 = float(step2(*batch[:4], batch[5], batch[7], 61))
    assert cont2 == cont
    assert torch.equal(step.flat, step2.flat)
    # validation loop over the full frames, result files in the reference's layout
    def frames():
        for k, s in enumerate(scenes):
            yield tuple(torch.from_numpy(np.asarray(a)).unsqueeze(0).cuda() for a in s[:8]) + (torch.tensor([[k]]),)
    rows, avg = validate.validate_scenes(model2, frames(), margin=4, out_dir=str(tmp_path), scene_names=['s0', 's1', 's2'])
    assert len(rows) == 3 and np.isfinite([v for r in rows for v in r.values()]).all()
    # the training loop's own periodic validation (train/cli.py:265-325) on the same frames: same MSE / BadPix averages as
    # the validate CLI's loop, the loss of the training family beside them, a checkpoint and a log row from its numbers
    from mmlf_amd.train import log_line, validation_pass
    lv, mse_v, bp_v = validation_pass(model2, frames(), uncert=(variant == 'upr'), margin=4, out_dir=str(tmp_path / 'val'),
                                      scene_names=['s0', 's1', 's2'])
    np.testing.assert_allclose([mse_v, bp_v], [avg['mse'], avg['badpix']], rtol=1e-6)
    assert np.isfinite(lv) and (tmp_path / 'val' / 'ours' / 'disp_maps' / 's1.pfm').exists()
    dl.ModelSaver()(str(tmp_path / 'val' / 'checkpoint.pt'), model2, step2, kw, None, 61, lv)
    assert torch.load(str(tmp_path / 'val' / 'checkpoint.pt'))['loss'] == lv and log_line(61, cont2, lv, mse_v, bp_v, 0.1)
    res = pfm.load(str(tmp_path / 'ours' / 'disp_maps' / 's2.pfm'))
    assert res.shape == (72, 72) and res.dtype == np.float32
    # the trained model is better than predicting zero on the frames it saw patches of
    # (BASE: L1 on the mean alone; the UPR loss spends its first steps on the log-variance)
    zero_mse = np.mean([np.mean(s[5][4:-4, 4:-4] ** 2) for s in scenes])
    if variant == 'base':
        assert avg['mse'] < zero_mse, (avg['mse'], zero_mse)
    if variant == 'upr':
        assert (tmp_path / 'scenes' / 's0' / 'uncert.pfm').exists()
