"""the training patch pipeline (reference hci4d.py transforms as
train/cli.py:72-94 composes them).  CPU: the oracle restatement against the reference's outputs
(tests/golden/g7_patch_pipeline.npz); GPU: the fused HIP gather against both."""
import os
import random

import numpy as np
import pytest
import torch

from mmlf_amd import synth

HERE = os.path.dirname(os.path.abspath(__file__))
NAMES = ('h', 'v', 'i', 'd', 'center', 'gt', 'mpi', 'mask')
# float tolerance: every op is a float32 blend / 3x3 colour mix of values in [0,2); the reference and the
# restatement differ only in the summation order of the contrast mean (pairwise float32 vs double)
TOL = 2e-6


@pytest.fixture(scope='module')
def g7():
    return np.load(os.path.join(HERE, 'golden', 'g7_patch_pipeline.npz'))


def _cases(g7):
    for tag in 'abcd':
        seed, H, W, ps, max_down, augment = [int(x) for x in g7[f'{tag}.cfg']]
        yield tag, seed, H, W, ps, max_down, bool(augment)


def test_oracle_matches_reference_chain(g7):
    from oracle import oracle as orc
    for tag, seed, H, W, ps, max_down, augment in _cases(g7):
        scene = synth.synth_scene(seed, H, W)
        for k in range(6):
            random.seed(100 * seed + k)
            out = orc.augment_sample(scene, ps, max_down, augment)
            for name, arr in zip(NAMES, out):
                ref = g7[f'{tag}.{k}.{name}']
                assert arr.shape == ref.shape, (tag, k, name)
                if name == 'mask':
                    assert np.array_equal(arr, ref), (tag, k, name)
                else:
                    assert np.abs(arr.astype(np.float64) - ref).max() <= TOL, (tag, k, name)


def test_oracle_fixed_preshift(g7):
    """--train_shift puts a fixed Shift in front of the chain (train/cli.py:93-94)."""
    from oracle import oracle as orc
    scene = synth.synth_scene(4, 64, 72)
    pre = orc.tf_shift(scene, 0.37)
    for k in range(3):
        random.seed(900 + k)
        out = orc.augment_sample(pre, 8, 2, True)
        for name, arr in zip(NAMES, out):
            ref = g7[f'e.{k}.{name}']
            if name == 'mask':
                assert np.array_equal(arr, ref)
            else:
                assert np.abs(arr.astype(np.float64) - ref).max() <= TOL, (k, name)


def test_host_draws_follow_the_reference_order(g7):
    """draw_sample must consume Python's random stream exactly like the reference chain, so that the
    next sample of a batch starts from the same state."""
    from oracle import oracle as orc
    from mmlf_amd import patches
    for tag, seed, H, W, ps, max_down, augment in _cases(g7):
        scene = synth.synth_scene(seed, H, W)
        random.seed(100 * seed)
        orc.augment_sample(scene, ps, max_down, augment)
        after_oracle = random.random()
        random.seed(100 * seed)
        patches.draw_sample((H, W), ps, max_down, augment)
        assert random.random() == after_oracle, tag


def test_rotation_sources_match_rotate90():
    from oracle import oracle as orc
    from mmlf_amd import patches
    V = 5
    lab = [np.broadcast_to((s * V + np.arange(V, dtype=np.float32))[:, None, None, None], (V, 3, 4, 4)).copy()
           for s in range(4)]
    data = (*lab, np.zeros((3, 4, 4), np.float32), np.zeros((4, 4), np.float32),
            np.zeros((1, 5, 4, 4), np.float32), np.zeros((4, 4), np.int64), np.zeros(1))
    tab = patches.rotation_sources(V)
    for r in range(4):
        for s in range(4):
            assert np.array_equal(data[s][:, 0, 0, 0].astype(np.int32), tab[r, s]), (r, s)
        data = orc.tf_rotate90(data)


def test_draw_rejects_frames_not_larger_than_the_crop():
    from mmlf_amd import patches
    with pytest.raises(ValueError):
        patches.draw_sample((24, 40), 8, 1, False)          # RandomCrop asserts h > size (hci4d.py:656)


def _compare(out, expect, tag):
    for name, arr, ref in zip(NAMES, out, expect):
        arr = arr.cpu().numpy()
        assert arr.shape == ref.shape, (tag, name, arr.shape, ref.shape)
        if name == 'mask':
            assert np.array_equal(arr, ref), (tag, name)
        else:
            assert np.abs(arr.astype(np.float64) - ref).max() <= TOL, (tag, name)


@pytest.mark.gpu
def test_device_pipeline_matches_reference_and_oracle(g7):
    from oracle import oracle as orc
    from mmlf_amd import patches
    for tag, seed, H, W, ps, max_down, augment in _cases(g7):
        scene = synth.synth_scene(seed, H, W)
        pipe = patches.PatchPipeline([scene], ps, max_down, augment)
        for k in range(6):
            random.seed(100 * seed + k)
            out = pipe.sample([0])
            _compare([t[0] for t in out[:8]], [g7[f'{tag}.{k}.{n}'] for n in NAMES], (tag, k, 'golden'))
            random.seed(100 * seed + k)
            _compare([t[0] for t in out[:8]], orc.augment_sample(scene, ps, max_down, augment), (tag, k, 'oracle'))


@pytest.mark.gpu
def test_device_pipeline_batches_and_preshift(g7):
    """A batch draws its samples one after the other from one random stream; --train_shift pre-shifts the
    cached scenes once."""
    from oracle import oracle as orc
    from mmlf_amd import patches
    scenes = [synth.synth_scene(10 + s, 80, 72) for s in range(3)]
    pipe = patches.PatchPipeline(scenes, 8, 2, True)
    random.seed(77)
    idx = [2, 0, 1, 1, 2]
    out = pipe.sample(idx)
    random.seed(77)
    for b, sc in enumerate(idx):
        _compare([t[b] for t in out[:8]], orc.augment_sample(scenes[sc], 8, 2, True), ('batch', b))
    assert out[8].shape == (5, 1) and [int(v) for v in out[8][:, 0]] == [12, 10, 11, 11, 12]
    scene = synth.synth_scene(4, 64, 72)
    pipe = patches.PatchPipeline([scene], 8, 2, True, train_shift=0.37)
    for k in range(3):
        random.seed(900 + k)
        out = pipe.sample([0])
        _compare([t[0] for t in out[:8]], [g7[f'e.{k}.{n}'] for n in NAMES], ('preshift', k))


@pytest.mark.gpu
def test_device_pipeline_full_size_properties():
    """512x512 scenes, ps=96: crops without augmentation are plain slices of the scene; the augmented batch
    keeps the colour-matrix invariant (rows of RedistColor sum to 1, so a grey pixel stays grey before
    brightness/contrast) -- checked through per-sample statistics that do not depend on the oracle."""
    from mmlf_amd import patches
    scene = synth.synth_scene(21, 512, 512)
    pipe = patches.PatchPipeline([scene], 96, 4, False)
    random.seed(5)
    out = pipe.sample([0] * 4)
    random.seed(5)
    for b in range(4):
        p = patches.draw_sample((512, 512), 96, 4, False)
        y, x = p['y0'], p['x0']
        assert np.array_equal(out[0][b].cpu().numpy(), scene[0][..., y:y + 96, x:x + 96])
        assert np.array_equal(out[5][b].cpu().numpy(), scene[5][y:y + 96, x:x + 96])
        assert np.array_equal(out[7][b].cpu().numpy(), scene[7][y:y + 96, x:x + 96])
    pipe = patches.PatchPipeline([scene], 96, 4, True)
    random.seed(6)
    out = pipe.sample([0] * 8)
    assert all(torch.isfinite(t).all() for t in out[:7])
    assert out[0].shape == (8, 9, 3, 96, 96) and out[6].shape == (8, 2, 5, 96, 96)
