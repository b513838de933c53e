"""GPU half of the round-5 bounds audit (tests/test_bounds_audit.py is the CPU half): a -DMMLF_BOUNDS_DEBUG build of the
library, made here with hipcc, counts every access of the convolution / weight-gradient kernels that leaves what the ABI says
the buffers hold; tools/bounds_check.py drives it through tools/kbench.py's launch sequence, two training steps and a wide
frame, in a process of its own (the library is chosen at load time).  Expected: zeros.

Kernels audited replace nn.Conv2d forward / backward (reference mmlf/model/feed_forward.py:123-125)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
@pytest.mark.timeout(900)
def test_no_access_leaves_the_buffers_in_a_bounds_counting_build(tmp_path):
    from mmlf_amd.csrc import build
    lib = build.build(verbose=False, extra_flags=['-DMMLF_BOUNDS_DEBUG'], lib=str(tmp_path / 'libmmlf_bounds.so'))
    env = dict(os.environ, MMLF_HIP_LIB=lib)
    batch = os.environ.get('MMLF_BOUNDS_BATCH', '512')          # kbench's own shape: the run that faulted once
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'tools', 'bounds_check.py'), batch], cwd=ROOT, env=env,
                       capture_output=True, text=True, timeout=800)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    report = json.loads(r.stdout.strip().splitlines()[-1])
    assert 'MMLF_BOUNDS_DEBUG=1' in report['build']
    for leg in ('kbench_280', 'kbench_70', 'train_base', 'train_dpp', 'eval_wide_frame'):
        assert all(v == 0 for v in report[leg].values()), (leg, report[leg])
    assert report['total'] == 0


@pytest.mark.gpu
def test_host_extent_check_passes_the_step_and_refuses_a_short_buffer(monkeypatch):
    """MMLF_CHECK_EXTENTS (round 6): the product kernels' range-checked descriptors DROP a stray access instead of faulting,
    so the product build needs a signal of its own -- the host holds the audited end of every convolution / weight-gradient
    launch against the bytes really behind each pointer.  A DPP train step and a tiled evaluation pass run clean under it
    (every launch checked), and a buffer without the prescribed slack is refused before anything is launched."""
    import numpy as np
    import torch
    from conftest import TINY_KW
    from mmlf_amd import engine, synth
    from mmlf_amd.feed_forward import FeedForward
    from mmlf_amd.train import TrainStep
    monkeypatch.setattr(engine, 'CHECK_EXTENTS', True)
    monkeypatch.setattr(engine, 'EXTENT_CHECKS', 0)
    dev = torch.device('cuda:0')
    kw = dict(TINY_KW, model_discrete=True)
    model = FeedForward(**kw)
    model.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in synth.synth_state(synth.param_spec(**kw), 2).items()})
    model.to(dev)
    stacks, gt, mask = synth.synth_inputs(3, 24, seed=4)
    t = [torch.from_numpy(s).to(dev) for s in stacks]
    step = TrainStep(model, lr=1e-3, loss_margin=3)
    loss = step(*t, torch.from_numpy(gt).to(dev), torch.from_numpy(mask).to(dev), 1)
    model.eval()
    with torch.no_grad():
        model(*t)
    torch.cuda.synchronize()
    assert np.isfinite(float(loss))
    # 2 nets x 2 streams x 2 blocks + 3 out blocks = 11 blocks: 2 forward convs each in both passes, 2 data + 2 weight gradients
    # (no data gradient into the images)
    assert engine.EXTENT_CHECKS >= 11 * 2 * 2 + 11 * 2 + 7, engine.EXTENT_CHECKS
    # a short input buffer: the image positions alone, without the tile padding and tap slack mmlf_grid_alloc_positions adds
    geo = engine.Geometry(3, 24, 24)
    cs = engine.cs_of(8)
    x = geo.buf(cs, dev)
    short = x[:geo.NQ * cs].clone()
    short.absmax = x.absmax
    out = geo.buf(cs, dev)
    w = torch.randn(8, 8, 2, 2, device=dev)
    pk = engine.pack_filter(w, 0, False)
    with pytest.raises(RuntimeError, match='MMLF_CHECK_EXTENTS.*`in`'):
        engine.conv(geo, short, cs, 8, pk, torch.zeros(8, device=dev), 8, out, cs, 0, 25, 25, True)
    engine.conv(geo, x, cs, 8, pk, torch.zeros(8, device=dev), 8, out, cs, 0, 25, 25, True)       # the full one passes
    torch.cuda.synchronize()
