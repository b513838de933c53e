"""Two host threads make the FIRST launches of a fresh process on one device at the same moment (round 5).

nn.DataParallel (reference mmlf/train/cli.py:159) calls forward from one thread per replica, and nothing stops two replicas
from sharing a device.  The library's one-time per-device set-up (hipFuncSetAttribute for the kernels that need more than
64 KB of LDS) used to publish "done" BEFORE doing the work, so the second thread could launch a 151 KB-LDS kernel ahead of
the attribute that allows it; it is a std::call_once per device now (csrc/common.h PerDeviceOnce) and a failed set-up is an
error, not a discarded return value.  The child process below is started before this process' GPU state matters to it: it
makes no GPU call before the two threads start."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r'''
import sys, threading
import numpy as np, torch
sys.path.insert(0, %r)
from mmlf_amd import synth
from mmlf_amd.feed_forward import FeedForward
kw = dict(model_ksize=2, model_in_blocks=3, model_out_blocks=8, model_chs=70, model_views=9, model_cross=False,
          model_uncert=False, model_unet=False, model_discrete=False, model_no_batchnorm=False,
          model_batchnorm_momentum=0.1, val_disp_min=-3.5, val_disp_max=3.5)
state = synth.synth_state(synth.param_spec(**kw), seed=5)
stacks, gt, mask = synth.synth_inputs(2, 20, seed=5)
gate = threading.Barrier(2)
outs, errs = [None, None], []

def worker(k):
    try:
        dev = torch.device('cuda:0')
        model = FeedForward(**kw)
        model.load_state_dict({n: torch.from_numpy(np.asarray(v)) for n, v in state.items()})
        model.to(dev).train()
        t = [torch.from_numpy(s).to(dev) for s in stacks]
        torch.cuda.synchronize()
        gate.wait()                       # both threads make their first kernel launches of this process together
        out = model(*t)
        out['mean'].sum().backward()
        torch.cuda.synchronize()
        outs[k] = (out['mean'].detach().cpu(), model.out_net[0][0].weight.grad.detach().cpu())
    except Exception as e:               # noqa: BLE001
        errs.append(repr(e))
        try:
            gate.abort()
        except Exception:
            pass

ths = [threading.Thread(target=worker, args=(k,)) for k in range(2)]
[t.start() for t in ths]
[t.join() for t in ths]
assert not errs, errs
assert torch.isfinite(outs[0][0]).all()
# same weights, same input, train mode: the two threads must agree bit for bit (deterministic kernels)
assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
print('two-thread first launch ok', float(outs[0][0].abs().mean()))
'''


@pytest.mark.gpu
@pytest.mark.timeout(600)
def test_two_threads_first_launch_in_a_fresh_process():
    r = subprocess.run([sys.executable, '-c', CHILD % ROOT], cwd=ROOT, capture_output=True, text=True, timeout=500)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert 'two-thread first launch ok' in r.stdout
