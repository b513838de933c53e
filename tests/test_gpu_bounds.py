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
