"""Register and scratch budgets of the step's hot kernels, read from the compiler (no GPU: hipcc cross-compiles gfx950).

A spill in one of these kernels does not fail any parity test -- it only costs time (round 5: the transposed epilogue's first
form spilled 24-124 bytes per lane in the wide kernels until its table reads were fenced) -- so the budgets the design
relies on are asserted here: no scratch in the launch kinds of a training step, the occupancy the launch bounds ask for,
and the three figures that let a BatchNorm-backward workgroup share a CU with a wide weight-gradient one
(2 x 232 + 48 <= 512 registers per SIMD lane: engine.py, MMLF_OVERLAP_WGRAD).
The kernels replace nn.Conv2d / nn.BatchNorm2d forward and backward (reference mmlf/model/feed_forward.py:123-135)."""
import os
import re
import subprocess
import tempfile

import pytest

from mmlf_amd.csrc import build

HIPCC = build.HIPCC


def _resource_usage():
    """kernel name -> (vgprs, agprs, scratch bytes per lane, waves per SIMD) for every kernel of the library's sources"""
    flags = [f for f in build.FLAGS if f not in ('-fPIC', '-Wall')]
    jobs = []
    with tempfile.TemporaryDirectory() as tmp:
        for src in build.SOURCES:
            out = os.path.join(tmp, src + '.co')
            cmd = [HIPCC, *flags, '--cuda-device-only', '-Rpass-analysis=kernel-resource-usage', '-c',
                   os.path.join(build.HERE, src), '-o', out]
            jobs.append(subprocess.Popen(cmd, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, text=True))
        texts = [j.communicate()[1] for j in jobs]
        assert all(j.returncode == 0 for j in jobs), texts
    usage = {}
    for text in texts:
        for block in re.split(r'remark: [^\n]*Function Name: ', text)[1:]:
            name = block.split(' [')[0]

            def num(key):
                m = re.search(key + r': (\d+)', block)
                return int(m.group(1)) if m else -1
            usage[name] = (num('VGPRs'), num('AGPRs'), num(r'ScratchSize \[bytes/lane\]'), num(r'Occupancy \[waves/SIMD\]'))
    return usage


@pytest.fixture(scope='module')
def usage():
    if not os.path.exists(HIPCC):
        pytest.skip('no hipcc')
    return _resource_usage()


def _demangled(usage, stem, args):
    """the entry of template kernel `stem` instantiated with the integer / bool arguments `args`"""
    enc = ''.join(f'Li{a}E' if not isinstance(a, bool) and a >= 0 else (f'Lin{-a}E' if not isinstance(a, bool) else f'Lb{int(a)}E')
                  for a in args)
    hits = [k for k in usage if stem in k and f'I{enc}E' in k]
    assert len(hits) == 1, (stem, args, hits)
    return usage[hits[0]]


# the launch kinds of a bs=512 BASE / UPR / DPP training step on 96x96 patches (profiles/r05_bs512_base_f16x3_kernel_stats.csv)
CONV_KINDS = [(18, 2, 0, 8, True), (18, 2, 4, 8, True), (18, 2, 17, 8, True), (18, 2, 2, 8, False),
              (5, 2, 0, 16, False), (5, 2, 2, 16, False), (5, 2, 4, 16, True), (5, 2, 17, 16, True), (5, 2, 1, 16, True)]


@pytest.mark.parametrize('args', CONV_KINDS)
def test_conv_launch_kinds_of_a_training_step_do_not_spill(usage, args):
    vgprs, agprs, scratch, waves = _demangled(usage, 'conv4tap_x6s_kernel', args)
    assert scratch == 0, (args, vgprs, scratch)
    assert waves >= (2 if args[0] == 18 else 4), (args, waves)       # one 512-thread / 1024-thread workgroup per CU
    assert vgprs <= (256 if args[0] == 18 else 128)


def test_weight_gradient_kernels_do_not_spill_and_leave_room_for_batchnorm(usage):
    wide = _demangled(usage, 'wgrad4tap_x6w_kernel', (3, 9, 2))
    assert wide[2] == 0 and wide[3] >= 2
    assert wide[0] <= 232, wide              # 2 waves x 232 registers per SIMD lane: 48 are left
    for mb, nb in ((5, 5), (2, 5), (3, 8), (2, 8), (2, 2)):
        v = _demangled(usage, 'wgrad4tap_x6n_kernel', (mb, nb, 2))
        assert v[2] == 0 and v[3] >= 2, (mb, nb, v)
    reduce_bwd = [v for k, v in usage.items() if 'bn_reduce_bwd_kernel' in k]
    assert len(reduce_bwd) == 1 and reduce_bwd[0][0] <= 48 and reduce_bwd[0][2] == 0, reduce_bwd
    rows_bwd = _demangled(usage, 'bn_rows_kernel', (1, 4))
    assert rows_bwd[0] <= 48 and rows_bwd[2] == 0, rows_bwd


def test_register_streamed_full_frame_kinds(usage):
    """the evaluation path's narrow launches (ReLU, 27 -> 70 and 70 -> 70) hold their 2 waves per SIMD without scratch"""
    for nch in (4, 9):
        for tr in (True, False):
            v = _demangled(usage, 'conv4tap_rs_kernel', (5, nch, 1, tr))
            assert v[2] == 0 and v[3] >= 2, (nch, tr, v)
