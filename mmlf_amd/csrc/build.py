"""Build libmmlf_hip.so (gfx950) in-tree with hipcc.  `python -m mmlf_amd.csrc.build`."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
SOURCES = ['conv.hip', 'elementwise.hip']
LIB = os.path.join(HERE, 'libmmlf_hip.so')
HIPCC = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
# -ffp-contract=off: elementwise kernels restate float32 expressions of the reference op by op
# (mul, mul, add); fused multiply-adds appear only where written (fmaf / MFMA).
FLAGS = ['-O3', '--offload-arch=gfx950', '-fPIC', '-std=c++17', '-Wall', '-Wno-unused-function', '-ffp-contract=off']


def stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = SOURCES + ['common.h', 'build.py', os.path.join('..', '..', 'include', 'mmlf_hip.h')]
    return any(os.path.getmtime(os.path.join(HERE, d)) > t for d in deps)


def build(force=False, verbose=True):
    if not force and not stale():
        return LIB
    objs = []
    for src in SOURCES:
        obj = os.path.join(HERE, src.replace('.hip', '.o'))
        cmd = [HIPCC, *FLAGS, '-c', os.path.join(HERE, src), '-o', obj]
        if verbose:
            print(' '.join(cmd), flush=True)
        subprocess.check_call(cmd)
        objs.append(obj)
    cmd = [HIPCC, '--offload-arch=gfx950', '-shared', '-fPIC', *objs, '-o', LIB]
    if verbose:
        print(' '.join(cmd), flush=True)
    subprocess.check_call(cmd)
    return LIB


if __name__ == '__main__':
    build(force='--force' in sys.argv)
