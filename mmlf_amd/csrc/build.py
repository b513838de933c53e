"""Build libmmlf_hip.so (gfx950) in-tree with hipcc.  `python -m mmlf_amd.csrc.build`."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
SOURCES = ['conv.hip', 'wgrad.hip', 'elementwise.hip']
LIB = os.path.join(HERE, 'libmmlf_hip.so')
HIPCC = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
# -ffp-contract=off: elementwise kernels restate float32 expressions of the reference op by op
# (mul, mul, add); fused multiply-adds appear only where written (fmaf / MFMA).
FLAGS = ['-O3', '--offload-arch=gfx950', '-fPIC', '-std=c++17', '-Wall', '-Wno-unused-function', '-ffp-contract=off']


def stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = SOURCES + ['common.h', 'conv_device.h', 'build.py', os.path.join('..', '..', 'include', 'mmlf_hip.h')]
    return any(os.path.getmtime(os.path.join(HERE, d)) > t for d in deps)


def source_revision():
    """short git hash of the tree the library is built from ('+' if the kernel sources differ from it), for mmlf_build_info()"""
    root = os.path.dirname(os.path.dirname(HERE))
    try:
        rev = subprocess.check_output(['git', '-C', root, 'rev-parse', '--short=12', 'HEAD'], stderr=subprocess.DEVNULL).decode().strip()
        dirty = subprocess.call(['git', '-C', root, 'diff', '--quiet', 'HEAD', '--', 'mmlf_amd/csrc', 'include'],
                                stderr=subprocess.DEVNULL) != 0
        return rev + ('+' if dirty else '')
    except (OSError, subprocess.CalledProcessError):
        return 'unknown'          # (no git on the GPU box's snapshot: the library that travels there was built here)


HASHED = SOURCES + ['common.h', 'conv_device.h', os.path.join('..', '..', 'include', 'mmlf_hip.h')]


def source_hash():
    """content hash of everything the kernels are compiled from: mmlf_build_info()'s `src=` field.  Unlike `git=` (the HEAD
    the library happened to be built at: build.stale() looks at file times, not at HEAD) it names the kernel sources
    themselves -- bench.py compares it between the library it times and the one the committed PMC passes were collected on."""
    import hashlib
    h = hashlib.sha1()
    for d in HASHED:
        with open(os.path.join(HERE, d), 'rb') as f:
            h.update(f.read())
    return h.hexdigest()[:12]


def build(force=False, verbose=True, extra_flags=(), lib=None):
    """extra_flags / lib: another build of the same sources somewhere else (tests/test_gpu_bounds.py: -DMMLF_BOUNDS_DEBUG)"""
    if lib is None and not extra_flags and not force and not stale():
        return LIB
    lib = lib or LIB
    tag = '' if lib == LIB else '.' + os.path.basename(lib).replace('.so', '')
    flags = [*FLAGS, f'-DMMLF_GIT_HASH="{source_revision()}"', f'-DMMLF_SRC_HASH="{source_hash()}"', *extra_flags]
    objs, jobs = [], []
    for src in SOURCES:                 # the translation units are independent: compiled side by side
        obj = os.path.join(os.path.dirname(lib), src.replace('.hip', tag + '.o'))
        cmd = [HIPCC, *flags, '-c', os.path.join(HERE, src), '-o', obj]
        if verbose:
            print(' '.join(cmd), flush=True)
        jobs.append((cmd, subprocess.Popen(cmd)))
        objs.append(obj)
    failed = [cmd for cmd, job in jobs if job.wait() != 0]
    if failed:
        raise subprocess.CalledProcessError(1, failed[0])
    cmd = [HIPCC, '--offload-arch=gfx950', '-shared', '-fPIC', *objs, '-o', lib]
    if verbose:
        print(' '.join(cmd), flush=True)
    subprocess.check_call(cmd)
    return lib


if __name__ == '__main__':
    if '--source-hash' in sys.argv:
        print(source_hash())
    elif '--source-revision' in sys.argv:
        print(source_revision())
    else:
        build(force='--force' in sys.argv)
