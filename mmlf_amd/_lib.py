"""ctypes binding of libmmlf_hip.so (the C ABI declared in include/mmlf_hip.h).

The product path has no CPU fallback: if the library is missing or a call fails, a
RuntimeError is raised (with the text from ``mmlf_last_error()``).
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# MMLF_HIP_LIB selects another build of the same ABI (kernel A/B experiments); never a fallback
LIB_PATH = os.environ.get('MMLF_HIP_LIB') or os.path.join(_HERE, 'csrc', 'libmmlf_hip.so')

_vp = ctypes.c_void_p
_i = ctypes.c_int
_i64 = ctypes.c_int64
_d = ctypes.c_double

# name -> (restype, argtypes); must list every symbol of include/mmlf_hip.h
SIGNATURES = {
    'mmlf_last_error': (ctypes.c_char_p, []),
    'mmlf_abi_version': (_i, []),
    'mmlf_build_info': (ctypes.c_char_p, []),
    'mmlf_build_is_ablation': (_i, []),
    'mmlf_conv_cus': (_i, []),
    'mmlf_audit_conv_h2': (_i, [_i] * 10 + [_vp]),
    'mmlf_audit_wgrad_h2': (_i, [_i] * 8 + [_vp]),
    'mmlf_grid_alloc_positions': (_i64, [_i, _i, _i]),
    'mmlf_packed_filter_floats': (_i64, [_i, _i]),
    'mmlf_wgrad_workspace_floats': (_i64, [_i, _i, _i, _i, _i]),
    'mmlf_pack_filter': (_i, [_vp, _vp, _i, _i, _i, _i, _vp]),
    'mmlf_conv2x2': (_i, [_vp, _i, _i, _vp, _vp, _i, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _i, _vp, _i, _vp]),
    'mmlf_packed_filter_split_bytes': (_i64, [_i, _i]),
    'mmlf_pack_filter_split': (_i, [_vp, _vp, _i, _i, _i, _i, _vp]),
    'mmlf_conv2x2_split': (_i, [_vp, _i, _i, _vp, _vp, _i, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _i, _vp, _i, _vp]),
    'mmlf_conv2x2_wgrad_h2': (_i, [_vp, _i, _i, _vp, _i, _i, _i, _vp, _vp, _i, _i, _vp, _i, _i, _i, _vp, _vp, _vp]),
    'mmlf_packed_filter_h2_bytes': (_i64, [_i, _i]),
    'mmlf_amax_entries': (_i64, [_i, _i, _i]),
    'mmlf_grid_pad_w': (_i, []),
    'mmlf_grid_pad_h': (_i, []),
    'mmlf_amax_head': (_i, []),
    'mmlf_amax_shard_stride': (_i, []),
    'mmlf_pack_filter_h2': (_i, [_vp, _vp, _i, _i, _i, _i, _vp]),
    'mmlf_packed_filter_h2_columns': (_i, [_i]),
    'mmlf_pack_filters_h2': (_i, [_vp, _i, _i, _vp]),
    'mmlf_conv2x2_h2': (_i, [_vp, _i, _i, _vp, _vp, _i, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _i, _vp, _i, _vp, _vp, _vp, _vp, _vp, _vp]),
    'mmlf_relu_mask_words': (_i64, [_i, _i, _i]),
    'mmlf_conv2x2_thin_workspace_floats': (_i64, [_i, _i, _i]),
    'mmlf_conv2x2_thin': (_i, [_vp, _i, _i, _vp, _vp, _i, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _i, _vp, _vp, _vp]),
    'mmlf_conv2x2_wgrad_thin_workspace_floats': (_i64, [_i]),
    'mmlf_conv2x2_wgrad_thin': (_i, [_vp, _i, _i, _vp, _i, _i, _i, _vp, _vp, _i, _i, _vp, _i, _i, _i, _vp]),
    'mmlf_conv2x2_blocks': (_i, [_i, _i, _i, _i, _i]),
    'mmlf_bn_stats_finalize': (_i, [_vp, _i, _i, _vp, _vp, _vp, _vp, _d, _d, _vp, _vp, _vp, _vp, _i, _i, _i, _vp]),
    'mmlf_conv2x2_wgrad': (_i, [_vp, _i, _i, _vp, _i, _i, _i, _vp, _vp, _i, _i, _vp, _i, _i, _i, _vp]),
    'mmlf_conv2x2_wgrad_split': (_i, [_vp, _i, _i, _vp, _i, _i, _i, _vp, _vp, _i, _i, _vp, _i, _i, _i, _vp]),
    'mmlf_bn_stats_train': (_i, [_vp, _i, _i, _vp, _vp, _vp, _vp, _d, _d, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    'mmlf_bn_coeffs_eval': (_i, [_vp, _vp, _vp, _vp, _d, _vp, _vp, _i, _vp]),
    'mmlf_fold_bn_eval': (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _vp]),
    'mmlf_bn_apply_relu': (_i, [_vp, _i, _i, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp, _vp]),
    'mmlf_bn_apply_relu4': (_i, [_vp, _i, _i, _vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp]),
    'mmlf_bn_bwd_reduce': (_i, [_vp, _i, _i, _vp, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _vp, _vp, _i, _i, _i, _i, _vp]),
    'mmlf_bn_bwd_apply': (_i, [_vp, _i, _i, _vp, _i, _i, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp]),
    'mmlf_pack_nchw': (_i, [_vp, _i, _vp, _i, _i, _i, _i, _vp, _vp]),
    'mmlf_zero_slack': (_i, [_vp, _i, _i, _i, _i, _vp, _vp]),
    'mmlf_zero_slack4': (_i, [_vp, _vp, _vp, _i, _i, _i, _vp]),
    'mmlf_unpack_nchw': (_i, [_vp, _i, _vp, _i, _i, _i, _i, _vp]),
    'mmlf_head_upr': (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    'mmlf_head_dpp': (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    'mmlf_head_upr_bwd': (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    'mmlf_head_dpp_bwd': (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    'mmlf_loss_fwd_bwd': (_i, [_i, _vp, _i, _vp, _vp, _vp, _d, _vp, _vp, _vp, _i, _vp, _i, _i, _i, _vp]),
    'mmlf_loss_multi_scratch_doubles': (_i64, [_i]),
    'mmlf_loss_multi_fwd_bwd': (_i, [_i, _vp, _i, _vp, _i, _vp, _vp, _vp, _d, _vp, _vp, _vp, _i, _vp, _vp, _i, _i, _i, _vp]),
    'mmlf_adam_step': (_i, [_vp, _vp, _vp, _vp, _i64, _d, _d, _d, _d, _i64, _d, _vp]),
    'mmlf_shift_views': (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    'mmlf_shift_pack': (_i, [_vp, _i, _vp, _i, _vp, _vp, _i, _i, _i, _i, _vp, _vp]),
    'mmlf_lmm_to_discrete': (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i64, _vp]),
    'mmlf_patch_gather': (_i, [_vp] * 5 + [_i] * 5 + [_vp] * 12 + [_i, _i, _vp]),
    'mmlf_patch_contrast': (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _vp]),
    'mmlf_ensamble_reduce': (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp]),
}

def _header_abi_version():
    """MMLF_ABI_VERSION as include/mmlf_hip.h defines it: ONE place holds the number (the library returns the same macro)"""
    import re
    path = os.path.join(os.path.dirname(_HERE), 'include', 'mmlf_hip.h')
    with open(path) as f:
        m = re.search(r'^#define\s+MMLF_ABI_VERSION\s+(\d+)', f.read(), re.M)
    if m is None:
        raise RuntimeError(f'{path}: no MMLF_ABI_VERSION')
    return int(m.group(1))


ABI_VERSION = _header_abi_version()     # bumped whenever an entry point's arguments change
_lib = None
BUILD_INFO = None


def validate(lib, path, environ=None):
    """Refuse a library this package must not call: another ABI version (its entry points would be called with shifted
    arguments) or a build that computes WRONG results by construction -- the timing ablations whose switches csrc/conv_device.h lists
    (`mmlf_build_is_ablation()`), unless MMLF_ALLOW_ABLATION=1 says the user wants exactly that (kernel A/B runs).
    `lib` is anything with the three entry points (the tests pass a stub).  Returns the build string."""
    environ = os.environ if environ is None else environ
    got = lib.mmlf_abi_version() if hasattr(lib, 'mmlf_abi_version') else None
    if got != ABI_VERSION:
        raise RuntimeError(f'{path} implements ABI version {got}, this package needs {ABI_VERSION}: rebuild it '
                           'with `python -m mmlf_amd.csrc.build --force`')
    if not hasattr(lib, 'mmlf_build_info') or not hasattr(lib, 'mmlf_build_is_ablation'):
        raise RuntimeError(f'{path} does not say what it was built with (no mmlf_build_info): rebuild it')
    info = lib.mmlf_build_info()
    info = info.decode() if isinstance(info, bytes) else str(info)
    if lib.mmlf_build_is_ablation() and environ.get('MMLF_ALLOW_ABLATION') != '1':
        raise RuntimeError(f'{path} is a timing-ablation build that computes WRONG results ({info}); it is refused unless '
                           'MMLF_ALLOW_ABLATION=1 is set (unset MMLF_HIP_LIB to use the product library)')
    return info


def load():
    """Load the shared library once; raise if it is absent (no fallback)."""
    global _lib, BUILD_INFO
    if _lib is None:
        # torch FIRST: it ships a HIP runtime of its own (torch/lib/libamdhip64.so) and this library is linked against the
        # system's (/opt/rocm/lib).  Loaded in the other order the process holds two runtimes and the first launch fails with
        # "no ROCm-capable device is detected" (seen when `python __graft_entry__.py smoke` loaded the library in build(), before
        # anything had imported torch); with torch's runtime already mapped the loader resolves this library's dependency to it.
        import torch  # noqa: F401
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f'{LIB_PATH} not found: build it with `python -m mmlf_amd.csrc.build` '
                '(the HIP path has no CPU fallback)')
        lib = ctypes.CDLL(LIB_PATH)
        for name in ('mmlf_abi_version', 'mmlf_build_info', 'mmlf_build_is_ablation'):
            if hasattr(lib, name):
                getattr(lib, name).restype, getattr(lib, name).argtypes = SIGNATURES[name]
        BUILD_INFO = validate(lib, LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)
            fn.restype = res
            fn.argtypes = args
        _lib = lib
    return _lib


def build_info():
    """the loaded library's mmlf_build_info() string (bench.py prints it in config.build)"""
    load()
    return BUILD_INFO


def last_error():
    return load().mmlf_last_error().decode()


def call(name, *args):
    """Invoke an int-returning entry point; nonzero status -> RuntimeError."""
    rc = getattr(load(), name)(*args)
    if rc != 0:
        raise RuntimeError(f'{name} failed: {last_error()}')


def ptr(t):
    """Device pointer of a torch tensor (None -> NULL)."""
    return None if t is None else t.data_ptr()


def stream_ptr():
    import torch
    return torch.cuda.current_stream().cuda_stream
