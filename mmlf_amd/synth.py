"""Deterministic synthetic weights / inputs shared by the golden-vector generator,
the parity tests, ``__graft_entry__.smoke`` and ``bench.py``.

Nothing here depends on torch's RNG: every array comes from
``numpy.random.RandomState(seed)`` (the legacy, stream-frozen generator), so the
same (hyper-parameters, seed) pair yields bit-identical tensors in the build
container (where the reference is importable and the goldens are produced) and on
the GPU box (where it is not).

The parameter tree mirrors the reference's ``state_dict`` key set
(reference mmlf/model/feed_forward.py:95-102,104-187).
"""
import numpy as np


def out_channels(model_uncert=False, model_discrete=False, model_views=9,
                 model_cross=False, **_):
    """Head width: 1 BASE / 2 UPR / steps DPP (reference feed_forward.py:179-183)."""
    if model_uncert:
        return 2
    if model_discrete:
        return (2 if model_cross else 4) * model_views * 3
    return 1


def param_spec(model_chs=70, model_in_blocks=3, model_out_blocks=8, model_views=9,
               model_uncert=False, model_discrete=False, model_cross=False, **_):
    """Ordered list of (state_dict key, shape, kind) for the default (k=2, BN) net.

    kind in {'conv_w', 'conv_b', 'bn_w', 'bn_b', 'bn_rm', 'bn_rv', 'bn_nbt'}.
    Order equals torch's ``state_dict()`` order of the reference module tree.
    """
    oc = out_channels(model_uncert, model_discrete, model_views, model_cross)
    spec = []

    def block(prefix, cin, cout, bn=True):
        spec.append((f'{prefix}.0.weight', (cout, cin, 2, 2), 'conv_w'))
        spec.append((f'{prefix}.0.bias', (cout,), 'conv_b'))
        spec.append((f'{prefix}.2.weight', (cout, cout, 2, 2), 'conv_w'))
        spec.append((f'{prefix}.2.bias', (cout,), 'conv_b'))
        if bn:
            spec.append((f'{prefix}.3.weight', (cout,), 'bn_w'))
            spec.append((f'{prefix}.3.bias', (cout,), 'bn_b'))
            spec.append((f'{prefix}.3.running_mean', (cout,), 'bn_rm'))
            spec.append((f'{prefix}.3.running_var', (cout,), 'bn_rv'))
            spec.append((f'{prefix}.3.num_batches_tracked', (), 'bn_nbt'))

    nets = ['in_net_hv'] + ([] if model_cross else ['in_net_id'])
    for net in nets:
        block(f'{net}.0', model_views * 3, model_chs)
        for i in range(1, model_in_blocks):
            block(f'{net}.{i}', model_chs, model_chs)
    c = (2 if model_cross else 4) * model_chs
    for i in range(model_out_blocks - 1):
        block(f'out_net.{i}', c, c)
    block(f'out_net.{model_out_blocks - 1}', c, oc, bn=False)
    # the head block is (conv c->oc, relu, conv oc->oc): fix second conv's cin
    k = f'out_net.{model_out_blocks - 1}.2.weight'
    spec = [(n, ((oc, oc, 2, 2) if n == k else s), kd) for n, s, kd in spec]
    return spec


def synth_state(spec, seed=0, trained_like=True):
    """name -> numpy array.  Conv weights/biases ~ U(-1/sqrt(fan_in), 1/sqrt(fan_in))
    (torch's default Conv2d bound); with ``trained_like`` the BN affine parameters and
    running statistics are non-trivial so that eval mode exercises them."""
    rs = np.random.RandomState(seed)
    out = {}
    fan_in = 1
    for name, shape, kind in spec:
        if kind == 'conv_w':
            fan_in = shape[1] * shape[2] * shape[3]
            b = 1.0 / np.sqrt(fan_in)
            out[name] = rs.uniform(-b, b, size=shape).astype(np.float32)
        elif kind == 'conv_b':
            b = 1.0 / np.sqrt(fan_in)
            out[name] = rs.uniform(-b, b, size=shape).astype(np.float32)
        elif kind == 'bn_w':
            out[name] = (rs.uniform(0.5, 1.5, size=shape) if trained_like
                         else np.ones(shape)).astype(np.float32)
        elif kind == 'bn_b':
            out[name] = (rs.uniform(-0.2, 0.2, size=shape) if trained_like
                         else np.zeros(shape)).astype(np.float32)
        elif kind == 'bn_rm':
            out[name] = (rs.uniform(-0.1, 0.1, size=shape) if trained_like
                         else np.zeros(shape)).astype(np.float32)
        elif kind == 'bn_rv':
            out[name] = (rs.uniform(0.05, 0.3, size=shape) if trained_like
                         else np.ones(shape)).astype(np.float32)
        elif kind == 'bn_nbt':
            out[name] = np.array(0, dtype=np.int64)
        else:  # pragma: no cover
            raise ValueError(kind)
    return out


def formula_state(shapes, seed=0):
    """name -> numpy array for ANY module tree, from its (name, shape) list alone: each tensor is drawn from its own
    RandomState(crc32(name) ^ seed), so the values do not depend on construction or iteration order.  Used for the
    flag combinations param_spec does not describe (e.g. --model_unet)."""
    import zlib
    out = {}
    for name, shape in shapes:
        rs = np.random.RandomState((zlib.crc32(name.encode()) ^ seed) & 0x7FFFFFFF)
        shape = tuple(shape)
        if name.endswith('num_batches_tracked'):
            out[name] = np.array(0, dtype=np.int64)
        elif name.endswith('running_var'):
            out[name] = rs.uniform(0.05, 0.3, size=shape).astype(np.float32)
        elif name.endswith('running_mean'):
            out[name] = rs.uniform(-0.1, 0.1, size=shape).astype(np.float32)
        elif len(shape) == 1:                 # bias or BatchNorm affine
            out[name] = rs.uniform(0.5, 1.5, size=shape).astype(np.float32) if 'weight' in name \
                else rs.uniform(-0.1, 0.1, size=shape).astype(np.float32)
        else:
            b = 1.0 / np.sqrt(np.prod(shape[1:]))
            out[name] = rs.uniform(-b, b, size=shape).astype(np.float32)
    return out


def synth_inputs(batch, ps, views=9, seed=0, ps_w=None):
    """Four EPI stacks U[0,1) of shape (B, views, 3, ps, ps_w), gt in [-2, 2),
    and an int32 loss mask (all ones; the train step applies the 11-px margin,
    reference train/cli.py:194)."""
    ps_w = ps if ps_w is None else ps_w
    rs = np.random.RandomState(1000 + seed)
    stacks = [rs.uniform(0.0, 1.0, size=(batch, views, 3, ps, ps_w)).astype(np.float32)
              for _ in range(4)]
    gt = (4.0 * rs.uniform(size=(batch, ps, ps_w)) - 2.0).astype(np.float32)
    mask = np.ones((batch, ps, ps_w), dtype=np.int32)
    return stacks, gt, mask


def synth_scene(seed, H, W, views=9, planes=2):
    """One cached scene tuple as reference hci4d.py:150-254 builds it:
    (h, v, i, d, center, gt, mpi, mask, index) with stacks (views,3,H,W) float32 in [0,1),
    gt in [-2,2), mpi (planes,5,H,W) float32 (plane 3 = weight, 4 = disparity), mask int64 {0,1}."""
    rs = np.random.RandomState(5000 + seed)
    stacks = [rs.uniform(0.0, 1.0, size=(views, 3, H, W)).astype(np.float32) for _ in range(4)]
    center = stacks[1][int(views / 2)].copy()
    gt = (4.0 * rs.uniform(size=(H, W)) - 2.0).astype(np.float32)
    mpi = rs.uniform(0.0, 1.0, size=(planes, 5, H, W)).astype(np.float32)
    mpi[:, 4] = (4.0 * mpi[:, 4] - 2.0).astype(np.float32)
    mask = (rs.uniform(size=(H, W)) > 0.2).astype(np.int64)
    return (*stacks, center, gt, mpi, mask, np.atleast_1d(seed))
