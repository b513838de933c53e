"""Round-6 timing proxy for a FUSED evaluation stream block (reference mmlf/model/feed_forward.py:139-157 inside
mmlf/model/ensamble.py:61-76): the two convolutions of a 70-channel stream block on full 512 x 512 frames, as the Ensamble runs them
(register-streamed kernel conv4tap_rs_kernel<5, 9, RELU, TR>), each launch on properly formed inputs of its own:

  conv1  x (H, W)       -> y (H + 1, W + 1)   pad 1, bias + ReLU
  conv2  y (H + 1, W + 1) -> out (H, W)       pad 0, folded BatchNorm bias + ReLU

Run once with the product library and once with a -DMMLF_ABL_RS_FUSE=1 build (MMLF_HIP_LIB=variants/lib_rsfuse.so
MMLF_ALLOW_ABLATION=1), in which conv1 stores nothing and conv2 reads every group's activations from its wave's first group (cache
hits: real values, no memory traffic).  The sum of the two ablated launches is the LEAST a fused conv1 + ReLU + conv2 kernel could
take -- the intermediate's exchange through LDS, the 1-row / 1-column recomputation of a 2-D tile and the streaming of the second
filter (both do not fit the LDS) are not counted.  An earlier form of this proxy inside the whole ESE pipeline was confounded:
skipping stores / loads changes the tensors every later launch multiplies, and the matrix cores' power -- hence the clock -- follows
the data (profiles/r06_ab_ese_fuse_confounded.log).

    python tools/kbench_fuse.py [members=8] [reps=9]
"""
import os
import sys

import torch

sys.path.insert(0, os.getcwd())
from mmlf_amd import engine, _lib  # noqa: E402

dev = torch.device('cuda:0')
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
REPS = int(sys.argv[2]) if len(sys.argv) > 2 else 9
H = W = 512
geo = engine.Geometry(B, H, W)


def timeit(fn):
    fn(); fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(REPS):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    ts.sort()
    return ts[len(ts) // 2], ts[0]


def grid_rand(cs, c, h, w, off):
    t = geo.buf(cs, dev)
    v = t[:geo.NQ * cs].view(B, geo.R, geo.P, cs)
    v.zero_()
    v[:, off:off + h, off:off + w, :c] = torch.randn((B, h, w, c), device=dev).clamp_(min=0)      # behind a ReLU: half zeros
    t.absmax = geo.amax_of(t, cs)
    return t


cin = cout = 70
cs = engine.cs_of(cin)
w = torch.randn(cout, cin, 2, 2, device=dev) * 0.03
b = torch.randn(cout, device=dev) * 0.1
pk = engine.pack_filter(w, 0, False)
x = grid_rand(cs, cin, H, W, 1)
y = grid_rand(cs, cout, H + 1, W + 1, 0)
o1, o0 = geo.buf(cs, dev), geo.buf(cs, dev)
t1 = timeit(lambda: engine.conv(geo, x, cs, cin, pk, b, cout, o1, cs, 0, H + 1, W + 1, True))
t0 = timeit(lambda: engine.conv(geo, y, cs, cout, pk, b, cout, o0, cs, geo.P + 1, H, W, True))
tag = os.environ.get('MMLF_HIP_LIB', 'product')
scale = 70.0 / B * 12                    # launches of each kind per 70-member scene: 4 streams x 3 blocks (block 0's conv1 is 27 -> 70)
print(f'{tag:28s} {B} members 512x512 70->70: conv1 (pad 1) {t1[0]:6.3f} ms (min {t1[1]:6.3f})  conv2 (pad 0) {t0[0]:6.3f} ms (min {t0[1]:6.3f})  '
      f'pair {t1[0] + t0[0]:6.3f} ms  => per 70-member scene, 12 blocks: {(t1[0] + t0[0]) * scale:6.1f} ms   [{_lib.build_info().split("MMLF_ABL_RS_FUSE=")[1].split()[0]}=MMLF_ABL_RS_FUSE]', flush=True)
