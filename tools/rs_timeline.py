"""Cycles per phase of a 32-position group in conv4tap_rs_kernel by the wave's own clock (s_memtime), 70 -> 70 forward at
bs=512.  Needs the diagnostic build:  tools/build_variant.sh rstl -DMMLF_RS_TIMELINE=1
    MMLF_HIP_LIB=variants/lib_rstl.so MMLF_CONV_RS=1 python tools/rs_timeline.py"""
import ctypes
import os
import sys
import numpy as np
import torch
sys.path.insert(0, os.getcwd())
from mmlf_amd import engine, _lib
dev = torch.device('cuda:0')
B, H, W, C = 512, 96, 96, 70
geo = engine.Geometry(B, H, W)
cs = engine.cs_of(C)
x = geo.buf(cs, dev)
v = x[:geo.NQ * cs].view(B, geo.R, geo.P, cs)
v.zero_()
v[:, 1:1 + H, 1:1 + W, :C] = torch.randn((B, H, W, C), device=dev).clamp_(min=0)
x.absmax = geo.amax_of(x, cs)
w = torch.randn(C, C, 2, 2, device=dev) * 0.03
b = torch.randn(C, device=dev) * 0.1
pk = engine.pack_filter(w, 0, False)
out = geo.buf(cs, dev)
for _ in range(3):
    engine.conv(geo, x, cs, C, pk, b, C, out, cs, 0, H + 1, W + 1, True)
torch.cuda.synchronize()
lib = _lib.load()
n = 5 * 2048
host = (ctypes.c_ulonglong * n)()
fn = lib.mmlf_debug_rs_timeline
fn.argtypes = [ctypes.c_void_p, ctypes.c_int]
assert fn(ctypes.byref(host), n) == 0
t = np.array(host, dtype=np.float64).reshape(-1, 5)
t = t[t[:, 4] > 0]
per = t[:, :4] / t[:, 4:5]
tot = per.sum(1)
print(f'{len(t)} waves, {t[:, 4].mean():.1f} groups per wave, cycles per group by the wave\'s clock: total {tot.mean():.0f}')
for name, col in zip(('issue of the 36 loads', 'scale + wait for the first data', 'steps (waits, split, MFMAs)', 'epilogue'), per.T):
    print(f'  {name:34s} {col.mean():8.0f}  ({100 * col.mean() / tot.mean():4.1f} %)   per wave min {col.min():.0f} max {col.max():.0f}')
for half, sel in (('waves 0-3', np.arange(len(t)) % 8 < 4), ('waves 4-7', np.arange(len(t)) % 8 >= 4)):
    print(f'  {half}: total {tot[sel].mean():.0f}, steps {per[sel, 2].mean():.0f}')
