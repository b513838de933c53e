"""What the 280 -> 280 forward launch takes when its activation operand arrives ALREADY split: per position
and 8-channel octet 16 bytes of f16 `hi` followed by 16 bytes of `lo` (scale 2^10), in the 4 bytes per element the float32
tensor takes -- against the production kernel on the float32 tensor.  Needs variants/lib_presplit.so
(tools/build_variant.sh presplit -DMMLF_ABL_PRESPLIT=1); checks the result against the production kernel's.
    MMLF_HIP_LIB=variants/lib_presplit.so python tools/presplit_bench.py   |   python tools/presplit_bench.py"""
import os
import sys
import torch
sys.path.insert(0, os.getcwd())
from mmlf_amd import engine
dev = torch.device('cuda:0')
PRE = 'presplit' in os.environ.get('MMLF_HIP_LIB', '')
B, H, W, C = 512, 96, 96, 280
geo = engine.Geometry(B, H, W)
cs = engine.cs_of(C)
torch.manual_seed(0)
x = geo.buf(cs, dev)
v = x[:geo.NQ * cs].view(B, geo.R, geo.P, cs)
v.zero_()
v[:, 1:1 + H, 1:1 + W, :C] = torch.randn((B, H, W, C), device=dev).clamp_(min=0)
x.absmax = geo.amax_of(x, cs)
w = torch.randn(C, C, 2, 2, device=dev) * 0.03
b = torch.randn(C, device=dev) * 0.1
pk = engine.pack_filter(w, 0, False)
out = geo.buf(cs, dev)
src = x
if PRE:      # (positions, octets, [hi x 8 | lo x 8]) f16, viewed as the float32 buffer the kernel is handed
    S = 1024.0
    n = x.numel() // 8
    src = geo.buf(cs, dev)
    for s0 in range(0, n, 1 << 24):             # in pieces: the temporaries are as large as the tensor
        xs = x[8 * s0:8 * min(n, s0 + (1 << 24))].view(-1, 8) * S
        hi = xs.half()
        lo = (xs - hi.float()).half()
        src[8 * s0:8 * s0 + xs.numel()].view(torch.float16).view(-1, 16)[:] = torch.cat([hi, lo], 1)
    src.absmax = x.absmax
run = lambda: engine.conv(geo, src, cs, C, pk, b, C, out, cs, 0, H + 1, W + 1, True)
run(); run()
torch.cuda.synchronize()
ts = []
for _ in range(9):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); run(); e1.record(); torch.cuda.synchronize()
    ts.append(e0.elapsed_time(e1))
ts.sort()
fl = 2.0 * B * (H + 1) * (W + 1) * C * 4 * C
chk = out[:geo.NQ * cs].view(B, geo.R, geo.P, cs)[3, 10:20, 10:20, :8].double()
print(f'{"presplit" if PRE else "production"} 280->280 fwd bs=512: median {ts[4]:.3f} ms  min {ts[0]:.3f} ms  {fl / ts[4] / 1e9:.1f} TFLOP/s  '
      f'checksum {float(chk.sum()):.6f} absmean {float(out.abs().mean()):.6f}', flush=True)
