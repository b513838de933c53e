"""which side is right at bs=512?  weight gradient of one 280->280 k=2 pad-1 convolution on a (512,280,96,96) input:
torch full batch vs torch in 4 chunks of 128 (float32, then the chunk sum in float64) vs the native wgrad kernel"""
import os, sys, torch
import torch.nn.functional as F
sys.path.insert(0, os.getcwd())
dev = torch.device('cuda:0')
torch.manual_seed(0)
B, C, H, W = 512, 280, 96, 96
x = torch.randn(B, C, H, W, device=dev).clamp_(min=0)
w = (torch.randn(C, C, 2, 2, device=dev) * 0.03).requires_grad_(True)
g = torch.randn(B, C, H + 1, W + 1, device=dev)
y = F.conv2d(x, w, None, padding=1)
y.backward(g)
full = w.grad.clone(); w.grad = None
del y
acc = torch.zeros_like(full, dtype=torch.float64)
for s in range(0, B, 128):
    y = F.conv2d(x[s:s + 128], w, None, padding=1)
    y.backward(g[s:s + 128])
    acc += w.grad.double(); w.grad = None
    del y
rel = float((full.double() - acc).norm() / acc.norm())
print('torch full-batch vs 4 x 128 chunks: relative L2', rel, flush=True)
# native
from mmlf_amd import engine, _lib
geo = engine.Geometry(B, H, W)
cs = engine.cs_of(C)
def to_grid(t, h, w_, off):
    buf = geo.buf(cs, dev)
    v = buf[:geo.NQ * cs].view(B, geo.R, geo.P, cs)
    v.zero_()
    v[:, off:off + h, off:off + w_, :C] = t.permute(0, 2, 3, 1)
    buf.absmax = geo.amax_of(buf, cs)
    return buf
xg = to_grid(x, H, W, 1)
gg = to_grid(g, H + 1, W + 1, 0)
gw, gb = torch.zeros(C, C, 2, 2, device=dev), torch.zeros(C, device=dev)
ws = engine._Workspace.get(dev)
for mode in ('f16x3', 'f32'):
    engine.CONV_MODE = mode
    gw.zero_(); gb.zero_()
    engine.wgrad(geo, xg, cs, C, gg, cs, C, 0, gw, gb, 0, ws.wgrad_ws(geo, C, C))
    torch.cuda.synchronize()
    print(f'native {mode} vs chunks: relative L2', float((gw.double() - acc).norm() / acc.norm()),
          ' vs torch full:', float((gw.double() - full.double()).norm() / full.double().norm()), flush=True)
