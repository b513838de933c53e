"""Where a conv wave spends its cycles: needs a library built from a conv.hip instrumented with s_memtime at five points of
the chunk loop (a ConvArgs::tl_out pointer taken from MMLF_TL_PTR): tools/timeline_conv.patch, applied to a COPY of
mmlf_amd/csrc/conv.hip and linked with elementwise.o into a scratch library; MMLF_HIP_LIB points this script at it.
DESIGN.md section 4.6 has the numbers."""
import os, sys, numpy as np, torch
sys.path.insert(0, os.getcwd())
os.environ.setdefault('MMLF_HIP_LIB', os.path.join(os.getcwd(), 'scratch/tl/lib_tl.so'))
from mmlf_amd import engine, _lib
dev = torch.device('cuda:0')
B, H, W = 512, 96, 96
geo = engine.Geometry(B, H, W)
def grid_rand(cs, c, h, w, off, relu=False):
    t = geo.buf(cs, dev)
    v = t[:geo.NQ * cs].view(B, geo.R, geo.P, cs); v.zero_()
    r = torch.randn((B, h, w, c), device=dev)
    v[:, off:off + h, off:off + w, :c] = r.clamp_(min=0) if relu else r
    t.absmax = geo.amax_of(t, cs)
    return t
for cin, cout in ((280, 280), (70, 70)):
    cs_in, cs_out = engine.cs_of(cin), engine.cs_of(cout)
    w = torch.randn(cout, cin, 2, 2, device=dev) * 0.03
    b = torch.randn(cout, device=dev) * 0.1
    pk = engine.pack_filter(w, 0, False)
    x = grid_rand(cs_in, cin, H, W, 1, relu=True)
    out = geo.buf(cs_out, dev)
    tl = torch.zeros(512 * 16 * 6, dtype=torch.int64, device=dev)
    for rep in range(3):
        tl.zero_()
        os.environ['MMLF_TL_PTR'] = str(tl.data_ptr())
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        engine.conv(geo, x, cs_in, cin, pk, b, cout, out, cs_out, 0, H + 1, W + 1, True)
        e1.record(); torch.cuda.synchronize()
    os.environ.pop('MMLF_TL_PTR')
    t = tl.cpu().numpy().reshape(-1, 6).astype(np.float64)
    t = t[t[:, 5] > 0]
    tot = t[:, :5].sum(1)
    print(f'{cin}->{cout}: {e0.elapsed_time(e1):.3f} ms, waves {len(t)}, chunks/wave {t[:,5].mean():.0f}')
    names = ['head(reads+split)', 'mfma blocks', 'dma wait', 'epilogue', 'barrier']
    for k, n in enumerate(names):
        print(f'   {n:18s} {t[:,k].mean()/t[:,5].mean():9.1f} cycles/chunk  {100*t[:,k].sum()/tot.sum():5.1f} %   (min wave {100*(t[:,k]/tot).min():4.1f} %, max {100*(t[:,k]/tot).max():4.1f} %)')
    print(f'   total {tot.mean()/t[:,5].mean():9.1f} cycles/chunk')
