import sys, time, torch
sys.path.insert(0, '.')
from mmlf_amd import engine, _lib
from mmlf_amd._lib import call, ptr
dev = torch.device('cuda:0')
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
H = W = 96
geo = engine.Geometry(B, H, W)
def bench(cin, cout, pad, reps=10):
    cs_in, cs_out = engine.cs_of(cin), engine.cs_of(cout)
    x = torch.randn(geo.alloc * cs_in, device=dev)
    w = torch.randn(cout, cin, 2, 2, device=dev) * 0.05
    if len(sys.argv) > 2:
        x.zero_(); w.zero_()
    b = torch.randn(cout, device=dev)
    pk = engine.pack_filter(w, 0, False)
    x.absmax = geo.amax_of(x, cs_in)
    out = torch.zeros(geo.alloc * cs_out, device=dev)
    shift, vh, vw = (0, H + 1, W + 1) if pad else (geo.P + 1, H, W)
    for _ in range(2):
        engine.conv(geo, x, cs_in, cin, pk, b, cout, out, cs_out, shift, vh, vw, True)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        engine.conv(geo, x, cs_in, cin, pk, b, cout, out, cs_out, shift, vh, vw, True)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    fl = 2.0 * B * vh * vw * cout * 4 * cin
    print(f'conv {cin}->{cout} pad{pad} B={B}: {ms:.3f} ms  {fl/ms/1e9:.1f} TFLOP/s algorithmic', flush=True)
    # wgrad
    g = torch.randn(geo.alloc * cs_out, device=dev)
    gw = torch.zeros(cout, cin, 2, 2, device=dev); gb = torch.zeros(cout, device=dev)
    ws = torch.empty(int(_lib.load().mmlf_wgrad_workspace_floats(cin, cout, B, H, W)), device=dev)
    x.absmax, g.absmax = geo.amax_of(x, cs_in), geo.amax_of(g, cs_out)     # what the producers maintain
    for _ in range(2):
        engine.wgrad(geo, x, cs_in, cin, g, cs_out, cout, shift, gw, gb, 0, ws)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(reps):
        engine.wgrad(geo, x, cs_in, cin, g, cs_out, cout, shift, gw, gb, 0, ws)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    print(f'wgrad {cin}->{cout} pad{pad} B={B}: {ms:.3f} ms  {fl/ms/1e9:.1f} TFLOP/s algorithmic', flush=True)
ZERO = len(sys.argv) > 2
bench(280, 280, 1)
bench(280, 280, 0)
bench(70, 70, 1)
bench(27, 70, 1)
bench(560, 280, 0)
bench(1120, 280, 0)
