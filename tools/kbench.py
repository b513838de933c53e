"""Kernel micro-benchmark for A/B runs on one box: the dominant launches of a bs=512 step, each timed with HIP
events, median over reps.  MMLF_HIP_LIB selects the library build.   python tools/kbench.py [B] [reps] [tag]"""
import os
import sys
import torch
sys.path.insert(0, os.getcwd())
from mmlf_amd import engine, _lib
from mmlf_amd._lib import call, ptr
dev = torch.device('cuda:0')
B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
REPS = int(sys.argv[2]) if len(sys.argv) > 2 else 7
TAG = sys.argv[3] if len(sys.argv) > 3 else os.environ.get('MMLF_HIP_LIB', 'default')
H = W = 96
geo = engine.Geometry(B, H, W)
NEW = hasattr(geo, 'amax_n')


def amax_for(t, cs):
    return geo.amax_of(t, cs) if NEW else t.abs().max().reshape(1)


ZEROS = float(os.environ['KBENCH_ZEROS']) if os.environ.get('KBENCH_ZEROS') else None
STAMPS = []      # KBENCH_STAMP=1: wall-clock windows of every timed loop (tools/power_kernels.sh aligns rocm-smi samples with them)


def timeit(fn):
    if os.environ.get('KBENCH_STAMP'):          # progress on stderr: which loop a fault or a hang belongs to
        print(f'kbench: loop {len(STAMPS)} starts', file=sys.stderr, flush=True)
    fn(); fn()
    torch.cuda.synchronize()
    ts = []
    import time
    t_begin = time.time()
    for _ in range(REPS):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    STAMPS.append((t_begin, time.time()))
    ts.sort()
    return ts[len(ts) // 2], ts[0]


def grid_rand(cs, c, h, w, off, relu=False):
    t = geo.buf(cs, dev)
    v = t[:geo.NQ * cs].view(B, geo.R, geo.P, cs)
    v.zero_()
    r = torch.randn((B, h, w, c), device=dev)
    if ZEROS is not None:           # KBENCH_ZEROS=p: a fraction p of every operand tensor is exact zeros (matrix-core power is data dependent)
        r = r.abs_() if relu else r
        r[torch.rand_like(r) < ZEROS] = 0
    elif relu:
        r = r.clamp_(min=0)
    v[:, off:off + h, off:off + w, :c] = r
    t.absmax = amax_for(t, cs)
    return t


def run(cin, cout):
    cs_in, cs_out = engine.cs_of(cin), engine.cs_of(cout)
    w = torch.randn(cout, cin, 2, 2, device=dev) * 0.03
    b = torch.randn(cout, device=dev) * 0.1
    pk, pkd = engine.pack_filter(w, 0, False), engine.pack_filter(w, 0, True)
    x = grid_rand(cs_in, cin, H, W, 1, relu=True)          # block input (extent H,W at (1,1))
    y = grid_rand(cs_out, cout, H + 1, W + 1, 0, relu=True)  # conv1 output
    out1, out0 = geo.buf(cs_out, dev), geo.buf(cs_out, dev)
    ws = engine._Workspace.get(dev)
    fl1 = 2.0 * B * (H + 1) * (W + 1) * cout * 4 * cin
    fl0 = 2.0 * B * H * W * cout * 4 * cin
    res = {}
    res['fwd_p1'] = (timeit(lambda: engine.conv(geo, x, cs_in, cin, pk, b, cout, out1, cs_out, 0, H + 1, W + 1, True)), fl1)
    if NEW and hasattr(geo, 'relu_mask'):
        mask = geo.relu_mask(dev)
        res['fwd_p1_mask'] = (timeit(lambda: engine.conv(geo, x, cs_in, cin, pk, b, cout, out1, cs_out, 0, H + 1, W + 1, True, mask_out=mask)), fl1)
    res['fwd_p0_stats'] = (timeit(lambda: engine.conv(geo, y, cs_out, cout, pk, b, cout, out0, cs_out, geo.P + 1, H, W, False,
                                                        bn_partial=ws.partial)), fl0)
    g0 = grid_rand(cs_out, cout, H, W, 1)                     # dz
    res['dgrad_p0_ref'] = (timeit(lambda: engine.conv(geo, g0, cs_out, cout, pkd, None, cin, out1, cs_in, 0, H + 1, W + 1, False,
                                                       ref=y, cs_ref=cs_out)), fl1)
    if NEW and hasattr(geo, 'relu_mask'):
        res['dgrad_p0_bits'] = (timeit(lambda: engine.conv(geo, g0, cs_out, cout, pkd, None, cin, out1, cs_in, 0, H + 1, W + 1, False,
                                                            mask_in=mask)), fl1)
    g1 = grid_rand(cs_out, cout, H + 1, W + 1, 0)             # dy
    res['dgrad_p1'] = (timeit(lambda: engine.conv(geo, g1, cs_out, cout, pkd, None, cin, out0, cs_in, geo.P + 1, H, W, False)), fl0)
    gw, gb = torch.zeros_like(w), torch.zeros(cout, device=dev)
    wsb = ws.wgrad_ws(geo, cin, cout) if NEW else ws.wgrad_ws(cin, cout)
    res['wgrad_p1'] = (timeit(lambda: engine.wgrad(geo, x, cs_in, cin, g1, cs_out, cout, 0, gw, gb, 0, wsb)), fl1)
    res['wgrad_p0'] = (timeit(lambda: engine.wgrad(geo, y, cs_out, cout, g0, cs_out, cout, geo.P + 1, gw, gb, 0, wsb)), fl0)
    for (k, ((med, mn), fl)), (t0, t1) in zip(res.items(), STAMPS[-len(res):]):
        stamp = f'  window {t0:.3f} {t1:.3f}' if os.environ.get('KBENCH_STAMP') else ''
        print(f'{TAG} {cin}->{cout} B={B} {k:14s} median {med:8.3f} ms  min {mn:8.3f} ms  {fl / med / 1e9:7.1f} TFLOP/s{stamp}', flush=True)


def run_wgrad(cin, cout):
    """weight gradient of a pad-1 convolution cin -> cout alone (shapes with cin != cout: the DPP head)"""
    cs_in, cs_out = engine.cs_of(cin), engine.cs_of(cout)
    x = grid_rand(cs_in, cin, H, W, 1, relu=True)
    g1 = grid_rand(cs_out, cout, H + 1, W + 1, 0)
    gw, gb = torch.zeros(cout, cin, 2, 2, device=dev), torch.zeros(cout, device=dev)
    wsb = engine._Workspace.get(dev).wgrad_ws(geo, cin, cout)
    med, mn = timeit(lambda: engine.wgrad(geo, x, cs_in, cin, g1, cs_out, cout, 0, gw, gb, 0, wsb))
    fl = 2.0 * B * (H + 1) * (W + 1) * cout * 4 * cin
    print(f'{TAG} {cin}->{cout} B={B} wgrad_p1       median {med:8.3f} ms  min {mn:8.3f} ms  {fl / med / 1e9:7.1f} TFLOP/s', flush=True)


if os.environ.get("KBENCH_SHAPES"):            # e.g. KBENCH_SHAPES=280x108,108x108 (the DPP head): weight gradients only
    for sh in os.environ["KBENCH_SHAPES"].split(","):
        run_wgrad(*[int(v) for v in sh.split("x")])
    sys.exit(0)
if os.environ.get("KBENCH_ONLY70", "0") == "0": run(280, 280)
if os.environ.get("KBENCH_ONLY280", "0") == "0": run(70, 70)
