"""Round-6 proxy: what are WHOLE-LINE activation reads worth on the TILED 70 -> 70 kernel?

conv4tap_x6s_kernel<5, 2, *, 16> fetches a chunk of 8 channels (32 bytes) of 32 positions per LDS-DMA piece: in NHWC (72 floats
= 288 bytes per position) that is 32 sector-sized reads 288 bytes apart, and the four chunks that share a 128-byte line come by
at different times -- an XCD's 32 resident 176 KB windows do not fit its 4 MB L2 between the visits, so the launch reads
2.38-2.46 GB for a 1.39 GB tensor (profiles/r05_pmc_exact_bytes.json, r05_pmc_narrow_vs_cus.log).

Here the SAME kernel (MMLF_PROXY_BLOCKED_A=1) reads its A pieces from a chunk-blocked copy of the input,
[tile of 512 positions][chunk of 8 channels][position][8 channels]: a piece is one contiguous 1 KiB run (8 whole lines), one
chunk of a 611-position window is 19.5 KB of whole lines, every line crosses the fabric once per window.  The copy is made
outside the timed launch; the results must be BIT-IDENTICAL to the NHWC launch (it is a layout change only) and are checked.
This measures the consumer-side bound of a blocked activation layout on the tiled kernel -- request rate and contention effects
included -- before any producer is rewritten for it.

    python tools/kbench_blocked.py [B=512] [reps=9]
"""
import os
import sys

import torch

sys.path.insert(0, os.getcwd())
from mmlf_amd import engine  # noqa: E402

dev = torch.device('cuda:0')
B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
REPS = int(sys.argv[2]) if len(sys.argv) > 2 else 9
H = W = 96
TILE = 512
geo = engine.Geometry(B, H, W)
assert engine.CONV_MODE == 'f16x3'


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


def grid_rand(cs, c, h, w, off, relu=False):
    t = geo.buf(cs, dev)
    v = t[:geo.NQ * cs].view(B, geo.R, geo.P, cs)
    v.zero_()
    r = torch.randn((B, h, w, c), device=dev)
    if relu:
        r = r.clamp_(min=0)
    v[:, off:off + h, off:off + w, :c] = r
    t.absmax = geo.amax_of(t, cs)
    return t


def blocked_copy(t, cs):
    """[position][cs] -> [tile][chunk][position in tile][8]; one more (zero) tile behind the last for its halo"""
    ntiles = -(-geo.NQ // TILE)                                 # (the library pads the grid to a multiple of 512 positions)
    n = (ntiles + 1) * TILE
    src = torch.zeros(n * cs, dtype=torch.float32, device=dev)
    m = min(t.numel(), src.numel())
    src[:m] = t[:m]
    out = src.view(ntiles + 1, TILE, cs // 8, 8).permute(0, 2, 1, 3).contiguous().view(-1)
    out.absmax = t.absmax
    return out


def main():
    cin = cout = 70
    cs = engine.cs_of(cin)
    w = torch.randn(cout, cin, 2, 2, device=dev) * 0.03
    b = torch.randn(cout, device=dev) * 0.1
    pk, pkd = engine.pack_filter(w, 0, False), engine.pack_filter(w, 0, True)
    ws = engine._Workspace.get(dev)
    P = geo.P
    x = grid_rand(cs, cin, H, W, 1, relu=True)            # block input: extent (H, W) at (1, 1)
    y = grid_rand(cs, cout, H + 1, W + 1, 0, relu=True)   # conv1 output: extent (H + 1, W + 1) at (0, 0)
    g0 = grid_rand(cs, cout, H, W, 1)                     # dz
    g1 = grid_rand(cs, cout, H + 1, W + 1, 0)             # dy
    mask = geo.relu_mask(dev)
    kinds = {
        'fwd_p1':        (x,  lambda inp, out: engine.conv(geo, inp, cs, cin, pk, b, cout, out, cs, 0, H + 1, W + 1, True)),
        'fwd_p1_mask':   (x,  lambda inp, out: engine.conv(geo, inp, cs, cin, pk, b, cout, out, cs, 0, H + 1, W + 1, True, mask_out=mask)),
        'fwd_p0_stats':  (y,  lambda inp, out: engine.conv(geo, inp, cs, cout, pk, b, cout, out, cs, P + 1, H, W, False, bn_partial=ws.partial)),
        'dgrad_p0_bits': (g0, lambda inp, out: engine.conv(geo, inp, cs, cout, pkd, None, cin, out, cs, 0, H + 1, W + 1, False, mask_in=mask)),
        'dgrad_p1':      (g1, lambda inp, out: engine.conv(geo, inp, cs, cout, pkd, None, cin, out, cs, P + 1, H, W, False)),
    }
    alg_bytes = {k: 4.0 * B * 70 * ((H * W) + (H + 1) * (W + 1)) for k in kinds}
    print(f'# tools/kbench_blocked.py: 70->70, B={B}, {H}x{W}, median of {REPS} (min); NHWC input against a chunk-blocked copy '
          f'[tile 512][chunk 8 ch][position][8 ch], same kernel conv4tap_x6s_kernel<5, 2, EPI, 16>', flush=True)
    tot = [0.0, 0.0]
    for name, (inp, fn) in kinds.items():
        out_a, out_b = geo.buf(cs, dev), geo.buf(cs, dev)
        os.environ.pop('MMLF_PROXY_BLOCKED_A', None)
        fn(inp, out_a)                                     # (also (re)writes the mask the dgrad kind reads)
        ta = timeit(lambda: fn(inp, out_a))
        inb = blocked_copy(inp, cs)
        os.environ['MMLF_PROXY_BLOCKED_A'] = '1'
        try:
            fn(inb, out_b)
            torch.cuda.synchronize()
            same = torch.equal(out_a[:geo.NQ * cs + (P + 1) * cs], out_b[:geo.NQ * cs + (P + 1) * cs])
            tb = timeit(lambda: fn(inb, out_b))
        finally:
            os.environ.pop('MMLF_PROXY_BLOCKED_A', None)
        gb = alg_bytes[name] / 1e9
        tot[0] += ta[0]; tot[1] += tb[0]
        print(f'{name:14s} NHWC {ta[0]:6.3f} ms ({ta[1]:6.3f})  {gb / ta[0] * 1e3:6.0f} GB/s   blocked {tb[0]:6.3f} ms ({tb[1]:6.3f})  '
              f'{gb / tb[0] * 1e3:6.0f} GB/s   {100 * (tb[0] / ta[0] - 1):+5.1f} %   bit-identical {same}', flush=True)
        assert same, name
        del out_a, out_b, inb
    print(f'sum of the five launch kinds: NHWC {tot[0]:.3f} ms, blocked {tot[1]:.3f} ms, {100 * (tot[1] / tot[0] - 1):+.1f} %')


if __name__ == '__main__':
    main()
