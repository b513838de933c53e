"""Where do two arithmetic modes part?  Runs the G2 BASE train step (B=2, 96x96) in two modes and prints, launch by
launch in execution order, the relative L2 difference of every conv / dgrad output and weight gradient.
  python tools/mode_diverge.py [modeA modeB]"""
import os
import sys
import numpy as np
import torch
sys.path.insert(0, os.getcwd())
sys.path.insert(0, os.path.join(os.getcwd(), 'tests'))
from conftest import BASE_KW
from mmlf_amd import engine, synth, loss as L
from mmlf_amd.feed_forward import FeedForward

modes = sys.argv[1:3] if len(sys.argv) > 2 else ['f32', 'f16x3']
dev = torch.device('cuda:0')
stacks, gt, mask = synth.synth_inputs(2, 96, seed=8)
mask = torch.from_numpy(mask).int() * L.create_mask_margin(mask.shape, 11)
orig_conv, orig_wgrad = engine.conv, engine.wgrad
rec = {}
for mode in modes:
    log = rec[mode] = []

    def conv(geo, x, cs_in, K, packed, bias, N, out, cs_out, *a, **k):
        orig_conv(geo, x, cs_in, K, packed, bias, N, out, cs_out, *a, **k)
        log.append((f'conv K={K} N={N} shift={a[0]} relu={a[3]} ' + ','.join(kk for kk in k if k[kk] is not None and kk in ('ref', 'mask_in', 'mask_out', 'bn_partial')),
                    out[:geo.NQ * cs_out].clone(), x[:geo.NQ * cs_in].clone()))

    def wgrad(geo, x, cs_in, cin, g, cs_g, cout, g_shift, gw, gb, *a, **k):
        w0, b0 = gw.clone(), gb.clone()
        orig_wgrad(geo, x, cs_in, cin, g, cs_g, cout, g_shift, gw, gb, *a, **k)
        torch.cuda.synchronize()
        log.append((f'wgrad {cin}->{cout} g_shift={g_shift}', (gw - w0).reshape(-1).clone(), g[:geo.NQ * cs_g].clone()))

    engine.conv, engine.wgrad, engine.CONV_MODE, engine.OVERLAP_WGRAD = conv, wgrad, mode, False
    m = FeedForward(**BASE_KW)
    m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in synth.synth_state(synth.param_spec(**BASE_KW), seed=21).items()})
    m.to(dev).train()
    out = m(*[torch.from_numpy(s).to(dev) for s in stacks])
    nfwd = len(log)
    val = L.MaskedL1Loss()(out, torch.from_numpy(gt).to(dev), mask.to(dev))
    val.backward()
    torch.cuda.synchronize()
    print(mode, 'loss', val.item(), 'launches', len(log), 'forward', nfwd)
if os.environ.get('DIVERGE_DUMP'):
    i = int(os.environ['DIVERGE_DUMP'])
    torch.save({m: (rec[m][i][0], rec[m][i][1].cpu(), rec[m][i][2].cpu()) for m in modes}, 'gpurun_out/diverge_dump.pt')
a, b = rec[modes[0]], rec[modes[1]]
for i, ((na, oa, ia), (nb, ob, ib)) in enumerate(zip(a, b)):
    d = float((oa.double() - ob.double()).norm() / max(float(oa.double().norm()), 1e-30))
    di = float((ia.double() - ib.double()).norm() / max(float(ia.double().norm()), 1e-30))
    if i >= nfwd - 4 or d > 1e-5:
        print(f'{i:3d} {na:60s} out diff {d:9.2e}   input diff {di:9.2e}   |out| {float(oa.norm()):9.3e}')
