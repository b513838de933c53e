"""Distance of each implementation's gradients from the reference's float64 run of the well-conditioned train step
(tests/golden/g11_conditioned_*.npz), in units of the reference's own float32 distance, per parameter tensor.
  python tools/grad_yardstick.py [base|upr]"""
import os
import sys
import numpy as np
import torch
sys.path.insert(0, os.getcwd())
sys.path.insert(0, os.path.join(os.getcwd(), 'tests'))
from conftest import BASE_KW, VARIANTS, load_golden
from mmlf_amd import engine, synth, loss as L
from mmlf_amd.feed_forward import FeedForward

variant = sys.argv[1] if len(sys.argv) > 1 else 'base'
g = load_golden(f'g11_conditioned_{variant}.npz')
g32 = {k[4:]: v for k, v in g.items() if k.startswith('f32/')}
g64 = {k[4:]: v for k, v in g.items() if k.startswith('f64/')}
kw = dict(BASE_KW, **VARIANTS[variant])
dev = torch.device('cuda:0')
stacks, gt, mask = synth.synth_inputs(2, 96, seed=int(g['seed']))
mask = torch.from_numpy(mask).int() * L.create_mask_margin(mask.shape, 11)
rows = {}
for mode in ('f16x3', 'f32', 'bf16x6', 'torch'):
    m = FeedForward(**kw)
    m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in synth.synth_state(synth.param_spec(**kw), seed=21).items()})
    m.to(dev).train()
    if mode == 'torch':
        m._native_ok = False
    else:
        engine.CONV_MODE = mode
    out = m(*[torch.from_numpy(s).to(dev) for s in stacks])
    crit = L.ImprovedUncertaintyL1Loss() if variant == 'upr' else L.MaskedL1Loss()
    val = crit(out, torch.from_numpy(gt).to(dev), mask.to(dev), None) if variant == 'upr' else crit(out, torch.from_numpy(gt).to(dev), mask.to(dev))
    val.backward()
    e = np.abs(out['mean'].detach().cpu().numpy().astype(np.float64) - g64['train_mean']).mean()
    eref = np.abs(g32['train_mean'].astype(np.float64) - g64['train_mean']).mean()
    print(f'{mode}: loss {val.item():.8f} (f64 {float(g64["loss"]):.8f}, ref32 {float(g32["loss"]):.8f}); depth MAE vs f64 {e:.3e} (ref32 {eref:.3e})')
    for n, p in m.named_parameters():
        got = p.grad.cpu().numpy().astype(np.float64)
        got = got.reshape(-1)[::97] if got.size > 4096 else got
        r64 = g64[f'grad_s/{n}']
        d = np.linalg.norm(got - r64)
        dref = np.linalg.norm(g32[f'grad_s/{n}'].astype(np.float64) - r64)
        rows.setdefault(n, {})[mode] = (d / max(dref, 1e-30), d / max(np.linalg.norm(r64), 1e-30), r64.size)
print(f'{"tensor":34s} n    ' + ''.join(f'{m:>18s}' for m in ('f16x3', 'f32', 'bf16x6', 'torch')) + '   (ratio to ref32 distance | relative to |g64|)')
for n, r in rows.items():
    print(f'{n:34s} {r["f16x3"][2]:5d}' + ''.join(f'  {r[m][0]:7.2f} {r[m][1]:8.1e}' for m in ('f16x3', 'f32', 'bf16x6', 'torch')))
for m in ('f16x3', 'f32', 'bf16x6', 'torch'):
    v = np.array([r[m][0] for n, r in rows.items() if not (n.endswith('.2.bias') and '.7.' not in n)])
    print(m, 'median ratio', np.median(v), 'max', v.max(), 'p90', np.percentile(v, 90))
