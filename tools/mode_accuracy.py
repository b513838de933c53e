"""End-to-end accuracy of the three conv arithmetic modes against the reference's own float32 outputs
(tests/golden/g2_full_base.npz: full-size net, 96x96): eval depth MAE, train-mode depth MAE, loss and the
relative L2 error of sampled parameter gradients."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, '.')
sys.path.insert(0, 'tests')
from mmlf_amd import engine, synth  # noqa: E402
from mmlf_amd.feed_forward import FeedForward  # noqa: E402
from mmlf_amd.loss import MaskedL1Loss, create_mask_margin  # noqa: E402
import bench  # noqa: E402

dev = torch.device('cuda:0')
g = np.load('tests/golden/g2_full_base.npz')
kw = bench.BASE_KW
for mode in ('f32', 'bf16x6', 'f16x3'):
    engine.CONV_MODE = mode
    m = FeedForward(**kw).to(dev)
    st = synth.synth_state(synth.param_spec(**kw), seed=21)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in st.items()})
    stacks, _, _ = synth.synth_inputs(1, 96, seed=7)
    m.eval()
    with torch.no_grad():
        out = m(*[torch.from_numpy(s).to(dev) for s in stacks])
    mae_eval = np.abs(out['mean'].cpu().numpy() - g['eval_mean']).mean()
    stacks, gt, mask = synth.synth_inputs(2, 96, seed=8)
    mask = torch.from_numpy(mask).int() * create_mask_margin(mask.shape, 11)
    m.train()
    out = m(*[torch.from_numpy(s).to(dev) for s in stacks])
    mae_train = np.abs(out['mean'].detach().cpu().numpy() - g['train_mean']).mean()
    loss = MaskedL1Loss()(out, torch.from_numpy(gt).to(dev), mask.to(dev))
    loss.backward()
    errs = []
    for n, p in m.named_parameters():
        key = f'grad_s/{n}'
        if key in g.files:
            ref = g[key]
            got = p.grad.detach().cpu().numpy().reshape(-1)[::97][:ref.size] if ref.size != p.numel() else p.grad.detach().cpu().numpy().reshape(ref.shape)
            if got.shape == ref.shape:
                errs.append(np.linalg.norm(got - ref) / max(np.linalg.norm(ref), 1e-30))
    print(f'{mode:7s}: eval depth MAE {mae_eval:.3e}  train depth MAE {mae_train:.3e}  loss rel err {abs(loss.item() - float(g["loss"])) / float(g["loss"]):.2e}'
          f'  grad rel-L2 median {np.median(errs):.3e} max {np.max(errs):.3e} ({len(errs)} tensors)')
