"""Shape fuzz of the native path against the module's own stock-torch path (the ops the reference runs) on CPU: tiny net,
odd batch sizes and frame shapes (H != W, pitches below / at / above the kernels' 32-position groups and 127-position
switch), train-mode forward + every parameter gradient, BASE / UPR / DPP.   python tools/shape_fuzz.py [cases]"""
import os
import sys
import numpy as np
import torch
sys.path.insert(0, os.getcwd())
from mmlf_amd import loss, synth
from mmlf_amd.feed_forward import FeedForward

KW = dict(model_ksize=2, model_in_blocks=2, model_out_blocks=3, model_chs=8, model_views=9, model_cross=False,
          model_uncert=False, model_unet=False, model_discrete=False, model_no_batchnorm=False,
          model_batchnorm_momentum=0.1, val_disp_min=-3.5, val_disp_max=3.5)
rs = np.random.RandomState(int(os.environ.get('SEED', '0')))
ncase = int(sys.argv[1]) if len(sys.argv) > 1 else 24
worst = 0.0
for case in range(ncase):
    variant = ['base', 'upr', 'dpp'][case % 3]
    B = int(rs.choice([1, 2, 3, 5]))
    H = int(rs.choice([9, 14, 17, 29, 30, 31, 32, 33, 47, 64, 96, 126, 127, 128, 130]))
    W = int(rs.choice([9, 14, 17, 29, 30, 31, 32, 33, 47, 64, 96, 125, 126, 127, 128, 130]))
    if B * H * W > 40000:
        B = 1
    # FUZZ_VIEWS=1 (round 6): the number of views per stack varies too -- 27 input channels is the README default, the reference
    # takes any odd count (3 / 5 / 7 / 11 views = 9 / 15 / 21 / 33 input channels = 2 ... 5 chunks; the DPP head is 4 x views x 3 wide)
    views = int(rs.choice([3, 5, 7, 9, 11])) if os.environ.get('FUZZ_VIEWS') else 9
    kw = dict(KW, model_views=views, model_uncert=variant == 'upr', model_discrete=variant == 'dpp')
    state = synth.synth_state(synth.param_spec(**kw), 100 + case)
    g = torch.Generator().manual_seed(case)
    stacks = [torch.rand((B, views, 3, H, W), generator=g) for _ in range(4)]
    gt = 4.0 * torch.rand((B, H, W), generator=g) - 2.0
    mask = (torch.rand((B, H, W), generator=g) > 0.2).int()
    res = {}
    for dev in ('cpu', 'cuda'):
        m = FeedForward(**kw)
        m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in state.items()})
        m = m.to(dev).train()
        out = m(*[s.to(dev) for s in stacks])
        if variant == 'upr':
            val = loss.ImprovedUncertaintyL1Loss()(out, gt.to(dev), mask.to(dev), None)
        elif variant == 'dpp':
            from mmlf_amd import dl
            val = loss.MaskedCrossEntropy()(out, dl.reg_to_class(gt, -3.5, 3.5, m.steps).to(dev), mask.to(dev))
        else:
            val = loss.MaskedL1Loss()(out, gt.to(dev), mask.to(dev))
        val.backward()
        res[dev] = (out['mean'].detach().cpu(), float(val), {n: p.grad.detach().cpu() for n, p in m.named_parameters()},
                    {n: b.detach().cpu() for n, b in m.named_buffers()})
    mae = float((res['cpu'][0] - res['cuda'][0]).abs().mean())
    gerr = max(float((res['cuda'][2][n] - gr).norm() / (gr.norm() + 1e-12)) for n, gr in res['cpu'][2].items() if float(gr.norm()) > 1e-6)
    berr = max(float((res['cuda'][3][n].double() - bb.double()).abs().max()) for n, bb in res['cpu'][3].items())
    worst = max(worst, gerr)
    flag = '' if (mae < 1e-4 and gerr < 5e-2 and berr < 1e-4 and abs(res['cpu'][1] - res['cuda'][1]) < 1e-4 * max(1, abs(res['cpu'][1]))) else '   <-- CHECK'
    print(f'{case:3d} {variant} views={views} B={B} H={H} W={W}: depth MAE {mae:.2e}  loss {res["cpu"][1]:.6f} / {res["cuda"][1]:.6f}  '
          f'worst grad rel {gerr:.2e}  buffers {berr:.1e}{flag}', flush=True)
print('worst gradient relative error', worst)
