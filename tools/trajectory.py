"""Do the three arithmetic modes TRAIN alike?  The same N optimisation steps (reference loop body mmlf/train/cli.py:185-258:
zero_grad, forward, L1 loss, backward, Adam) -- same initial weights, same data, same learnable target as tools/soak.py --
under MMLF_CONV_MODE = f32 (exact-f32 MFMA), bf16x6 (exact 3 x bf16 split) and f16x3 (the default: 22 significant bits per
operand), one after the other in one process.  Prints the loss every `every` steps per mode, and at the end, per mode, the
relative L2 distance of every parameter tensor from the f32 run's (median / 90th percentile / worst tensor) next to how far
the f32 run itself moved from the initial weights.  A chaotic system amplifies rounding differences, so the curves are not
expected to be equal -- they are expected to be as close to each other as two float32 implementations are: the yardstick
printed beside them is the f32 mode run twice with the batch order of patches rotated by one (same set of patches, same
per-batch statistics up to summation order), i.e. a pure rounding-order perturbation of the exact-f32 path.

    python tools/trajectory.py [steps=200] [B=64] [every=10] [full|tiny]
"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.getcwd())
import bench  # noqa: E402
from mmlf_amd import engine  # noqa: E402
from mmlf_amd.feed_forward import FeedForward  # noqa: E402
from mmlf_amd.train import TrainStep  # noqa: E402

TINY = dict(bench.BASE_KW, model_in_blocks=2, model_out_blocks=3, model_chs=8)


def run(mode, steps, B, every, kw, ps, lr=1e-3, roll=0, dev=None):
    """-> (losses at every `every`-th step, {name: final parameter (cpu)}, {name: initial parameter})"""
    dev = dev or torch.device('cuda:0')
    old, engine.CONV_MODE = engine.CONV_MODE, mode
    try:
        torch.manual_seed(0)
        model = FeedForward(**kw).to(dev)
        init = {k: v.detach().cpu().clone() for k, v in model.named_parameters()}
        step = TrainStep(model, lr=lr, loss_margin=min(11, ps // 4))
        gen = torch.Generator(device=dev).manual_seed(0)
        stacks = [torch.rand((B, 9, 3, ps, ps), device=dev, generator=gen) for _ in range(4)]
        gt = (stacks[0][:, 4].mean(1) * 4 - 2).contiguous()            # learnable: a smooth function of the centre view
        if roll:
            stacks = [s.roll(roll, 0).contiguous() for s in stacks]
            gt = gt.roll(roll, 0).contiguous()
        mask = torch.ones((B, ps, ps), dtype=torch.int32, device=dev)
        losses = []
        for i in range(steps):
            loss = step(*stacks, gt, mask, i + 1)
            if (i + 1) % every == 0 or i == 0:
                losses.append(float(loss))
        final = {k: v.detach().cpu().clone() for k, v in model.named_parameters()}
    finally:
        engine.CONV_MODE = old
    return losses, final, init


def distances(a, b):
    """relative L2 distance per tensor (tensors that are all zero in `b` -- BatchNorm biases at initialisation -- are left out)"""
    return {k: float((a[k] - b[k]).norm() / b[k].norm()) for k in a if float(b[k].norm()) > 0}


def summary(d):
    v = np.array(sorted(d.values()))
    worst = max(d, key=d.get)
    return f'median {np.median(v):.3e}  p90 {v[int(0.9 * (len(v) - 1))]:.3e}  worst {v[-1]:.3e} ({worst})'


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 64
    every = int(sys.argv[3]) if len(sys.argv) > 3 else 10
    tiny = len(sys.argv) > 4 and sys.argv[4] == 'tiny'
    kw, ps = (TINY, 32) if tiny else (bench.BASE_KW, 96)
    print(f'# tools/trajectory.py: {steps} BASE steps, bs={B}, ps={ps}, {"tiny (chs=8)" if tiny else "full-size"} net, lr 1e-3, '
          f'same seed / data / target in every mode', flush=True)
    res = {}
    for mode, roll in (('f32', 0), ('f32', 1), ('bf16x6', 0), ('f16x3', 0)):
        tag = mode + ('/rolled' if roll else '')
        res[tag] = run(mode, steps, B, every, kw, ps, roll=roll)
        print(f'{tag:>12} loss', ' '.join(f'{v:.5f}' for v in res[tag][0]), flush=True)
    ref_l, ref_w, init = res['f32']
    print(f'{"f32 vs init":>22}: {summary(distances(ref_w, init))}   (how far training moved the weights)')
    for tag in ('f32/rolled', 'bf16x6', 'f16x3'):
        l, w, _ = res[tag]
        dl = max(abs(a - b) / max(abs(b), 1e-30) for a, b in zip(l, ref_l))
        print(f'{tag + " vs f32":>22}: {summary(distances(w, ref_w))}   max relative loss difference {dl:.3e}   final loss {l[-1]:.5f} vs {ref_l[-1]:.5f}')
    # the two split modes run the SAME kernels in the same summation order and differ in the operand split alone (exact 3 x bf16
    # against 2 x f16 = 22 bits): their distance isolates what the narrower operands do to a trajectory
    l, w, _ = res['f16x3']
    lb, wb, _ = res['bf16x6']
    dl = max(abs(a - b) / max(abs(b), 1e-30) for a, b in zip(l, lb))
    print(f'{"f16x3 vs bf16x6":>22}: {summary(distances(w, wb))}   max relative loss difference {dl:.3e}')
    # per tensor: relative L2 distance of the final weights from the f32 run's
    cols = {tag: distances(res[tag][1], ref_w) for tag in ('f32/rolled', 'bf16x6', 'f16x3')}
    cols['f16x3 vs bf16x6'] = distances(w, wb)
    print(f'{"tensor":<34}' + ''.join(f'{t:>18}' for t in cols))
    for k in cols['f16x3']:
        print(f'{k:<34}' + ''.join(f'{cols[t].get(k, float("nan")):18.3e}' for t in cols))
    print('trajectory ok' if all(np.isfinite(res[t][0]).all() for t in res) else 'NaN')


if __name__ == '__main__':
    main()
