"""Soak run of the training step at BASELINE.json's size: N steps of BASE bs=512 on a learnable synthetic target (a smooth
function of the centre view), reporting every 50 steps the loss, patches/s over the window, and the allocator's peak and
current bytes -- steady rate, no growth, falling loss, no NaN.    python tools/soak.py [steps]"""
import os
import sys
import time
import torch
sys.path.insert(0, os.getcwd())
import bench
from mmlf_amd.feed_forward import FeedForward
from mmlf_amd.train import TrainStep
dev = torch.device('cuda:0')
N = int(sys.argv[1]) if len(sys.argv) > 1 else 300
B = 512
torch.manual_seed(0)
model = FeedForward(**bench.BASE_KW).to(dev)
step = TrainStep(model, lr=1e-3, loss_margin=11)
gen = torch.Generator(device=dev).manual_seed(0)
stacks = [torch.rand((B, 9, 3, 96, 96), device=dev, generator=gen) for _ in range(4)]
gt = (stacks[0][:, 4].mean(1) * 4 - 2).contiguous()
mask = torch.ones((B, 96, 96), dtype=torch.int32, device=dev)
first = None
t0 = time.time()
for i in range(N):
    loss = step(*stacks, gt, mask, i + 1)
    if (i + 1) % 50 == 0:
        lv = float(loss)                      # (synchronises)
        dt = time.time() - t0
        first = lv if first is None else first
        print(f'step {i + 1:4d}  loss {lv:.5f}  {50 * B / dt:7.1f} patches/s  allocated {torch.cuda.memory_allocated(dev) / 2**30:6.1f} GiB  '
              f'peak {torch.cuda.max_memory_allocated(dev) / 2**30:6.1f} GiB  reserved {torch.cuda.memory_reserved(dev) / 2**30:6.1f} GiB', flush=True)
        assert lv == lv, 'NaN'
        t0 = time.time()
assert lv < 0.7 * first, (first, lv)
print('soak ok')
