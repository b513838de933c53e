"""Throughput of the on-device training patch pipeline (mmlf_amd/patches.py): 16 cached 512x512 scenes,
batches of 512 augmented 96x96 patches, host parameter draws included."""
import random
import sys
import time

import torch

sys.path.insert(0, '.')
from mmlf_amd import patches, synth  # noqa: E402

B, ps = 512, 96
scenes = [synth.synth_scene(s, 512, 512, planes=1) for s in range(16)]
pipe = patches.PatchPipeline(scenes, ps, 4, True)
random.seed(0)
idx = [random.randrange(4096) for _ in range(B)]
for _ in range(2):
    out = pipe.sample(idx)
torch.cuda.synchronize()
t0 = time.time()
n = 10
for _ in range(n):
    out = pipe.sample(idx)
torch.cuda.synchronize()
dt = (time.time() - t0) / n
t1 = time.time()
for _ in range(n):
    for _ in range(B):
        patches.draw_sample((512, 512), ps, 4, True)
host = (time.time() - t1) / n
bytes_out = sum(t.numel() * t.element_size() for t in out[:8])
print(f'patch pipeline: {B / dt:.0f} patches/s ({dt * 1e3:.2f} ms per batch of {B}; host draws {host * 1e3:.2f} ms of it), '
      f'{bytes_out / 1e9:.2f} GB written per batch = {bytes_out / dt / 1e12:.2f} TB/s')
