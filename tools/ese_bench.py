import sys, time, torch, numpy as np
sys.path.insert(0, '.')
from mmlf_amd import synth
from mmlf_amd.feed_forward import FeedForward
from mmlf_amd.ensamble import Ensamble
dev = torch.device('cuda:0')
kw = dict(model_ksize=2, model_in_blocks=3, model_out_blocks=8, model_chs=70, model_views=9, model_cross=False, model_uncert=True, model_unet=False, model_discrete=False, model_no_batchnorm=False, model_batchnorm_momentum=0.1, val_disp_min=-3.5, val_disp_max=3.5)
torch.manual_seed(0)
m = FeedForward(**kw).to(dev).eval()
ens = Ensamble(m, -3.5, 3.5, 0.1).eval()
S = int(sys.argv[1]) if len(sys.argv) > 1 else 512
stacks = [torch.rand(1, 9, 3, S, S, device=dev) for _ in range(4)]
with torch.no_grad():
    for it in range(2):
        torch.cuda.synchronize(); t0 = time.time()
        out = ens(*stacks)
        torch.cuda.synchronize(); dt = time.time() - t0
        print(f'ESE {S}x{S}: {dt:.3f} s/scene, {70*2529.3*(S/512)**2/dt/1e3:.1f} TFLOP/s algorithmic, peak mem {torch.cuda.max_memory_allocated()/2**30:.1f} GiB', flush=True)
print({k: tuple(v.shape) for k, v in out.items()})
