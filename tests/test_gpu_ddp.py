"""Two ranks on ONE MI355X (gloo carries the all-reduce; the compute path is the HIP one):
exercises the bucket hooks fired from the native backward, the global masked-mean denominator
and the lock-step of the replicas.  The 8-GPU RCCL run itself is the driver's."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import TINY_KW, VARIANTS
from mmlf_amd import synth

pytestmark = pytest.mark.gpu
B, PS = 4, 16


def _free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def _make(seed, variant='upr'):
    from mmlf_amd.feed_forward import FeedForward
    kw = dict(TINY_KW, **VARIANTS[variant])
    m = FeedForward(**kw)
    m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in
                       synth.synth_state(synth.param_spec(**kw), seed).items()})
    return m.to('cuda:0')


def _data():
    stacks, gt, mask = synth.synth_inputs(B, PS, seed=6)
    mask[0, :, :9] = 0
    return [torch.from_numpy(s).cuda() for s in stacks], torch.from_numpy(gt).cuda(), torch.from_numpy(mask).cuda()


def _worker(rank, world, port, out_dir, variant='upr'):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY='0')
    from mmlf_amd.train import TrainStep
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        torch.cuda.set_device(0)
        step = TrainStep(_make(3 + rank, variant), lr=1e-2, loss_margin=3)
        assert step.distributed and step.buckets is not None and step.variant == variant
        fired = []
        orig = step.buckets.ready
        step.buckets.ready = lambda g, key: (fired.append(key), orig(g, key))[1]
        stacks, gt, mask = _data()
        lo, hi = rank * B // world, (rank + 1) * B // world
        losses, grads = [], []
        for it in (1, 2):
            losses.append(float(step(*[s[lo:hi].contiguous() for s in stacks], gt[lo:hi].contiguous(),
                                     mask[lo:hi].contiguous(), it)))
            grads.append(step.grad.cpu() / world)            # the all-reduced gradient of this step, averaged
        torch.cuda.synchronize()
        torch.save({'flat': step.flat.cpu(), 'losses': losses, 'fired': fired, 'grads': grads},
                   os.path.join(out_dir, f'r{rank}.pt'))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize('variant', ['upr', 'dpp'])
def test_two_ranks_one_gpu_native_path(tmp_path, variant):
    """variant 'dpp' (round 6): BASELINE.json configs[3] -- the discrete-posterior net under data parallelism -- on the
    native path: mmlf_loss_fwd_bwd KIND_CE with the all-reduced denominator (den_override), the 108-channel head's
    gradient through the bucket hooks (reference mmlf/train/cli.py:159,201-207,247-255)."""
    from mmlf_amd.train import TrainStep
    _mk = lambda seed: _make(seed, variant)
    atol = 1e-4 if variant == 'dpp' else 3e-5       # (dpp: see tests/test_ddp_gloo.py -- unshadowed noise-driven +-lr steps)
    mp.spawn(_worker, args=(2, _free_port(), str(tmp_path), variant), nprocs=2, join=True)
    r0, r1 = torch.load(tmp_path / 'r0.pt'), torch.load(tmp_path / 'r1.pt')
    assert torch.equal(r0['flat'], r1['flat'])
    # hooks fire in backward order: head block first, shared stream nets last
    assert r0['fired'][:3] == ['out_net.2', 'out_net.1', 'out_net.0']
    assert r0['fired'][3:5] == ['in_net_id', 'in_net_hv']
    # single-process emulation with the same kernels: two replicas, averaged gradients
    stacks, gt, mask = _data()
    steps = [TrainStep(_mk(3), lr=1e-2, loss_margin=3) for _ in range(2)]
    margin = steps[0]._mask(mask)
    total = float(margin.sum())
    solid = None
    for it in (1, 2):
        for r, st in enumerate(steps):
            st.grad.zero_()
            den = torch.tensor([total / 2], dtype=torch.float64, device='cuda:0')
            loss = st._native_fwd_bwd(*[s[2 * r:2 * r + 2].contiguous() for s in stacks], gt[2 * r:2 * r + 2].contiguous(),
                                      margin[2 * r:2 * r + 2].contiguous(), den)
            if r == 0:
                np.testing.assert_allclose(float(loss), r0['losses'][it - 1], rtol=1e-5)
        avg = (steps[0].grad + steps[1].grad) / 2
        ok = avg.abs() > 1e-5
        solid = ok if solid is None else solid & ok
        for st in steps:
            st.grad.copy_(avg)
            st.adam_steps += 1
            st._adam(st.current_lr(it), 1.0)
    torch.testing.assert_close(r0['flat'][solid.cpu()], steps[0].flat.cpu()[solid.cpu()], rtol=1e-4, atol=atol)
    # ... and an INDEPENDENT reference: the same two shards through the stock-torch module path on the CPU
    # (nn.Conv2d / BatchNorm2d / autograd), gradients averaged by hand, the same Adam arithmetic
    # Conditioning of that comparison: Adam's update is lr * m / sqrt(v), so a gradient difference d on an element of magnitude
    # |g| moves its weight by ~ lr * d / |g|.  UPR: this library's gradients differ from the CPU's by ~1e-8 absolute and every
    # element above 1e-5 is held to 3e-5.  DPP: the cross entropy's gradients are sums of softmax - target terms that cancel
    # (they add up to zero over the classes of a pixel), so the float32 noise of the two implementations is ~1e-6 absolute next
    # to many gradients of 1e-5: there the comparison is made where it is conditioned -- elements above 1e-3 in both steps, 10 %
    # of an lr step -- and the per-step ALL-REDUCED GRADIENT of the ranks is held against the CPU reference directly.
    cpu_floor, cpu_atol = (1e-3, 1e-3) if variant == 'dpp' else (1e-5, atol)
    cpu_solid, rel = None, []
    cpu_steps = [TrainStep(_mk(3).cpu(), lr=1e-2, loss_margin=3) for _ in range(2)]
    cstacks, cgt, cmargin = [s.cpu() for s in stacks], gt.cpu(), margin.cpu()
    for it in (1, 2):
        for r, st in enumerate(cpu_steps):
            st.grad.zero_()
            st.model.train()
            den = torch.tensor([total / 2], dtype=torch.float64)
            loss = st._torch_fwd_bwd(*[s[2 * r:2 * r + 2].contiguous() for s in cstacks], cgt[2 * r:2 * r + 2].contiguous(),
                                     cmargin[2 * r:2 * r + 2].contiguous(), den)
            if r == 0:
                np.testing.assert_allclose(float(loss), r0['losses'][it - 1], rtol=2e-5)
        avg = (cpu_steps[0].grad + cpu_steps[1].grad) / 2
        ok = avg.abs() > cpu_floor
        cpu_solid = ok if cpu_solid is None else cpu_solid & ok
        rel.append(float((r0['grads'][it - 1] - avg).norm() / avg.norm()))
        for st in cpu_steps:
            st.grad.copy_(avg)
            st.adam_steps += 1
            st._adam(st.current_lr(it), 1.0)
    # the ranks' all-reduced gradients against the CPU reference's: step 1 on identical weights (float32 level), step 2 on weights
    # that went through one Adam update (elements whose step-1 gradient is rounding noise moved by +-lr in either implementation)
    assert rel[0] <= 1e-3 and rel[1] <= 1e-1, rel
    cpu_solid &= solid.cpu()
    assert float(cpu_solid.float().mean()) > (0.02 if variant == 'dpp' else 0.5)
    if variant == 'dpp':
        # step 2 runs on weights whose noise-driven elements moved by +-lr in either implementation (its gradients differ by
        # 3 % in L2, held above): the weights are held to "all but 1 % of the conditioned elements within a tenth of an lr step,
        # none further than one lr step" (measured: 0.4 % beyond 1e-3, worst 4e-3); a sign error would show as 2e-2 everywhere
        d = (r0['flat'][cpu_solid] - cpu_steps[0].flat[cpu_solid]).abs()
        assert float((d > cpu_atol).float().mean()) <= 1e-2 and float(d.max()) <= 1e-2, (float((d > cpu_atol).float().mean()), float(d.max()))
    else:
        torch.testing.assert_close(r0['flat'][cpu_solid], cpu_steps[0].flat[cpu_solid], rtol=1e-4, atol=cpu_atol)


def _rccl_single_rank_worker(_rank, port, out_dir):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY='0')
    from mmlf_amd.train import TrainStep
    torch.cuda.set_device(0)
    dist.init_process_group('nccl', rank=0, world_size=1, device_id=torch.device('cuda:0'))
    try:
        res = {}
        stacks, gt, mask = _data()
        for tag, force in (('plain', False), ('rccl', True)):
            step = TrainStep(_make(3, 'dpp'), lr=1e-2, loss_margin=3, force_distributed=force)
            assert step.distributed == force and (step.buckets is not None) == force
            losses = [float(step(*stacks, gt, mask, it)) for it in (1, 2, 3)]
            torch.cuda.synchronize()
            res[tag] = {'flat': step.flat.cpu(), 'grad': step.grad.cpu(), 'losses': losses}
        torch.save(res, os.path.join(out_dir, 'rccl.pt'))
    finally:
        dist.destroy_process_group()


def test_rccl_carries_the_gradient_in_a_group_of_one(tmp_path):
    """RCCL itself under the data-parallel path (round 6): no multi-GPU box has ever been available to this build, so the
    collective library's streams and the kernels this package launches through ctypes had never met.  A process group of ONE
    rank on the `nccl` backend (= RCCL) with `force_distributed=True`: the weight / buffer broadcasts, the mask-count
    all-reduce issued before forward and waited for before the loss kernel, the bucket all-reduces fired from the native
    backward (async, on RCCL's stream) and waited for before Adam.  A one-rank sum changes no value, so three steps must
    give the bits of the non-distributed step -- which they only do if every stream dependency is in place (a gradient read
    before its kernels finished, or Adam running before a bucket's wait, would show).  Not a bandwidth measurement."""
    mp.spawn(_rccl_single_rank_worker, args=(_free_port(), str(tmp_path)), nprocs=1, join=True)
    res = torch.load(tmp_path / 'rccl.pt')
    assert res['plain']['losses'] == res['rccl']['losses']
    assert torch.equal(res['plain']['grad'], res['rccl']['grad'])
    assert torch.equal(res['plain']['flat'], res['rccl']['flat'])
    assert torch.isfinite(res['rccl']['flat']).all()


def test_bench_two_ranks_gloo_rehearsal(tmp_path):
    """bench.py's N>1 path end to end on the one GPU of this box: torch.distributed.run launcher, process-group
    init, sharded batch, bucketed all-reduce hooks fired from the native backward, max-over-ranks timing and rank 0's
    JSON line -- with gloo standing in for RCCL (`--backend gloo`; the 8-GPU RCCL run is the driver's)."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0', MASTER_ADDR='127.0.0.1')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
           '--master-port', str(_free_port()), os.path.join(root, 'bench.py'), '--gpus', '2', '--global-batch', '8',
           '--patch', '32', '--steps', '1', '--warmup', '1', '--backend', 'gloo', '--no-cpu-baseline', '--no-f32-leg', '--ese-size', '48']
    res = subprocess.run(cmd, cwd=root, env=env, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stderr[-2000:]
    lines = [l for l in res.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1, res.stdout[-2000:]            # rank 0 alone prints
    line = json.loads(lines[0])
    assert line['n_gpus'] == 2 and line['config']['per_gpu_batch'] == 4 and line['config']['parallelism'] == 'dp2'
    assert line['scaling'] == 'strong' and line['value'] > 0 and np.isfinite(line['config']['loss'])
    assert 'gloo' in line['config']['backend']
    # round 6: the workload names itself (not a BASELINE shape here), the line carries the host's enqueue time and one wait per
    # bucket, and the DPP / UPR variants run through the same two-rank path (BASELINE.json configs[3] is DPP under data parallelism)
    assert 'not a BASELINE.json shape' in line['config']['workload'] and '2 GPUs, data parallel' in line['config']['workload']
    assert line['host_enqueue_ms'] > 0 and len(line['allreduce_ms_by_bucket']) == line['config']['buckets']
    for v in ('dpp', 'upr'):
        leg = line[v]
        assert leg['value'] > 0 and np.isfinite(leg['loss']) and leg['per_gpu_batch'] == 4 and leg['buckets'] == line['config']['buckets']
        assert len(leg['allreduce_ms_by_bucket']) == leg['buckets'] and v.upper() in leg['workload']
    assert line['dpp']['gradient_bytes'] > line['config']['gradient_bytes']          # the 108-channel head
    # BASELINE.json configs[4] under --gpus N: one light field per rank, replicas only, slowest rank reported
    rep = line['ese_replicas']
    assert rep['scenes'] == 2 and rep['members'] == 70 and rep['finite'] and rep['value'] > 0
    assert abs(rep['scenes_per_s'] - 2 / rep['value']) < 1e-2 * rep['scenes_per_s']


def test_bench_rccl_self_rehearsal_prints_one_line():
    """bench.py --rccl-self (round 6): every torch.distributed call of the script's N > 1 path -- process group on the nccl
    backend (= RCCL), the distributed TrainStep, per-bucket waits, the dpp / upr legs, the ESE replica reductions -- in a group
    of ONE rank on this box's GPU.  RCCL prints a version banner to stdout when its communicator is made: the script must
    still put exactly ONE line there, the JSON line (the driver parses it)."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0', MASTER_ADDR='127.0.0.1', MASTER_PORT=str(_free_port()))
    cmd = [sys.executable, os.path.join(root, 'bench.py'), '--gpus', '1', '--rccl-self', '--global-batch', '4', '--patch', '32',
           '--steps', '2', '--warmup', '1', '--no-cpu-baseline', '--ese-size', '48']
    res = subprocess.run(cmd, cwd=root, env=env, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stderr[-2000:]
    out = [l for l in res.stdout.splitlines() if l.strip()]
    assert len(out) == 1 and out[0].startswith('{'), res.stdout[-1500:]
    line = json.loads(out[0])
    assert 'rccl-self' in line['config']['backend'] and line['config']['buckets'] == 3
    assert len(line['allreduce_ms_by_bucket']) == 3 and line['dpp']['value'] > 0 and line['upr']['value'] > 0
    assert line['ese_replicas']['scenes'] == 1 and line['ese_replicas']['finite']
