#!/usr/bin/env python3
"""bench.py -- 96x96 EPI patches/s, BASE fwd+bwd+Adam, global bs=512, on N MI355X of one node.

  python bench.py --gpus 1 --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
         --master-port P bench.py --gpus N --steps K --warmup W

A "step" is one pass of the hot path (reference mmlf/train/cli.py:227-258: zero_grad, forward,
loss, backward, Adam) over one batch of synthetic patches already resident in HBM.  The global
batch is fixed at 512 (BASELINE.json), sharded 512/N per rank (strong scaling); one RCCL
all-reduce of the flat gradient per step, bucketed and overlapped with backward.
Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

BASE_KW = dict(model_ksize=2, model_in_blocks=3, model_out_blocks=8, model_chs=70, model_views=9,
               model_cross=False, model_uncert=False, model_unet=False, model_discrete=False,
               model_no_batchnorm=False, model_batchnorm_momentum=0.1, val_disp_min=-3.5, val_disp_max=3.5)
# SURVEY.md section 8(d) / BASELINE.md section 3: conv MACs only, 2 FLOP/MAC, unpadded channels
GFLOP_PER_PATCH = {'base': 268.373, 'upr': 268.437, 'dpp': 277.718}
PEAK_F32_MFMA_TFLOPS = 157.3      # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, exact f32
PEAK_BF16_MFMA_TFLOPS = 2500.0    # dense bf16 / f16 MFMA; the split kernels spend 3 (f16x3) or 6 (bf16x6) passes per f32 product
# the committed rocprofv3 PMC passes `roofline.traffic` is read from (tools/profile_round.sh writes it; ONE file, named in
# the line as `traffic_source`: an older round's numbers are never substituted silently)
PMC_SUMMARY = os.environ.get('MMLF_PMC_SUMMARY', os.path.join('profiles', 'r06_pmc_bs512_base_summary.json'))
KW_EXTRA = {'base': {}, 'upr': {'model_uncert': True}, 'dpp': {'model_discrete': True}}


def pmc_traffic(kernel, min_ns=3e6, max_ns=1e12):
    """(HBM bytes per launch of `kernel`, source file) -- `kernel` is a name prefix: the epilogue variants of one kernel
    template are averaged, weighted by their launch counts -- from the committed rocprofv3 PMC passes (separate FETCH_SIZE /
    WRITE_SIZE runs of this same command, tools/profile_round.sh; gfx950 correction 2*FETCH_SIZE + WRITE_SIZE, KB ->
    bytes).  Counters cannot be read from inside a timed run: this is the committed measurement of the same kernels; if
    the file is missing or does not hold the kernel, traffic is null and stderr says so."""
    path = PMC_SUMMARY if os.path.isabs(PMC_SUMMARY) else os.path.join(ROOT, PMC_SUMMARY)
    try:
        with open(path) as f:
            d = json.load(f)
        hit = [v for k, v in d.items() if kernel in k and min_ns < v.get('avg_ns', 0) < max_ns]     # (default: the 280-wide launches)
        if hit:
            n = sum(v['launches'] for v in hit)
            return round(sum(v['hbm_bytes_per_launch'] * v['launches'] for v in hit) / n), PMC_SUMMARY
        why = f'no launches of {kernel} in {PMC_SUMMARY}'
    except (OSError, ValueError, KeyError) as e:
        why = f'{PMC_SUMMARY}: {e}'
    print(f'bench.py: roofline.traffic unavailable ({why}); run tools/profile_round.sh', file=sys.stderr, flush=True)
    return None, f'unavailable: {why}'


def kernel_sources_of(build):
    """the part of an mmlf_build_info() string that identifies the kernels: the content hash of their sources (src=) and
    every build switch -- not git=, which names the commit the library happened to be built at (a documentation commit
    changes it, a kernel edit in a dirty tree does not)"""
    return ' '.join(sorted(f for f in build.split() if not f.startswith('git=')))


def pmc_build(loaded_build):
    """(build string the committed PMC passes were collected on, or None for summaries older than round 6; whether it is the
    library this run times).  The counters are a committed measurement, not a live one: when the kernels have changed since, the
    over-fetch they report may no longer be the timed library's -- the line says which build they belong to and stderr warns."""
    path = PMC_SUMMARY if os.path.isabs(PMC_SUMMARY) else os.path.join(ROOT, PMC_SUMMARY)
    try:
        with open(path) as f:
            meta = json.load(f).get('_meta') or {}
    except (OSError, ValueError):
        meta = {}
    b = meta.get('build')
    same = b is not None and kernel_sources_of(b) == kernel_sources_of(loaded_build)
    if not same:
        print(f'bench.py: WARNING roofline.traffic comes from {PMC_SUMMARY}, collected on build [{b}]; the library timed here is '
              f'[{loaded_build}] -- re-run tools/profile_round.sh if the kernels changed', file=sys.stderr, flush=True)
    return b, same


def host_threads():
    """threads this process may actually use (the GPU box gives a 1-GPU job a share of the host's cores)"""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except AttributeError:
        pass
    try:                                     # cgroup v2 CPU quota, if any
        quota, period = open('/sys/fs/cgroup/cpu.max').read().split()
        if quota != 'max':
            n = max(1, min(n, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return n


def cpu_baseline(variant, patch):
    """The reference's CPU arithmetic on the host cores of this box: mmlf_amd.FeedForward on CPU tensors runs the
    module tree with the stock torch (mkldnn) ops the reference executes -- nn.Conv2d / BatchNorm2d / ReLU,
    autograd backward -- and TrainStep's CPU branch does zero_grad / loss / backward / Adam like
    mmlf/train/cli.py:243-258.  (The reference itself cannot travel to the GPU box.)  BASELINE.md section 4:
    B=8, 1 warm-up + 3 timed iterations, median."""
    import statistics
    from mmlf_amd.feed_forward import FeedForward
    from mmlf_amd.train import TrainStep
    kw = dict(BASE_KW, **KW_EXTRA[variant])
    cores = host_threads()
    torch.set_num_threads(cores)
    torch.manual_seed(0)
    model = FeedForward(**kw)
    step = TrainStep(model, lr=1e-3, loss_margin=11)
    B = 8
    g = torch.Generator().manual_seed(0)
    stacks = [torch.rand((B, 9, 3, patch, patch), generator=g) for _ in range(4)]
    gt = 4.0 * torch.rand((B, patch, patch), generator=g) - 2.0
    mask = torch.ones((B, patch, patch), dtype=torch.int32)
    times = []
    for it in range(4):
        t0 = time.time()
        step(*stacks, gt, mask, it + 1)
        times.append(time.time() - t0)
    med = statistics.median(times[1:])
    return {'value': round(B / med, 4), 'unit': 'patches/s', 'cores': cores, 'kind': 'port',
            'sample': f'{variant.upper()} fwd+loss+bwd+Adam on B={B} patches {patch}x{patch}, stock torch CPU ops '
                      f'(mkldnn conv / native_batch_norm / autograd, the ops the reference runs), {cores} threads, '
                      f'1 warm-up + 3 timed steps, median {med:.2f} s/step'}


def workload_name(variant, world, global_batch, patch):
    """config.workload: what was actually run, and which BASELINE.json config that is (configs[1] = BASE bs=512 on one GPU,
    [2] = UPR bs=512 on one GPU, [3] = DPP bs=512 over 8 GPUs; anything else is named as the shape it is)"""
    what = (f'{variant.upper()} fwd+bwd+Adam, global bs={global_batch} ps={patch} synthetic EPI patches, default torch init, '
            f'{world} GPU' + ('s, data parallel' if world > 1 else ''))
    std = global_batch == 512 and patch == 96
    if std and world == 1 and variant in ('base', 'upr'):
        return what + f' (BASELINE.json configs[{1 if variant == "base" else 2}])'
    if std and variant == 'dpp':
        return what + (' (BASELINE.json configs[3])' if world == 8 else f' (BASELINE.json configs[3] is this on 8 GPUs; this run: {world})')
    if std:
        return what + f' (the workload of BASELINE.json configs[{1 if variant == "base" else 2}] sharded over {world} GPUs)'
    return what + ' (not a BASELINE.json shape)'


def timed_steps(step, data, steps, first_it, sync):
    """`steps` optimisation steps between two sync()s.  Returns (seconds, last loss, host_enqueue_ms): the last is the mean
    wall time from a step's start until step() RETURNS (everything enqueued, nothing waited for) -- if that is not well under
    the step's GPU time the rank is launch-bound (at 64 patches per rank a step is ~58 ms of GPU work behind ~1 000 launches
    from Python / ctypes)."""
    stacks, gt, mask = data
    sync()
    host = 0.0
    t0 = time.time()
    for k in range(steps):
        h0 = time.time()
        loss = step(*stacks, gt, mask, first_it + k)
        host += time.time() - h0
    sync()
    return time.time() - t0, loss, 1e3 * host / max(1, steps)


class stdout_to_stderr:
    """RCCL prints a banner (version, HIP / ROCm version, hostname, library path) to STDOUT when its first communicator is made;
    this script's contract is ONE JSON line there.  File-descriptor level, so that the library's own writes are caught."""

    def __enter__(self):
        sys.stdout.flush()
        self.saved = os.dup(1)
        os.dup2(2, 1)

    def __exit__(self, *exc):
        sys.stdout.flush()
        os.dup2(self.saved, 1)
        os.close(self.saved)


FORCE_DIST = False     # --rccl-self: the data-parallel path in a process group of ONE rank (RCCL's API and streams under the step)


def make_step(variant, B, patch, dev, seed):
    from mmlf_amd.feed_forward import FeedForward
    from mmlf_amd.train import TrainStep
    kw = dict(BASE_KW, **KW_EXTRA[variant])
    torch.manual_seed(0)
    model = FeedForward(**kw).to(dev)
    step = TrainStep(model, lr=1e-3, loss_margin=11, force_distributed=FORCE_DIST)
    gen = torch.Generator(device=dev).manual_seed(seed)
    stacks = [torch.rand((B, 9, 3, patch, patch), device=dev, generator=gen) for _ in range(4)]
    gt = 4.0 * torch.rand((B, patch, patch), device=dev, generator=gen) - 2.0
    mask = torch.ones((B, patch, patch), dtype=torch.int32, device=dev)
    return step, stacks, gt, mask


def extra_leg(variant, B, patch, dev, steps, peak):
    """one more BASELINE.json config through the same train step, single GPU: value + whole-step TFLOP/s"""
    step, stacks, gt, mask = make_step(variant, B, patch, dev, seed=1)
    step(*stacks, gt, mask, 1)
    dt, loss, host_ms = timed_steps(step, (stacks, gt, mask), steps, 2, torch.cuda.synchronize)
    value = B * steps / dt
    tf = value * GFLOP_PER_PATCH[variant] * (patch / 96.0) ** 2 / 1e3
    out = {'value': round(value, 2), 'unit': 'patches/s', 'steps': steps, 'per_gpu_batch': B,
           'ms_per_step': round(1e3 * dt / steps, 2), 'whole_step_tflops': round(tf, 1),
           'frac_of_peak': round(tf / peak, 4), 'loss': round(float(loss), 6), 'host_enqueue_ms': round(host_ms, 2),
           'workload': workload_name(variant, 1, B, patch)}
    del step, stacks, gt, mask
    torch.cuda.empty_cache()
    return out


def allreduce_report(step, dev):
    """per step, max over ranks: how long the compute stream stood behind each bucket's all-reduce after backward had been
    enqueued (one entry per bucket, in firing order = backward order), and their sum"""
    if step.buckets is None or not step.buckets.wait_events:
        return None, None
    ev = step.buckets.wait_events
    per = torch.tensor([sum(s[b][0].elapsed_time(s[b][1]) for s in ev) / len(ev) for b in range(len(ev[0]))],
                       dtype=torch.float64, device=dev)
    tot = per.sum().reshape(1)
    dist.all_reduce(per, op=dist.ReduceOp.MAX)
    dist.all_reduce(tot, op=dist.ReduceOp.MAX)
    return round(float(tot), 3), [round(float(v), 3) for v in per]


def ddp_leg(variant, global_batch, patch, dev, steps, world, rank, sync):
    """another variant through the SAME N-rank path as the main leg (sharded batch, bucketed all-reduce, max-over-ranks time):
    BASELINE.json configs[3] is the DPP net under data parallelism"""
    B = global_batch // world
    step, stacks, gt, mask = make_step(variant, B, patch, dev, seed=rank)
    step(*stacks, gt, mask, 1)
    step.buckets.wait_events = []
    dt, loss, host_ms = timed_steps(step, (stacks, gt, mask), steps, 2, sync)
    t = torch.tensor([dt, host_ms], dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dt, host_ms = float(t[0]), float(t[1])
    ar_total, ar_buckets = allreduce_report(step, dev)
    out = {'value': round(global_batch * steps / dt, 2), 'unit': 'patches/s', 'steps': steps, 'per_gpu_batch': B,
           'ms_per_step': round(1e3 * dt / steps, 2), 'loss': round(float(loss), 6), 'host_enqueue_ms': round(host_ms, 2),
           'allreduce_ms': ar_total, 'allreduce_ms_by_bucket': ar_buckets, 'buckets': len(step.buckets.ranges),
           'gradient_bytes': int(step.grad.numel()) * 4, 'workload': workload_name(variant, world, global_batch, patch)}
    del step, stacks, gt, mask
    torch.cuda.empty_cache()
    return out


def autograd_leg(variant, B, patch, dev, steps, ref_ms):
    """The loop INTEGRATION.md section 1 advertises as the drop-in surface -- reference mmlf/train/cli.py:243-258 --
    instead of TrainStep: optimizer.zero_grad(); model(h, v, i, d) with the head outputs materialised (the UPR
    posterior: 2 GB per step); the mmlf_amd.loss module; loss.backward() through the one autograd node;
    torch.optim.Adam.step()."""
    from mmlf_amd import loss as loss_mod
    from mmlf_amd.feed_forward import FeedForward
    kw = dict(BASE_KW, **KW_EXTRA[variant])
    torch.manual_seed(0)
    model = FeedForward(**kw).to(dev)
    model.train()
    opt = torch.optim.Adam(model.parameters(), lr=1e-3)
    gen = torch.Generator(device=dev).manual_seed(1)
    stacks = [torch.rand((B, 9, 3, patch, patch), device=dev, generator=gen) for _ in range(4)]
    gt = 4.0 * torch.rand((B, patch, patch), device=dev, generator=gen) - 2.0
    mask = (torch.ones((B, patch, patch), dtype=torch.int32) * loss_mod.create_mask_margin((B, patch, patch), 11)).to(dev)
    crit = loss_mod.ImprovedUncertaintyL1Loss() if variant == 'upr' else loss_mod.MaskedL1Loss()

    def one():
        opt.zero_grad()
        out = model(*stacks)
        val = crit(out, gt, mask, None) if variant == 'upr' else crit(out, gt, mask)
        val.backward()
        opt.step()
        return val

    one()
    torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(steps):
        val = one()
    torch.cuda.synchronize()
    dt = time.time() - t0
    ms = 1e3 * dt / steps
    res = {'value': round(B * steps / dt, 2), 'unit': 'patches/s', 'steps': steps, 'ms_per_step': round(ms, 2),
           'loss': round(float(val.detach()), 6), 'vs_train_step': round(ref_ms / ms, 4) if ref_ms else None}
    del model, opt, stacks, gt, mask
    torch.cuda.empty_cache()
    return res


def ese_leg(dev, peak, size=512, seed=2):
    """BASELINE.json configs[4] on one GPU: one 512x512 light field through the 70-member Ensamble (eval;
    reference mmlf/model/ensamble.py:58-118)"""
    from mmlf_amd.ensamble import Ensamble
    from mmlf_amd.feed_forward import FeedForward
    torch.manual_seed(0)
    model = FeedForward(**dict(BASE_KW, model_uncert=True)).to(dev).eval()
    ens = Ensamble(model, -3.5, 3.5, 0.1).eval()
    gen = torch.Generator(device=dev).manual_seed(seed)
    stacks = [torch.rand((1, 9, 3, size, size), device=dev, generator=gen) for _ in range(4)]
    times = []
    with torch.no_grad():
        for _ in range(3):
            torch.cuda.synchronize()
            t0 = time.time()
            out = ens(*stacks)
            torch.cuda.synchronize()
            times.append(time.time() - t0)
    dt = min(times[1:])
    tf = 70 * 2529.3 * (size / 512.0) ** 2 / dt / 1e3          # 2529.3 GFLOP per 512^2 UPR forward
    res = {'value': round(dt, 4), 'unit': 's/scene', 'higher_is_better': False, 'members': 70, 'frame': f'{size}x{size}',
           'algorithmic_tflops': round(tf, 1), 'frac_of_peak': round(tf / peak, 4),
           'finite': bool(torch.isfinite(out['mean']).all())}
    del ens, model, stacks, out
    torch.cuda.empty_cache()
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=3)
    ap.add_argument('--warmup', type=int, default=1)
    ap.add_argument('--variant', default='base', choices=['base', 'upr', 'dpp'])
    ap.add_argument('--global-batch', type=int, default=512)
    ap.add_argument('--patch', type=int, default=96)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-f32-leg', action='store_true', help='skip the extra exact-f32-MFMA measurement')
    ap.add_argument('--no-extra-legs', action='store_true', help='skip the UPR / DPP / shard-64 / ESE measurements')
    ap.add_argument('--ese-size', type=int, default=512, help='frame size of the ESE legs (BASELINE.json configs[4]: 512)')
    ap.add_argument('--rccl-self', action='store_true',
                    help='with --gpus 1: run the N > 1 code path of this script (process group on the nccl backend, distributed '
                         'TrainStep, bucket waits, dpp / upr legs, ESE replica reductions) in a group of ONE rank -- a rehearsal of '
                         'every torch.distributed call under RCCL on a one-GPU box; no link traffic, not a scaling measurement')
    ap.add_argument('--backend', default='nccl', choices=['nccl', 'gloo'],
                    help='gloo: rehearsal of the N>1 path with every rank on whatever GPUs exist (one is enough)')
    args = ap.parse_args()

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    global FORCE_DIST
    FORCE_DIST = bool(args.rccl_self)
    assert not FORCE_DIST or world == 1, '--rccl-self is the one-rank rehearsal'
    dist_on = world > 1 or FORCE_DIST          # everything below that talks to torch.distributed
    if dist_on:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29533')
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    assert world == args.gpus, f'--gpus {args.gpus} but WORLD_SIZE={world}'
    assert torch.cuda.is_available(), 'bench.py needs an MI355X'
    dev_index = local_rank % torch.cuda.device_count() if args.backend == 'gloo' else local_rank
    torch.cuda.set_device(dev_index)
    dev = torch.device('cuda', dev_index)
    if dist_on:
        with stdout_to_stderr():         # (communicator set-up and its first collective: whatever the library prints goes to stderr)
            if args.backend == 'nccl':       # RCCL over xGMI
                dist.init_process_group('nccl', rank=rank, world_size=world, device_id=dev)
            else:
                dist.init_process_group('gloo', rank=rank, world_size=world)
            dist.barrier()
            torch.cuda.synchronize()

    from mmlf_amd import engine, _lib

    assert args.global_batch % world == 0
    B = args.global_batch // world
    step, stacks, gt, mask = make_step(args.variant, B, args.patch, dev, seed=rank)

    def sync():
        if dist_on:
            dist.barrier()
        torch.cuda.synchronize()

    it = 1
    for _ in range(args.warmup):
        step(*stacks, gt, mask, it)
        it += 1
    sync()
    engine.PROFILE = []           # (tag, flops, start_event, end_event) of the 280-wide conv / weight-gradient launches
    if step.buckets is not None:
        step.buckets.wait_events = []
    dt, loss, host_ms = timed_steps(step, (stacks, gt, mask), args.steps, it, sync)      # sync: barrier + torch.cuda.synchronize()
    it += args.steps
    prof, engine.PROFILE = engine.PROFILE, None
    peak_gib = torch.cuda.max_memory_allocated(dev) / 2.0 ** 30        # this rank's peak of live tensors over warm-up + timed steps
    # further legs (N=1 only): the same step in the two arithmetic modes that carry no precision asterisk -- the exact-f32
    # MFMA kernels (5 steps) and the exact 3 x bf16 split (3 steps) -- same tensors, same shapes
    mode_legs = {}
    if not dist_on and not args.no_f32_leg:
        for mode_name, nsteps in (('f32', 5), ('bf16x6', 3)):
            if engine.CONV_MODE == mode_name:
                continue
            mode = engine.CONV_MODE
            engine.CONV_MODE = mode_name
            step(*stacks, gt, mask, it)
            sync()
            engine.PROFILE = []
            t1 = time.time()
            for _ in range(nsteps):
                step(*stacks, gt, mask, it)
            sync()
            dt1 = time.time() - t1
            p1, engine.PROFILE = engine.PROFILE, None
            engine.CONV_MODE = mode
            p1 = [r for r in p1 if r[0] == 'conv']
            s1 = sum(r[2].elapsed_time(r[3]) for r in p1) * 1e-3
            a1 = sum(r[1] for r in p1) / s1 / 1e12 if s1 > 0 else 0.0
            pk = PEAK_F32_MFMA_TFLOPS if mode_name == 'f32' else PEAK_BF16_MFMA_TFLOPS / 6
            mode_legs[mode_name] = {
                'value': round(args.global_batch * nsteps / dt1, 3), 'unit': 'patches/s', 'steps': nsteps,
                'ms_per_step': round(1e3 * dt1 / nsteps, 2),
                'kernel': ('conv4tap_kernel<9> (v_mfma_f32_32x32x2_f32)' if mode_name == 'f32'
                           else 'conv4tap_x6s_kernel<18, 3, EPI> (v_mfma_f32_16x16x32_bf16, six cross terms)'),
                'achieved': round(a1, 2), 'peak': round(pk, 1), 'frac': round(a1 / pk, 4),
                'launches': len(p1), 'avg_ms': round(1e3 * s1 / max(1, len(p1)), 3)}
    tmax = torch.tensor([dt, host_ms], dtype=torch.float64, device=dev)
    tmin = tmax.clone()
    if dist_on:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(tmin, op=dist.ReduceOp.MIN)       # a slow rank shows as max well above min
    dt, dt_min, host_ms = float(tmax[0]), float(tmin[0]), float(tmax[1])
    loss_val = float(loss)

    n_buckets = len(step.buckets.ranges) if step.buckets is not None else 0
    grad_bytes = int(step.grad.numel()) * 4
    allreduce_ms, allreduce_by_bucket = allreduce_report(step, dev) if dist_on else (None, None)
    # BASELINE.json configs[4] under --gpus N: replicas only -- every rank runs its own 512x512 light field
    # through the 70-member Ensamble, no collective on the data path; reported as max-over-ranks seconds per scene
    ese_rep = None
    ddp_legs = {}
    if dist_on and not args.no_extra_legs:
        passes_ = {'f16x3': 3, 'bf16x6': 6}.get(engine.CONV_MODE)
        peak_ = PEAK_BF16_MFMA_TFLOPS / passes_ if passes_ else PEAK_F32_MFMA_TFLOPS
        del step, stacks, gt, mask
        torch.cuda.empty_cache()
        sync()
        for v in ('dpp', 'upr'):              # the other variants under the same data parallelism (configs[3] = DPP x 8 GPUs)
            if v != args.variant:
                ddp_legs[v] = ddp_leg(v, args.global_batch, args.patch, dev, 2, world, rank, sync)
        sync()
        mine = ese_leg(dev, peak_, size=args.ese_size, seed=2 + rank)
        worst = torch.tensor([mine['value']], dtype=torch.float64, device=dev)
        ok = torch.tensor([1.0 if mine['finite'] else 0.0], dtype=torch.float64, device=dev)
        dist.all_reduce(worst, op=dist.ReduceOp.MAX)
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        ese_rep = dict(mine, value=round(float(worst), 4), scenes=world, scenes_per_s=round(world / float(worst), 3),
                       finite=bool(float(ok) > 0),
                       note='one light field per GPU, no data-path collective; value = slowest rank, scenes_per_s = N / value')
    if rank == 0:
        value = args.global_batch * args.steps / dt
        wprof = [r for r in prof if r[0].startswith('wgrad')]
        nprof = [r for r in prof if r[0] == 'conv70']
        prof = [r for r in prof if r[0] == 'conv']
        secs = sum(r[2].elapsed_time(r[3]) for r in prof) * 1e-3
        flops = sum(r[1] for r in prof)
        alg_bytes = round(sum(r[4] for r in prof) / max(1, len(prof)))      # true in + out of a launch, mean over the launches
        achieved = flops / secs / 1e12 if secs > 0 else 0.0
        passes = {'f16x3': 3, 'bf16x6': 6}.get(engine.CONV_MODE)
        split = passes is not None
        peak = PEAK_BF16_MFMA_TFLOPS / passes if split else PEAK_F32_MFMA_TFLOPS
        kname = f'conv4tap_x6s_kernel<18, {2 if passes == 3 else 3}' if split else 'conv4tap_kernel<9>'
        dtype = {'f16x3': 'f32 via 2 x f16 operand split (22 significant bits per operand, locally scaled; 3 MFMA '
                          'passes, f32 accumulate)',
                 'bf16x6': 'f32 via exact 3 x bf16 operand split (6 MFMA passes, f32 accumulate)'}.get(engine.CONV_MODE, 'f32')
        traffic = pmc_traffic(kname) if args.global_batch == 512 and world == 1 else (None, 'n/a: not the bs=512 single-GPU shape')
        traffic_build, traffic_same = pmc_build(_lib.build_info()) if traffic[0] else (None, None)
        line = {
            'metric': '96x96 EPI patches/sec fwd+bwd, bs=512', 'value': round(value, 3), 'unit': 'patches/s',
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': round(1e3 * dt / args.steps, 3),
            'higher_is_better': True, 'scaling': 'strong', 'vs_baseline': None,
            'dtype': dtype, 'data': 'synthetic',
            'config': {'workload': workload_name(args.variant, world, args.global_batch, args.patch),
                       'per_gpu_batch': B, 'parallelism': f'dp{world}', 'loss': round(loss_val, 6)},
            'host_enqueue_ms': round(host_ms, 2),
            'whole_step_tflops': round(value * GFLOP_PER_PATCH[args.variant] / 1e3, 2),
            'peak_hbm_gib': round(peak_gib, 1),
            'roofline': {'bound': 'mfma', 'achieved': round(achieved, 2), 'peak': round(peak, 1),
                         'unit': 'TFLOP/s', 'frac': round(achieved / peak, 4),
                         'traffic': traffic[0], 'traffic_source': traffic[1],
                         'traffic_build': traffic_build, 'traffic_build_is_timed_build': traffic_same,
                         'kernel': kname + (', EPI> (280->280 forward + data-gradient launches, all epilogue variants)' if split
                                            else ' (280->280 forward + data-gradient launches)'),
                         'peak_is': (f'dense 16-bit MFMA 2500 TFLOP/s / {passes} passes per f32 product' if split
                                     else 'f32 MFMA 157.3 TFLOP/s'),
                         'frac_of_f32_mfma_peak': round(achieved / PEAK_F32_MFMA_TFLOPS, 4),
                         'algorithmic_bytes': alg_bytes,
                         'traffic_ratio': round(traffic[0] / alg_bytes, 3) if traffic[0] and alg_bytes else None,
                         'launches': len(prof), 'avg_ms': round(1e3 * secs / max(1, len(prof)), 3)},
        }
        line['config']['build'] = _lib.build_info()
        line['config']['conv_cus'] = int(_lib.load().mmlf_conv_cus())
        line['config']['overlap_wgrad'] = int(engine.OVERLAP_WGRAD)
        if nprof:      # the kernel furthest from its roof: the 70 -> 70 stream-layer convolutions, HBM-bound (70 FLOP/B < ridge)
            nsecs = sum(r[2].elapsed_time(r[3]) for r in nprof) * 1e-3
            nbytes = sum(r[4] for r in nprof)
            nflops = sum(r[1] for r in nprof)
            ntraffic = (pmc_traffic('conv4tap_x6s_kernel<5, 2', 5e5, 3e6) if args.global_batch == 512 and world == 1
                        else (None, 'n/a: not the bs=512 single-GPU shape'))
            nalg = round(nbytes / len(nprof))
            line['roofline_narrow'] = {
                'bound': 'hbm', 'achieved': round(nbytes / nsecs / 1e9, 1), 'peak': 8000.0, 'unit': 'GB/s',
                'frac': round(nbytes / nsecs / 1e9 / 8000.0, 4),
                'peak_is': 'HBM3E spec 8 TB/s (MI355X_MICROARCH.md); 6.29 TB/s is the best streaming rate measured on this part',
                'frac_of_measured_hbm': round(nbytes / nsecs / 1e9 / 6290.0, 4),
                'kernel': 'conv4tap_x6s_kernel<5, 2, EPI, 16> (70->70 forward + data-gradient launches of the four stream nets)',
                'traffic': ntraffic[0], 'traffic_source': ntraffic[1], 'algorithmic_bytes': nalg,
                'traffic_ratio': round(ntraffic[0] / nalg, 3) if ntraffic[0] else None,
                'mfma_tflops': round(nflops / nsecs / 1e12, 1), 'launches': len(nprof),
                'avg_ms': round(1e3 * nsecs / len(nprof), 3)}
        if wprof:      # the 280-wide weight gradient: the largest single kernel of the step, same peak definition
            side = [r for r in wprof if r[0] == 'wgrad_side']
            main = [r for r in wprof if r[0] == 'wgrad']
            # the side-stream launches share the CUs with BatchNorm kernels (that is what they are there for): the time
            # between their events is not the kernel's; the roofline figure is taken on the main-stream launches (the same
            # kernel on the same shape, conv2's gradient) and the side-stream average is printed beside it
            wprof = main or wprof
            wsecs = sum(r[2].elapsed_time(r[3]) for r in wprof) * 1e-3
            wach = sum(r[1] for r in wprof) / wsecs / 1e12
            walg = round(sum(r[4] for r in wprof) / len(wprof))
            avg = lambda rs: round(sum(r[2].elapsed_time(r[3]) for r in rs) / len(rs), 3) if rs else None
            wname = f'wgrad4tap_x6w_kernel<3, 9, {2 if passes == 3 else 3}>' if split else 'wgrad4tap_kernel<9>'
            wtraffic = (pmc_traffic(wname.split('<')[0]) if args.global_batch == 512 and world == 1
                        else (None, 'n/a: not the bs=512 single-GPU shape'))
            line['roofline_wgrad'] = {
                'bound': 'mfma', 'achieved': round(wach, 2), 'peak': round(peak, 1), 'unit': 'TFLOP/s',
                'frac': round(wach / peak, 4), 'traffic': wtraffic[0], 'traffic_source': wtraffic[1],
                'kernel': wname + ' + scales / reduce launches (280->280 weight + bias gradient, in the step'
                                  + ('; main-stream launches only: MMLF_OVERLAP_WGRAD=1 runs the conv1 gradients on a side '
                                     'stream beside the BatchNorm-backward kernels, avg_ms_side_stream)' if side and main else ')'),
                'launches': len(wprof), 'avg_ms': round(1e3 * wsecs / len(wprof), 3),
                'avg_ms_main_stream': avg(main), 'avg_ms_side_stream': avg(side),
                'algorithmic_bytes': walg, 'traffic_ratio': round(wtraffic[0] / walg, 3) if wtraffic[0] else None}
        if dist_on:
            line['ms_per_step_by_rank'] = {'min': round(1e3 * dt_min / args.steps, 3), 'max': round(1e3 * dt / args.steps, 3)}
            line['config']['buckets'] = n_buckets
            line['config']['buckets_env'] = os.environ.get('MMLF_GRAD_BUCKETS')
            line['config']['gradient_bytes'] = grad_bytes
            if ese_rep is not None:
                line['ese_replicas'] = ese_rep
            line['config']['buckets_note'] = 'default 3 is provisional: chosen on a gloo rehearsal, not on xGMI (MMLF_GRAD_BUCKETS)'
            line['allreduce_ms'] = allreduce_ms
            line['allreduce_ms_by_bucket'] = allreduce_by_bucket
            line['allreduce_note'] = ('per step, max over ranks: time the compute stream waits for the bucket all-reduces '
                                      'after backward is enqueued (0 = fully overlapped with backward); by_bucket in firing order')
            line['host_enqueue_note'] = ('host_enqueue_ms: wall time until step() returns, max over ranks; a rank is launch-bound when '
                                         'this approaches ms_per_step'
                                         + (' -- NOT under gloo: its waits block the host until the gradient exists, so the figure equals '
                                            'the step time here; under RCCL a wait is a stream dependency' if args.backend != 'nccl' else ''))
            for v, leg in ddp_legs.items():
                line[v] = leg
        if args.backend != 'nccl':
            line['config']['backend'] = args.backend + ' (rehearsal: not an xGMI measurement)'
        if FORCE_DIST:
            line['config']['backend'] = (args.backend + ', --rccl-self: the N > 1 code path in a process group of ONE rank (every '
                                         'torch.distributed call of this script under the backend; no link traffic, not a scaling measurement)')
        if 'f32' in mode_legs:
            line['exact_f32_mfma_path'] = mode_legs['f32']
        if 'bf16x6' in mode_legs:
            line['bf16x6_path'] = mode_legs['bf16x6']
        if mode_legs:      # the same step in the arithmetic modes that are no narrower than the reference's fp32, beside `value`
            line['strict_precision'] = dict({k: v['value'] for k, v in mode_legs.items()}, unit='patches/s',
                                            note='bf16x6 = exact 3 x bf16 operand split, f32 = exact-f32 MFMA; `value` above is the '
                                                 'f16x3 mode (22 significant bits per operand, float32-level error measured)')
        if not dist_on and not args.no_extra_legs and args.global_batch == 512 and args.patch == 96:
            # the other BASELINE.json configs, driver-timed in the same run (single GPU each)
            del step, stacks, gt, mask
            torch.cuda.empty_cache()
            for key, (variant, b, n) in {'upr': ('upr', 512, 2), 'dpp': ('dpp', 512, 2), 'shard64': ('base', 64, 5)}.items():
                if key != args.variant:
                    line[key] = extra_leg(variant, b, 96, dev, n, peak)
            line['shard64']['note'] = 'the per-GPU share of the bs=512 batch on 8 GPUs (configs[3] shape), one GPU, no collective'
            line['shard64']['vs_bs512'] = round(line['shard64']['value'] / value, 4)
            # the reference's own loop shape on the drop-in module (autograd + torch.optim.Adam), driver-timed
            line['autograd_loop'] = {v: autograd_leg(v, 512, 96, dev, 2, line[v]['ms_per_step'] if v in line else 1e3 * dt / args.steps)
                                     for v in ('base', 'upr')}
            line['autograd_loop']['note'] = ('model(h,v,i,d) -> mmlf_amd.loss module -> loss.backward() -> torch.optim.Adam.step() '
                                             '(mmlf/train/cli.py:243-258), head outputs materialised; vs_train_step = TrainStep ms / this ms')
            line['ese'] = ese_leg(dev, peak, size=args.ese_size)
        if world == 1 and not args.no_cpu_baseline:
            line['cpu_baseline'] = cpu_baseline(args.variant, args.patch)
        print(json.dumps(line), flush=True)
    if dist_on:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
