"""Train step: counterpart of the loop body of reference mmlf/train/cli.py:185-258.

``TrainStep`` owns what the reference's loop owns around the model: the loss selection
(train/cli.py:247-255), the 11-px loss margin (:194), the LR warm start / cooling (:233-241),
``zero_grad -> forward -> loss -> backward -> Adam.step`` (:243-258).

Data parallelism replaces ``torch.nn.DataParallel`` (train/cli.py:159): one process per GPU, each
rank trains on its shard of the batch with replica-local BatchNorm statistics (as DataParallel
replicas do), and ONE sum all-reduce of the flat float32 gradient crosses xGMI per step, issued per
bucket as soon as backward has produced it (RCCL: backend "nccl"; gloo on CPU for tests).

On CUDA tensors everything between the input stacks and the updated parameters runs in HIP kernels
(engine.Trunk, mmlf_loss_fwd_bwd, mmlf_adam_step); on CPU tensors (gloo rehearsal, CPU tests) the
module's torch plumbing path + autograd + the same flat Adam arithmetic in torch ops is used.
"""
import math

import torch
import torch.distributed as dist

from . import _lib, dl, loss as loss_mod
from ._lib import call, ptr

KINDS = {'base': loss_mod.KIND_L1, 'upr': loss_mod.KIND_UPR, 'dpp': loss_mod.KIND_CE}


def flatten_parameters(model):
    """Re-point every parameter at a view of ONE flat float32 buffer (state_dict keys, shapes and
    values unchanged).  Returns (flat, [(name, offset, numel)])."""
    params = list(model.named_parameters())
    dev = params[0][1].device
    total = sum(p.numel() for _, p in params)
    flat = torch.empty(total, dtype=torch.float32, device=dev)
    layout, o = [], 0
    for name, p in params:
        n = p.numel()
        flat[o:o + n].copy_(p.detach().reshape(-1))
        p.data = flat[o:o + n].view_as(p)
        layout.append((name, o, n))
        o += n
    return flat, layout


LOG_HEADER = f'{"iter":>7}, loss_train,   loss_val,        mse, badpix_007, time_elapsed'      # train/cli.py:168


def log_line(i, loss_train, loss_val_avg, mse_avg, bad_pix_avg, time_elap):
    """one row of the reference's log.csv / stdout table (train/cli.py:327)"""
    return (f'{i:>7}, {float(loss_train):.8f}, {float(loss_val_avg):.8f}, {float(mse_avg):.8f}, '
            f'{float(bad_pix_avg):.8f}, {float(time_elap):.8f}')


@torch.no_grad()
def validation_pass(val_model, batches, uncert=False, loss_multimodal=False, margin=15, out_dir=None, scene_names=None):
    """The periodic validation inside the training loop (reference train/cli.py:265-318, every --val_interval
    iterations): eval mode; per validation batch the `margin`-px frame mask (--val_loss_margin), the forward pass of
    `val_model` (the network, or its Ensamble under --val_ensamble), the TRAINING loss family on it -- the uncertainty
    loss for --model_uncert, else the L1 loss (also for --model_discrete: the loop validates the DPP head's `mean` with
    L1, :291-300), their multimodal forms on `mpi` under --train_loss_multimodal, never a padding mask -- plus masked MSE
    and BadPix(0.07), and the result files (mean, logvar) in the dataset's save_batch layout.  `batches` yields the
    loader's 9-tuples (h, v, i, d, center, gt, mpi, mask, index).  Returns (loss_val_avg, mse_avg, bad_pix_avg) as
    floats: what ModelSaver (`loss=`) and the log row take (:320-327)."""
    from . import results
    if out_dir is not None and scene_names is None:
        raise ValueError('validation_pass: out_dir needs scene_names (the dataset\'s scenes_names, indexed by `index`)')
    val_model.eval()
    loss_fn = loss_mod.MultiMaskedL1Loss() if loss_multimodal else loss_mod.MaskedL1Loss()
    loss_uncert_fn = loss_mod.ImprovedMultiUncertaintyL1Loss() if loss_multimodal else loss_mod.ImprovedUncertaintyL1Loss()
    mse_fn, bad_pix_fn = loss_mod.MaskedMSELoss(), loss_mod.MaskedBadPix()
    loss_val_avg, mse_avg, bad_pix_avg, n = 0.0, 0.0, 0.0, 0
    for data in batches:
        h, v, i_, d, center, gt, mpi, _, index = data
        mask = loss_mod.create_mask_margin(gt.shape, margin).to(gt.device)
        output = val_model(h, v, i_, d)
        target = mpi if loss_multimodal else gt
        loss_val = (loss_uncert_fn if uncert else loss_fn)(output, target, mask)
        loss_val_avg += float(loss_val)
        mse_avg += float(mse_fn(output, gt, mask))
        bad_pix_avg += float(bad_pix_fn(output, gt, mask))
        if out_dir is not None:                  # valset.save_batch(output_dir, index, mean, logvar), :309-316
            results.save_batch(out_dir, scene_names, index, gt=gt, result=output['mean'], uncert=output.get('logvar'),
                               center=center, views=(h, v, i_, d))
        n += 1
    if n == 0:      # (the reference divides by j + 1 of an empty loop: NameError; an empty validation set is an error here too)
        raise ValueError('validation_pass: no validation batches')
    return loss_val_avg / n, mse_avg / n, bad_pix_avg / n


DEFAULT_BUCKETS = 3


def bucket_count(default=DEFAULT_BUCKETS):
    """MMLF_GRAD_BUCKETS=<n>: how many all-reduces carry the gradient of a step (1 ... 10: at most one per out_net block
    and stream net; default 3).  Fewer, larger messages cost less launch / protocol overhead per byte on point-to-point
    xGMI links and give the collective's kernels fewer chances to queue behind a running 7 ms persistent convolution;
    more of them start earlier under backward.  Why 3 and not 10 (the default until round 5): the only place several
    ranks of this code have run together is the five-rank gloo rehearsal on one GPU, and there ten asynchronous
    all-reduces in flight per step took 94-106 s per step against 0.41-0.43 s with three
    (profiles/r05_bench_5rank_gloo_rehearsal.json; the compute-unit cap made no difference) -- a property of that
    backend's worker threads, maybe, but the message count is the one variable it exposed, and three 6 MB messages leave
    the same ~2 MB tail behind backward as ten.  The first run on real links decides; bench.py reports the value in
    config.buckets."""
    import os
    try:
        n = int(os.environ.get('MMLF_GRAD_BUCKETS', default))
    except ValueError:
        n = default
    return max(1, n)


class GradBuckets:
    """Contiguous slices of the flat gradient, all-reduced as soon as the backward pass has finished the layers they
    cover.  The finest division is one slice per out_net block / stream net (`keys`, in the order backward completes them:
    out_net.7 ... out_net.0, in_net_id, in_net_hv -- the reverse of their order in the flat buffer, so any run of
    consecutive keys is one contiguous range); `n_buckets` < len(keys) coalesces runs of consecutive keys into one
    all-reduce each, issued when the LAST key of the run is ready."""

    def __init__(self, layout, group=None, n_buckets=None):
        self.group = group
        fine = {}
        for name, o, n in layout:
            key = name.split('.')[0] if name.startswith('in_net') else '.'.join(name.split('.')[:2])
            lo, hi = fine.get(key, (o, o + n))
            fine[key] = (min(lo, o), max(hi, o + n))
        # completion order of backward = descending offset in the flat buffer
        self.keys = sorted(fine, key=lambda k: -fine[k][0])
        n_buckets = bucket_count() if n_buckets is None else int(n_buckets)
        n_buckets = max(1, min(n_buckets, len(self.keys)))
        # runs of consecutive keys, as even in count as possible (the wide blocks are equal in bytes; the two stream nets
        # together are 5 % of the gradient and share the last run's tail)
        bounds = [round(i * len(self.keys) / n_buckets) for i in range(n_buckets + 1)]
        self.ranges, self.bucket_of = {}, {}
        for b in range(n_buckets):
            run = self.keys[bounds[b]:bounds[b + 1]]
            lo, hi = min(fine[k][0] for k in run), max(fine[k][1] for k in run)
            assert hi - lo == sum(fine[k][1] - fine[k][0] for k in run), 'bucket keys must be contiguous in the flat gradient'
            self.ranges[run[-1]] = (lo, hi)             # keyed by the key that completes the run
            for k in run:
                self.bucket_of[k] = run[-1]
        self.pending = []
        self.done = set()
        # bench.py: a list that receives, per step, one (start, end) HIP event pair per bucket around that bucket's wait (in the
        # order the buckets were fired: the time the compute stream stood behind each all-reduce after backward was enqueued)
        self.wait_events = None

    def ready(self, flat_grad, key):
        """backward has enqueued every gradient of `key`: fire its bucket if that completes the bucket's run"""
        if key not in self.bucket_of:
            return
        last = self.bucket_of[key]
        if last in self.done or key != last:
            return
        self._fire(flat_grad, last)

    def _fire(self, flat_grad, last):
        self.done.add(last)
        lo, hi = self.ranges[last]
        self.pending.append(dist.all_reduce(flat_grad[lo:hi], op=dist.ReduceOp.SUM, group=self.group, async_op=True))

    def finish(self, flat_grad):
        for last in self.ranges:
            if last not in self.done:
                self._fire(flat_grad, last)
        timed = self.wait_events is not None and flat_grad.is_cuda
        pairs = []
        for w in self.pending:      # how long the compute stream stands behind each collective once backward is through
            if timed:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
            w.wait()
            if timed:
                e1.record()
                pairs.append((e0, e1))
        if timed:
            self.wait_events.append(pairs)
        self.pending, self.done = [], set()


class TrainStep:
    def __init__(self, model, lr, variant=None, warm_start=False, cooling=0, betas=(0.9, 0.999), eps=1e-8,
                 loss_margin=11, process_group=None, loss_multimodal=False, loss_padding=None,
                 train_eval_mode=False, train_eval_mode_start=0, loss_strongest=False, force_distributed=False):
        self.model = model
        # --train_eval_mode / --train_eval_mode_start (train/cli.py:227-230): BatchNorm uses its running statistics
        self.eval_mode, self.eval_mode_start = bool(train_eval_mode), int(train_eval_mode_start)
        self.lr, self.warm_start, self.cooling = float(lr), bool(warm_start), int(cooling)
        self.betas, self.eps, self.margin = betas, float(eps), int(loss_margin)
        self.variant = variant or ('upr' if model.uncert else ('dpp' if model.discrete else 'base'))
        # --train_loss_multimodal: `gt` is the multi-plane tensor (B,P,5,H,W) (train/cli.py:120-123,201-225)
        self.multimodal, self.loss_padding = bool(loss_multimodal), loss_padding
        # --train_loss_strongest (train/cli.py:190-192): `gt` is the multi-plane tensor too, and the target is the depth of
        # the plane with the largest alpha; the CLI rejects it together with --train_loss_multimodal (train/cli.py:61)
        self.strongest = bool(loss_strongest)
        assert not (self.strongest and self.multimodal)
        self.flat, self.layout = flatten_parameters(model)
        self.grad = torch.zeros_like(self.flat)
        self.exp_avg = torch.zeros_like(self.flat)
        self.exp_avg_sq = torch.zeros_like(self.flat)
        self.adam_steps = 0
        # force_distributed: take the data-parallel path (broadcasts, mask-count and bucket all-reduces) also in a process group
        # of ONE rank -- the only way to put RCCL itself under this code on a one-GPU box (tests/test_gpu_ddp.py)
        self.distributed = dist.is_available() and dist.is_initialized() and (dist.get_world_size(process_group) > 1
                                                                              or bool(force_distributed))
        self.group = process_group
        self.world = dist.get_world_size(process_group) if self.distributed else 1
        self.buckets = GradBuckets(self.layout, process_group, self._agreed_bucket_count()) if self.distributed else None
        self._grads = {name: self.grad[o:o + n].view_as(dict(model.named_parameters())[name])
                       for name, o, n in self.layout}
        self._margin_mask = {}
        if self.distributed:   # every replica starts from rank 0's weights and buffers (DataParallel replicate)
            dist.broadcast(self.flat, 0, group=process_group)
            self.sync_buffers()

    # ------------------------------------------------------------------ pieces
    def _agreed_bucket_count(self):
        """MMLF_GRAD_BUCKETS is read per process: ranks that disagreed would issue different numbers and sizes of
        all-reduces (a hang, or gradients summed into the wrong slices).  Rank 0's value is used by everyone, and a rank
        whose own environment said something else says so once."""
        if not self.distributed:
            return None
        mine = bucket_count()
        n = torch.tensor([mine], dtype=torch.int64, device=self.flat.device)
        dist.broadcast(n, 0, group=self.group)
        n = int(n)
        if n != mine:
            import warnings
            warnings.warn(f'MMLF_GRAD_BUCKETS: this rank has {mine}, rank 0 has {n}; using rank 0\'s')
        return n

    def sync_buffers(self):
        """BatchNorm buffers of rank 0 win (DataParallel keeps GPU0's running statistics)."""
        if self.distributed:
            for _, b in self.model.named_buffers():
                dist.broadcast(b, 0, group=self.group)

    def current_lr(self, i):
        lr = self.lr
        if self.warm_start and i <= 1000:            # train/cli.py:233-236 (lr = 0 at i = 0)
            lr = self.lr * float(i) / 1000.0
        if self.cooling > 0 and i >= self.cooling:   # train/cli.py:238-241
            lr = self.lr / (10.0 ** (i / self.cooling - 1.0))
        return lr

    def _mask(self, mask):
        key = (tuple(mask.shape), str(mask.device))
        if key not in self._margin_mask:
            self._margin_mask[key] = loss_mod.create_mask_margin(mask.shape, self.margin).to(mask.device).int()
        return mask.int() * self._margin_mask[key]    # train/cli.py:194

    @staticmethod
    def strongest_gt(mpi):
        """train/cli.py:190-192: per pixel the depth (plane 4) of the layer whose alpha (plane 3) is largest, first one on
        ties (torch.max).  (B, P, 5, H, W) -> (B, H, W); the reference's bare ``.squeeze()`` also drops a batch axis of 1."""
        inds = torch.max(mpi[:, :, 3, :, :], dim=1)[1].unsqueeze(1)
        return torch.gather(mpi[:, :, 4, :, :], dim=1, index=inds).squeeze(1)

    def _den_begin(self, mask):
        """The loss denominator under data parallelism is the GLOBAL count of valid pixels / world size (the reference
        evaluates the loss on the gathered batch, train/cli.py:245-255).  The count's all-reduce is ISSUED here, in front
        of the forward pass, and waited for by `_den_end` in front of the loss kernel (which reads the denominator from a
        device scalar): no rank stands at a rendezvous before its first kernel -- until round 5 this was a blocking
        all-reduce ahead of forward."""
        if not self.distributed:
            return None
        cnt = mask.sum().double().reshape(1)
        return cnt, dist.all_reduce(cnt, group=self.group, async_op=True)

    def _den_end(self, started):
        if not isinstance(started, tuple):      # None, or a denominator the caller already holds (tests)
            return started
        cnt, work = started
        work.wait()                 # (a stream dependency for RCCL; host-blocking for gloo)
        return cnt / self.world

    # ------------------------------------------------------------------ the step
    def __call__(self, h, v, i_, d, gt, mask, iteration):
        """One optimisation step on this rank's shard.  Returns the (rank-local) loss tensor."""
        model = self.model
        if self.eval_mode and iteration >= self.eval_mode_start:
            model.eval()
        else:
            model.train()
        lr = self.current_lr(iteration)
        if self.strongest:
            gt = self.strongest_gt(gt).contiguous()
        mask = self._mask(mask)
        den = self._den_begin(mask)
        self.grad.zero_()
        if h.is_cuda and model._native_ok:
            loss = self._native_fwd_bwd(h, v, i_, d, gt, mask, den)
        else:
            loss = self._torch_fwd_bwd(h, v, i_, d, gt, mask, den)
        if self.distributed:
            self.buckets.finish(self.grad)
        self.adam_steps += 1
        self._adam(lr, 1.0 / self.world)
        return loss

    def _native_fwd_bwd(self, h, v, i_, d, gt, mask, den):
        model = self.model
        p = model._tensor_dict()
        with torch.no_grad():
            out, tape = model._trunk.forward(p, [h, v, i_, d], model.training, True)
            den = self._den_end(den)
            if self.multimodal:
                loss, gout = self._multimodal_loss(out, gt, mask, den)
            else:
                kind = KINDS[self.variant]
                grid, half = None, 0.0
                if kind == loss_mod.KIND_CE:
                    grid = model._grid('torch', out.device)
                    half = (model.disp_max - model.disp_min) / model.steps / 2.0
                pad = self.loss_padding
                if pad is not None and kind == loss_mod.KIND_UPR:     # train/cli.py:221-222: only the UPR loss takes it
                    mp = (torch.abs(gt) < pad).int()
                    aux = self._scaled_aux(self._aux_override(loss_mod.KIND_UPR_PADDED, gt, mp))
                    loss, gout = loss_mod.native_multi_loss(loss_mod.KIND_UPR_PADDED, out, gt, mask, mp, None, 0.0,
                                                            True, den, aux)
                else:
                    loss, gout = loss_mod.native_loss(kind, out, gt, mask, grid, half, True, den)
            on_done = (lambda key: self.buckets.ready(self.grad, key)) if self.distributed else None
            model._trunk.backward(p, tape, gout, self._grads, on_done)
        return loss

    def _padded_mpi(self, mpi):
        """--train_loss_padding with --train_loss_multimodal: planes outside the range lose their alpha
        (train/cli.py:219-220).  The DPP target is built from the UNPADDED planes (:201-204 run before :219)."""
        if self.loss_padding is None or self.variant == 'dpp':
            return mpi
        mpi = mpi.clone()
        mpi[:, :, 3] *= (torch.abs(mpi[:, :, 4]) < self.loss_padding).float()
        return mpi

    def _multimodal_heads_loss(self, heads, mpi, mask):
        model = self.model
        mpi = self._padded_mpi(mpi)
        if self.variant == 'upr':
            return loss_mod.ImprovedMultiUncertaintyL1Loss()(heads, mpi, mask)
        if self.variant == 'dpp':
            tgt = dl.mpi_to_weights(mpi, model.disp_min, model.disp_max, model.steps)
            return loss_mod.MaskedCrossEntropy()(heads, tgt, mask)
        return loss_mod.MultiMaskedL1Loss()(heads, mpi, mask)

    def _aux_override(self, kind, target, mask_padding):
        """whole-batch sums the multimodal UPR / padded UPR losses normalise by, summed over the ranks (the
        reference evaluates the loss on the gathered batch, train/cli.py:245-255)"""
        if not self.distributed:
            return None
        if kind == loss_mod.KIND_MULTI_UPR:
            tot = target[:, :, 3].sum(1)
            aux = torch.stack([tot.sum().double(), (tot < 0.01).sum().double()])
        else:
            aux = torch.stack([mask_padding.sum().double(), torch.zeros((), dtype=torch.float64, device=target.device)])
        dist.all_reduce(aux, group=self.group)
        return aux

    def _multimodal_loss(self, out, mpi, mask, den):
        """multimodal losses over (B,P,5,H,W) targets: fused HIP value + gradient (mmlf_loss_multi_fwd_bwd)"""
        model = self.model
        mpi = self._padded_mpi(mpi)
        kind = {'base': loss_mod.KIND_MULTI_L1, 'upr': loss_mod.KIND_MULTI_UPR, 'dpp': loss_mod.KIND_MULTI_CE}[self.variant]
        grid, half = None, 0.0
        if kind == loss_mod.KIND_MULTI_CE:
            grid = model._grid('torch', out.device)
            half = (model.disp_max - model.disp_min) / model.steps / 2.0
        aux = self._aux_override(kind, mpi, None) if kind == loss_mod.KIND_MULTI_UPR else None
        return loss_mod.native_multi_loss(kind, out, mpi, mask, None, grid, half, True, den, self._scaled_aux(aux))

    def _scaled_aux(self, aux):
        # the kernel forms mean = aux[0] / n_local and n_local / aux[1]: pass the global sums divided by the world size
        return None if aux is None else aux / self.world

    def _torch_fwd_bwd(self, h, v, i_, d, gt, mask, den):
        model = self.model
        for _, p in model.named_parameters():
            p.grad = None
        out = model(h, v, i_, d)
        den = self._den_end(den)
        if self.multimodal:
            loss = self._multimodal_heads_loss(out, gt, mask)
        elif self.variant == 'upr':
            mp = None if self.loss_padding is None else (torch.abs(gt) < self.loss_padding).int()
            loss = loss_mod.ImprovedUncertaintyL1Loss()(out, gt, mask, mp)
        elif self.variant == 'dpp':
            tgt = dl.reg_to_class(gt, model.disp_min, model.disp_max, model.steps)
            loss = loss_mod.MaskedCrossEntropy()(out, tgt, mask)
        else:
            loss = loss_mod.MaskedL1Loss()(out, gt, mask)
        if den is not None:     # global masked mean: rescale the local mean to the shared denominator
            cnt = mask.sum().double()
            loss = loss * (cnt / den.squeeze()).float() if cnt > 0 else loss
        loss.backward()
        for name, p in model.named_parameters():
            self._grads[name].copy_(p.grad)
            p.grad = None
        return loss.detach()

    def _adam(self, lr, grad_scale):
        b1, b2 = self.betas
        t = self.adam_steps
        if self.flat.is_cuda:
            call('mmlf_adam_step', ptr(self.flat), ptr(self.grad), ptr(self.exp_avg), ptr(self.exp_avg_sq),
                 self.flat.numel(), lr, b1, b2, self.eps, t, grad_scale, _lib.stream_ptr())
            return
        with torch.no_grad():
            g = self.grad * grad_scale
            self.exp_avg.lerp_(g, 1 - b1)
            self.exp_avg_sq.mul_(b2).addcmul_(g, g, value=1 - b2)
            bc1, bc2 = 1 - b1 ** t, 1 - b2 ** t
            denom = (self.exp_avg_sq.sqrt() / math.sqrt(bc2)).add_(self.eps)
            self.flat.addcdiv_(self.exp_avg, denom, value=-lr / bc1)

    # ------------------------------------------------------------------ checkpoint compatibility
    def optimizer_state_dict(self):
        """A torch.optim.Adam-shaped state dict (reference dl.py:58-70 stores optimizer.state_dict())."""
        state = {}
        for idx, (name, o, n) in enumerate(self.layout):
            shape = dict(self.model.named_parameters())[name].shape
            state[idx] = {'step': torch.tensor(float(self.adam_steps)),
                          'exp_avg': self.exp_avg[o:o + n].view(shape).clone(),
                          'exp_avg_sq': self.exp_avg_sq[o:o + n].view(shape).clone()}
        group = {'lr': self.lr, 'betas': self.betas, 'eps': self.eps, 'weight_decay': 0, 'amsgrad': False,
                 'maximize': False, 'foreach': None, 'capturable': False, 'differentiable': False,
                 'fused': None, 'decoupled_weight_decay': False, 'params': list(range(len(self.layout)))}
        return {'state': state if self.adam_steps else {}, 'param_groups': [group]}

    def load_optimizer_state_dict(self, sd):
        for idx, (name, o, n) in enumerate(self.layout):
            st = sd['state'].get(idx)
            if st is None:
                continue
            self.exp_avg[o:o + n].copy_(st['exp_avg'].reshape(-1))
            self.exp_avg_sq[o:o + n].copy_(st['exp_avg_sq'].reshape(-1))
            self.adam_steps = int(st['step'])
