"""Per-scene validation step: counterpart of the loop body of reference
mmlf/validate/cli.py:249-282 (and of the periodic validation in mmlf/train/cli.py:269-306).

The scene goes through the model un-tiled, then masked MSE and BadPix(0.07) are computed with
a `margin`-px frame removed (validate/cli.py:271-280)."""
import torch

from . import loss as loss_mod


@torch.no_grad()
def validate_scene(model, h_views, v_views, i_views, d_views, gt, margin=15):
    """Returns (output dict, mse, badpix).  `model` is a FeedForward or an Ensamble."""
    model.eval()
    mask = loss_mod.create_mask_margin(gt.shape, margin).to(gt.device)
    output = model(h_views, v_views, i_views, d_views)
    mse = loss_mod.MaskedMSELoss()(output, gt, mask)
    badpix = loss_mod.MaskedBadPix()(output, gt, mask)
    return output, mse, badpix


def model_from_checkpoint(path, model_discrete=False, val_disp_min=-3.5, val_disp_max=3.5, val_ensamble=False,
                          val_disp_step=0.1, train_shift=0.0, device='cuda', map_location=None):
    """validate/cli.py:213-241: the network is rebuilt from the checkpoint's own hyper-parameters, with the CLI's overrides
    (`model_discrete`, the disparity range, `train_shift`), its weights loaded, wrapped in an Ensamble under --val_ensamble.
    Returns (model, hyper-parameter dict, parameter count as the CLI prints it)."""
    from .ensamble import Ensamble
    from .feed_forward import FeedForward
    state = torch.load(path, map_location=map_location or device)
    kwargs = dict(state['hyper_parameters'])
    kwargs.update({'model_discrete': model_discrete, 'val_disp_min': val_disp_min, 'val_disp_max': val_disp_max,
                   'train_shift': train_shift})
    model = FeedForward(**kwargs).to(device)
    model.load_state_dict(state['model_state_dict'])
    if val_ensamble:
        model = Ensamble(model, val_disp_min, val_disp_max, val_disp_step)
    return model, kwargs, sum(p.numel() for p in model.parameters())


def table_rows(avg, runtime):
    """the two LaTeX table lines the validate CLI prints at its end (validate/cli.py:350-351); `avg` as validate_scenes
    returns it, `runtime` the LAST scene's (the CLI prints the loop variable)"""
    return ('MSE & BadPix007 & KLD_UM & KLD_MM & KLD & - & TIME \\\\',
            f"{avg['mse']:.3f} & {avg['badpix']:.3f} & {avg['kld_um']:.3f} & {avg['kld_mm']:.3f} & {avg['kld']:.3f} & - & "
            f"{runtime:.3f} \\\\")


@torch.no_grad()
def validate_scenes(model, scenes, val_disp_min=-3.5, val_disp_max=3.5, margin=15, out_dir=None, scene_names=None,
                    n_bins=108, reference_compat=True):
    """The validation loop of reference mmlf/validate/cli.py:249-351 on device tensors: per scene the forward pass
    (FeedForward or Ensamble), masked MSE / BadPix(0.07) with a `margin`-px frame removed, the predictive distribution on
    `n_bins` disparity bins (Ensamble: Laplace mixture of its members; DPP: the posterior; UPR: one Laplace; BASE: the bin
    of the mean), its KL divergence from the multi-plane ground truth over all / multimodal / unimodal pixels, the NLL
    of the ground-truth planes -- and, with out_dir, the result files (`results.save_batch`, hci4d.py:295-413).
    `scenes` yields (h, v, i, d, center, gt, mpi, mask, index) tuples with a batch axis of one, as the reference's
    DataLoader does.  Returns (per-scene list of dicts, dict of averages).

    reference_compat=True (default): the averages are the reference's printed table row.  Its `kl_divergence` and
    `nll_discrete` smooth and renormalise their ARGUMENTS in place, and the loop (validate/cli.py:313-337) hands the same
    arrays to three KL calls in a row: KLD is computed on distributions smoothed once, KLD_MM on ones smoothed twice,
    KLD_UM three times, and for --model_discrete `dist` is the `posterior` array `nll_discrete` has already rewritten
    (divided by 7).  A one-hot BASE distribution is very sensitive to that smoothing (KLD_MM 8.0 against 8.6), so the
    sequence is reproduced, on copies: the model's output dict is not modified and the files hold the original posterior,
    as the reference's do (it saves before it evaluates).  reference_compat=False evaluates every metric on distributions
    smoothed exactly once (what the helper functions compute when called on fresh arrays).
    `runtime` stops where the reference's clock does (validate/cli.py:309): after the forward pass and the host copies,
    before any distribution metric."""
    import time
    from . import dl, metrics, results
    if out_dir is not None and scene_names is None:
        raise ValueError('validate_scenes: out_dir needs scene_names (the dataset\'s scenes_names, indexed by `index`)')
    model.eval()
    inner = getattr(model, 'model', model)                       # Ensamble wraps the network
    inner = getattr(inner, 'module', inner)
    rows = []
    for data in scenes:
        h, v, i_, d, center, gt, mpi, _, index = data
        t0 = time.time()
        output, mse, badpix = validate_scene(model, h, v, i_, d, gt, margin)
        mean, logvar = output['mean'], output.get('logvar')
        means, logvars = output.get('means'), output.get('logvars')
        row = {'mse': float(mse), 'badpix': float(badpix)}       # (float(): the device has finished the forward pass)
        if mean.is_cuda:
            torch.cuda.synchronize(mean.device)
        runtime = time.time() - t0                               # validate/cli.py:309
        if out_dir is not None:
            lmm = None if means is None else torch.stack([means, torch.exp(logvars)], 0)     # :300-303
            results.save_batch(out_dir, scene_names, index, gt=gt, result=mean, uncert=logvar, runtime=runtime,
                               gmm=lmm, nll=output.get('scores'), posterior=output.get('posterior'), center=center,
                               views=(h, v, i_, d))
        seq = bool(reference_compat)
        dist_gt = dl.mpi_to_weights(mpi, val_disp_min, val_disp_max, n_bins)
        mm = metrics.multimodal_mask(mpi)
        if means is not None and logvars is not None:            # validate/cli.py:317-319 (--val_ensamble)
            dist = metrics.lmm_to_discrete(n_bins, val_disp_min, val_disp_max, means, torch.exp(logvars))
            nll = torch.zeros((), dtype=torch.float64, device=mean.device)
        elif output.get('scores') is not None:                   # :320-322 (--model_discrete)
            dist = output['posterior'].clone() if seq else output['posterior']
            weights = dl.mpi_to_weights(mpi, inner.disp_min, inner.disp_max, inner.steps)
            nll = metrics.nll_discrete(weights, dist, val_disp_min, val_disp_max, None, inplace=seq)
        elif logvar is not None:                                 # :323-325 (--model_uncert)
            dist = metrics.laplace_to_discrete(n_bins, val_disp_min, val_disp_max, mean, logvar)
            nll = metrics.nll_laplace(mpi, mean, logvar, None)
        else:                                                    # :326-331
            dist = metrics.mean_to_discrete(n_bins, val_disp_min, val_disp_max, mean)
            nll = metrics.nll_laplace(mpi, mean, torch.zeros_like(mean), None)
        # three calls on the SAME two tensors (validate/cli.py:333-335): each smooths them once more when seq
        row['kld'] = float(metrics.kl_divergence(dist, dist_gt, inplace=seq))
        row['kld_mm'] = float(metrics.kl_divergence(dist, dist_gt, mm, inplace=seq))
        row['kld_um'] = float(metrics.kl_divergence(dist, dist_gt, 1.0 - mm, inplace=seq))
        row['nll'] = float(nll)
        row['runtime'] = runtime
        rows.append(row)
    avg = {k: sum(r[k] for r in rows) / max(1, len(rows)) for k in ('mse', 'badpix', 'kld', 'kld_mm', 'kld_um', 'nll')}
    return rows, avg
