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


@torch.no_grad()
def validate_scenes(model, scenes, val_disp_min=-3.5, val_disp_max=3.5, margin=15, out_dir=None, scene_names=None,
                    n_bins=108):
    """The validation loop of reference mmlf/validate/cli.py:249-351 on device tensors: per scene the forward pass
    (FeedForward or Ensamble), masked MSE / BadPix(0.07) with a `margin`-px frame removed, the predictive distribution on
    `n_bins` disparity bins (Ensamble: Laplace mixture of its members; DPP: the posterior; UPR: one Laplace; BASE: the bin
    of the mean), its KL divergence from the multi-plane ground truth over all / multimodal / unimodal pixels, the NLL
    of the ground-truth planes -- and, with out_dir, the result files (`results.save_batch`, hci4d.py:295-413).
    `scenes` yields (h, v, i, d, center, gt, mpi, mask, index) tuples with a batch axis of one, as the reference's
    DataLoader does.  Returns (per-scene list of dicts, dict of averages = the reference's table row)."""
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
        dist_gt = dl.mpi_to_weights(mpi, val_disp_min, val_disp_max, n_bins)
        mm = metrics.multimodal_mask(mpi)
        if means is not None and logvars is not None:            # validate/cli.py:317-319 (--val_ensamble)
            dist = metrics.lmm_to_discrete(n_bins, val_disp_min, val_disp_max, means, torch.exp(logvars))
            nll = torch.zeros((), dtype=torch.float64, device=mean.device)
        elif output.get('scores') is not None:                   # :320-322 (--model_discrete)
            dist = output['posterior']
            weights = dl.mpi_to_weights(mpi, inner.disp_min, inner.disp_max, inner.steps)
            nll = metrics.nll_discrete(weights, output['posterior'], val_disp_min, val_disp_max, None)
        elif logvar is not None:                                 # :323-325 (--model_uncert)
            dist = metrics.laplace_to_discrete(n_bins, val_disp_min, val_disp_max, mean, logvar)
            nll = metrics.nll_laplace(mpi, mean, logvar, None)
        else:                                                    # :326-331
            dist = metrics.mean_to_discrete(n_bins, val_disp_min, val_disp_max, mean)
            nll = metrics.nll_laplace(mpi, mean, torch.zeros_like(mean), None)
        row = {'mse': float(mse), 'badpix': float(badpix), 'kld': float(metrics.kl_divergence(dist, dist_gt)),
               'kld_mm': float(metrics.kl_divergence(dist, dist_gt, mm)),
               'kld_um': float(metrics.kl_divergence(dist, dist_gt, 1.0 - mm)), 'nll': float(nll)}
        if mean.is_cuda:
            torch.cuda.synchronize(mean.device)
        row['runtime'] = time.time() - t0
        if out_dir is not None:
            lmm = None if means is None else torch.stack([means, torch.exp(logvars)], 0)     # :300-303
            results.save_batch(out_dir, scene_names, index, gt=gt, result=mean, uncert=logvar, runtime=row['runtime'],
                               gmm=lmm, nll=output.get('scores'), posterior=output.get('posterior'), center=center,
                               views=(h, v, i_, d))
        rows.append(row)
    avg = {k: sum(r[k] for r in rows) / max(1, len(rows)) for k in ('mse', 'badpix', 'kld', 'kld_mm', 'kld_um', 'nll')}
    return rows, avg
