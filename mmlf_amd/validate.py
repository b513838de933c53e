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
