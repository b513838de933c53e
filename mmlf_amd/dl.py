"""Bin <-> disparity helpers, reference mmlf/utils/dl.py:109-131,160-182."""
import torch


def reg_to_class(arr, start, stop, n_steps):
    """Continuous disparity -> one-hot over n_steps bins (reference dl.py:109-131).
    The bin half-width is (stop-start)/n_steps/2 while the bin spacing is (stop-start)/(n_steps-1),
    so values between two bins hit NO class; that behaviour is preserved."""
    half = (stop - start) / n_steps / 2.0
    centres = torch.linspace(start, stop, n_steps).view(1, -1, 1, 1).to(arr.device)
    return (torch.abs(centres - arr.unsqueeze(1)) < half).float()


def class_to_reg(arr, start, stop, n_steps):
    """One-hot (or multi-hot on ties) -> disparity (reference dl.py:160-182)."""
    centres = torch.linspace(start, stop, n_steps).view(1, -1, 1, 1).to(arr.device)
    return torch.sum(centres * arr, 1)


def mpi_to_weights(arr, start, stop, n_steps):
    """Multi-plane target (B, P, 5, H, W) -> per-bin weights (B, n_steps, H, W): plane p adds its
    alpha (channel 3) to every bin whose centre is within half a bin of its disparity (channel 4).
    Reference dl.py:134-157."""
    half = (stop - start) / n_steps / 2.0
    centres = torch.linspace(start, stop, n_steps).view(1, -1, 1, 1, 1).to(arr.device)
    alpha = arr[:, :, 3].unsqueeze(1)
    disp = arr[:, :, 4].unsqueeze(1)
    return ((torch.abs(centres - disp) < half).float() * alpha).sum(2)


class ModelSaver:
    """Checkpoint writer with the reference's file format (dl.py:7-74): a dict with
    'model_state_dict', 'optimizer_state_dict', 'hyper_parameters', 'epoch', 'iteration', 'loss'
    (+ extra kwargs), written with torch.save.  `optimizer` may be a torch optimizer or a
    mmlf_amd.train.TrainStep (which emits a torch.optim.Adam-shaped state dict)."""

    def __init__(self, only_best=False):
        self.only_best = only_best
        self.best_loss = None

    def __call__(self, fname, model, optimizer=None, hyper_parameters=None, epoch=None, iteraration=None,
                 loss=None, **kwargs):
        if self.only_best and loss is not None:
            if self.best_loss is not None and self.best_loss < loss:
                return
            self.best_loss = loss
        module = getattr(model, 'module', model)          # DataParallel / DDP-style wrappers
        opt_state = None
        if optimizer is not None:
            opt_state = (optimizer.optimizer_state_dict() if hasattr(optimizer, 'optimizer_state_dict')
                         else optimizer.state_dict())
        state = {'model_state_dict': {k: v.detach().clone() for k, v in module.state_dict().items()},
                 'optimizer_state_dict': opt_state, 'hyper_parameters': hyper_parameters, 'epoch': epoch,
                 'iteration': iteraration, 'loss': loss}
        state.update(kwargs)
        torch.save(state, fname)


def load_checkpoint(fname, model, optimizer=None, lr=None, map_location=None):
    """Resume like reference train/cli.py:137-157: drop 'tmp' keys, load model and optimizer state,
    force the learning rate, return the stored iteration."""
    state = torch.load(fname, map_location=map_location)
    sd = {k: v for k, v in state['model_state_dict'].items() if 'tmp' not in k}
    getattr(model, 'module', model).load_state_dict(sd)
    if optimizer is not None and state.get('optimizer_state_dict') is not None:
        if hasattr(optimizer, 'load_optimizer_state_dict'):
            optimizer.load_optimizer_state_dict(state['optimizer_state_dict'])
            if lr is not None:
                optimizer.lr = float(lr)
        else:
            optimizer.load_state_dict(state['optimizer_state_dict'])
            if lr is not None:
                for g in optimizer.param_groups:
                    g['lr'] = lr
    return state.get('iteration'), state
