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
