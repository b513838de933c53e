"""Small numpy helpers shared by the GPU parity tests."""
import numpy as np


def _master(dy, dx, variant):
    if variant == 0:
        return dy, dx
    if variant == 1:
        return dx, dy
    return dx, 1 - dy


def variant_filter(w, variant, inverse=False):
    """w_v[..., dy, dx] = w[..., master(dy, dx)] (or its inverse scatter for gradients)."""
    out = np.empty_like(w)
    for dy in range(2):
        for dx in range(2):
            sy, sx = _master(dy, dx, variant)
            if inverse:
                out[..., sy, sx] = w[..., dy, dx]
            else:
                out[..., dy, dx] = w[..., sy, sx]
    return out
