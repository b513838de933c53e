"""PFM files, byte-compatible with reference mmlf/utils/pfm.py:6-93 (the format its result writers and the
4D light field benchmark tooling exchange disparity maps in).

``save`` writes what the reference writes for the same array: header ``Pf`` / ``PF``, ``"<width> <height>"``,
the scale as ``%f`` (negated for little-endian data), then the float32 rows in the array's own (logical) order --
no vertical flip: callers flip (reference hci4d.py:358-366 passes ``np.flip(x, 0)``).  ``load`` returns the array
in file order for either endianness.  Tensors (CPU or CUDA) are accepted and copied to host first.
"""
import re
import sys

import numpy as np


def _to_numpy(image):
    if hasattr(image, 'detach'):            # torch tensor, any device
        image = image.detach().cpu().numpy()
    return image


def save(filename, image, scale=1.0):
    """reference pfm.py:56-93"""
    image = _to_numpy(image)
    if image.dtype.name != 'float32':
        raise Exception('Image dtype must be float32.')
    if image.ndim == 3 and image.shape[2] == 3:
        color = True
    elif image.ndim == 2 or (image.ndim == 3 and image.shape[2] == 1):
        color = False
    else:
        raise Exception('Image must have H x W x 3, H x W x 1 or H x W dimensions.')
    endian = image.dtype.byteorder
    if endian == '<' or (endian == '=' and sys.byteorder == 'little'):
        scale = -scale
    with open(filename, 'wb') as f:
        f.write(b'PF\n' if color else b'Pf\n')
        f.write(b'%d %d\n' % (image.shape[1], image.shape[0]))
        f.write(b'%f\n' % scale)
        f.write(np.ascontiguousarray(image).tobytes())      # logical (C) order, also for flipped views


def load(filename):
    """reference pfm.py:6-53: (H, W) or (H, W, 3) float32 in file order; the scale's sign gives the byte order"""
    with open(filename, 'rb') as f:
        header = f.readline().rstrip()
        if header == b'PF':
            color = True
        elif header == b'Pf':
            color = False
        else:
            raise Exception('Not a PFM file.')
        dim = re.match(r'^(\d+)\s(\d+)\s$', f.readline().decode('utf-8'))
        if not dim:
            raise Exception('Malformed PFM header.')
        width, height = map(int, dim.groups())
        scale = float(f.readline().rstrip())
        data = np.frombuffer(f.read(), dtype=('<' if scale < 0 else '>') + 'f4').copy()   # writable, like np.fromfile
    return np.reshape(data, (height, width, 3) if color else (height, width))
