import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


def load_golden(name):
    return dict(np.load(os.path.join(GOLDEN, name), allow_pickle=False))


TINY_KW = dict(model_ksize=2, model_in_blocks=2, model_out_blocks=3, model_chs=8,
               model_views=9, model_cross=False, model_uncert=False, model_unet=False,
               model_discrete=False, model_no_batchnorm=False,
               model_batchnorm_momentum=0.1, val_disp_min=-3.5, val_disp_max=3.5)
BASE_KW = dict(TINY_KW, model_in_blocks=3, model_out_blocks=8, model_chs=70)
VARIANTS = {'base': {}, 'upr': {'model_uncert': True}, 'dpp': {'model_discrete': True}}


@pytest.fixture(scope='session')
def oracle():
    from oracle import oracle as orc
    orc.build()
    return orc
