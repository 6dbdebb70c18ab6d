import glob
import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run on the GPU box)')


def golden_layer_files():
    files = sorted(glob.glob(os.path.join(GOLDEN, 'g*.npz')))
    # g12 = initialiser draws, g13 / g17 = whole-model fixtures (tests/test_models.py, tests/test_timit_parity.py)
    return [f for f in files if os.path.basename(f)[:3] not in ('g12', 'g13', 'g17')]


def load_golden(path):
    z = np.load(path)
    rec = {k: z[k] for k in z.files if k != 'config'}
    cfg = json.loads(str(z['config']))
    return rec, cfg


def layer_kwargs(cfg):
    """Split a golden config into (rank, oracle kwargs)."""
    kind = cfg['kind']
    if kind == 'QuaternionDense':
        return 0, dict(activation=cfg.get('activation'))
    rank = int(kind[-2])
    kw = dict(strides=cfg.get('strides', 1), padding=cfg.get('padding', 'valid'),
              data_format=cfg.get('data_format', 'channels_last'),
              dilation_rate=cfg.get('dilation_rate', 1), activation=cfg.get('activation'))
    for k in ('strides', 'dilation_rate'):
        if isinstance(kw[k], list):
            kw[k] = tuple(kw[k])
    return rank, kw


@pytest.fixture(scope='session')
def golden_dir():
    return GOLDEN
