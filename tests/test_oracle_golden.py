"""CPU: the C oracle (oracle/qk_oracle.c) and the torch-CPU reference port
(oracle/ref_port.py) against every golden fixture generated from the reference's own
layer code (oracle/make_golden.py)."""
import os

import numpy as np
import pytest
import torch

from conftest import golden_layer_files, layer_kwargs, load_golden
from oracle import oracle, ref_port

FILES = golden_layer_files()


def _close(got, want, tol=1e-12):
    scale = max(1.0, float(np.abs(want).max()))
    assert got.shape == want.shape
    err = float(np.abs(got - want).max())
    assert err <= tol * scale, 'max abs err %g (scale %g)' % (err, scale)


@pytest.mark.parametrize('path', FILES, ids=[os.path.basename(f)[:-4] for f in FILES])
def test_c_oracle_matches_reference_golden(path):
    rec, cfg = load_golden(path)
    rank, kw = layer_kwargs(cfg)
    bias = rec.get('bias')
    y = oracle.forward(rec['x'], rec['kernel'], bias, rank, **kw)
    assert list(y.shape) == [s for s in cfg['output_shape']]
    _close(y, rec['y'])
    dx, dw, db = oracle.backward(rec['x'], rec['kernel'], bias, rec['dy'], rank, **kw)
    _close(dx, rec['dx'])
    _close(dw, rec['dkernel'])
    if bias is not None:
        _close(db, rec['dbias'])


@pytest.mark.parametrize('path', FILES, ids=[os.path.basename(f)[:-4] for f in FILES])
def test_torch_port_matches_reference_golden(path):
    rec, cfg = load_golden(path)
    rank, kw = layer_kwargs(cfg)
    x = torch.tensor(rec['x'], dtype=torch.float64, requires_grad=True)
    w = torch.tensor(rec['kernel'], dtype=torch.float64, requires_grad=True)
    b = None
    if 'bias' in rec:
        b = torch.tensor(rec['bias'], dtype=torch.float64, requires_grad=True)
    if rank == 0:
        y = ref_port.dense_forward(x, w, b, activation=kw['activation'])
    else:
        y = ref_port.conv_forward(x, w, b, rank, **kw)
    _close(y.detach().numpy(), rec['y'])
    (y * torch.tensor(rec['dy'], dtype=torch.float64)).sum().backward()
    _close(x.grad.numpy(), rec['dx'])
    _close(w.grad.numpy(), rec['dkernel'])
    if b is not None:
        _close(b.grad.numpy(), rec['dbias'])


def test_dense_is_not_plain_hamilton():
    """dense.py:139-143 builds the TRANSPOSED table (conj(W) (x) x); guard against 'fixing' it."""
    rng = np.random.RandomState(0)
    x = rng.randn(3, 8)
    w = rng.randn(2, 8)
    y_dense = oracle.forward(x, w, None, 0, activation=None)
    y_conv_table = oracle.forward(x, w, None, 0, activation=None, conj=0)
    assert np.abs(y_dense - y_conv_table).max() > 1e-3
    # conj(W) (x) x == hamilton with negated imaginary parts
    wc = w.copy()
    wc[:, 2:] *= -1
    _close(oracle.forward(x, wc, None, 0, activation=None, conj=0), y_dense)


def test_tf_same_padding_rule():
    # extra pad goes right/bottom for even totals (tf.nn.convolution SAME)
    assert oracle.tf_pads(17, 4, 2, 1, 'same') == (1, 2)
    assert oracle.tf_pads(10, 3, 1, 1, 'same') == (1, 1)
    assert oracle.tf_pads(15, 3, 1, 2, 'causal') == (4, 0)
    assert oracle.conv_output_length(17, 4, 'same', 2) == 9
    assert oracle.conv_output_length(19, 3, 'valid', 1, 2) == 15
