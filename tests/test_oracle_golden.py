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


@pytest.mark.parametrize('path', FILES, ids=[os.path.basename(f)[:-4] for f in FILES])
def test_sampled_oracle_entry_points_match_reference_golden(path):
    """qko_fwd_at / qko_dx_at / qko_dw_at / qko_dbias (the oracle at sampled indices, used by the full-size GPU parity
    tests) at EVERY index of every fixture: (1) to 1e-12 against the full oracle on the same float32-representable
    operands (the sampled entry points take float32 activations), (2) against the values the reference's own code
    produced, to the float32 rounding of the operands (2e-6)."""
    rec, cfg = load_golden(path)
    rank, kw = layer_kwargs(cfg)
    bias = rec.get('bias')
    x32, dy32 = rec['x'].astype(np.float32), rec['dy'].astype(np.float32)
    w = rec['kernel']
    y_full = oracle.forward(x32, w, bias, rank, **kw)
    y_at = oracle.forward_at(x32, w, bias, np.arange(y_full.size), rank, **kw).reshape(y_full.shape)
    _close(y_at, y_full)
    _close(y_at, rec['y'], tol=2e-6)
    y32 = y_full.astype(np.float32)               # relu mask: y > 0 survives the rounding (no float32 underflow at these magnitudes)
    assert np.array_equal(y32 > 0, y_full > 0)
    dx_f, dw_f, db_f = oracle.backward(x32, w, bias, dy32, rank, y=y_full, **kw)
    dx, dw, db = oracle.backward_at(x32, w, dy32, rank, y=y32, dx_idx=np.arange(x32.size), dw_idx=np.arange(w.size),
                                    want_dbias=True, **kw)
    _close(dx.reshape(x32.shape), dx_f)
    _close(dw.reshape(w.shape), dw_f)
    _close(db, db_f if db_f is not None else oracle.backward(x32, w, np.zeros(w.shape[-1]), dy32, rank, y=y_full, **kw)[2])
    # the reference's values: only where no output sits within rounding distance of the relu kink
    if kw.get('activation') != 'relu' or np.abs(rec['y'][np.abs(rec['y']) > 0]).min() > 1e-5:
        _close(dx.reshape(x32.shape), rec['dx'], tol=5e-6)
        _close(dw.reshape(w.shape), rec['dkernel'], tol=5e-6)
        if bias is not None:
            _close(db, rec['dbias'], tol=5e-6)


def test_sampled_oracle_subset_and_order():
    """Arbitrary index subsets in arbitrary order, x omitted when only dx is wanted."""
    rng = np.random.RandomState(5)
    x = rng.randn(2, 6, 7, 8).astype(np.float32)
    w = rng.randn(3, 2, 2, 12)
    b = rng.randn(12)
    kw = dict(strides=(1, 2), padding='same', activation='relu')
    y = oracle.forward(x, w, b, 2, **kw)
    dy = rng.randn(*y.shape).astype(np.float32)
    dx_f, dw_f, db_f = oracle.backward(x, w, b, dy, 2, y=y, **kw)
    iy = rng.permutation(y.size)[:50]
    _close(oracle.forward_at(x, w, b, iy, 2, **kw), y.reshape(-1)[iy])
    ix, iw = rng.permutation(x.size)[:40], rng.permutation(w.size)[:30]
    dx, dw, db = oracle.backward_at(None, w, dy, 2, y=y, dx_idx=ix, x_shape=x.shape, **kw)
    assert dw is None and db is None
    _close(dx, dx_f.reshape(-1)[ix])
    dx, dw, db = oracle.backward_at(x, w, dy, 2, y=y, dw_idx=iw, want_dbias=True, **kw)
    assert dx is None
    _close(dw, dw_f.reshape(-1)[iw])
    _close(db, db_f)
