"""TEST INFRASTRUCTURE -- numpy/ctypes front end of the C oracle (oracle/qk_oracle.c).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this
module, and only as the checker.  It restates, on the CPU and in float64, what
complexnn/conv.py:288-345 and complexnn/dense.py:126-164 of the reference compute
(forward) and what TF autodiff of those graphs returns (backward).

Padding helpers restate the third-party semantics the reference relies on
(keras.utils.conv_utils.conv_output_length; tf.nn.convolution SAME / VALID; Keras'
'causal' left padding) -- see SURVEY.md section 7 "hard parts".
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SRC = os.path.join(_HERE, 'qk_oracle.c')
_BUILD = os.path.join(_HERE, '_build')
_SO = os.path.join(_BUILD, 'libqk_oracle.so')


class _Desc(ctypes.Structure):
    _fields_ = [('rank', ctypes.c_int32), ('batch', ctypes.c_int32),
                ('in_sp', ctypes.c_int32 * 3), ('out_sp', ctypes.c_int32 * 3),
                ('cq', ctypes.c_int32), ('fq', ctypes.c_int32),
                ('kernel', ctypes.c_int32 * 3), ('stride', ctypes.c_int32 * 3),
                ('dil', ctypes.c_int32 * 3), ('pad_lo', ctypes.c_int32 * 3),
                ('ch_first', ctypes.c_int32), ('conj', ctypes.c_int32),
                ('relu', ctypes.c_int32), ('has_bias', ctypes.c_int32)]


def build(force=False):
    """Compile the C restatement with gcc (no GPU, no torch)."""
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(_SRC):
        os.makedirs(_BUILD, exist_ok=True)
        subprocess.check_call(['gcc', '-O2', '-std=c99', '-fopenmp', '-shared', '-fPIC', _SRC, '-o', _SO])
    return _SO


_lib = None


def _load():
    global _lib
    if _lib is None:
        lib = ctypes.CDLL(build())
        dp = ctypes.POINTER(ctypes.c_double)
        lib.qko_fwd.argtypes = [ctypes.POINTER(_Desc), dp, dp, dp, dp]
        lib.qko_fwd.restype = None
        lib.qko_bwd.argtypes = [ctypes.POINTER(_Desc)] + [dp] * 7
        lib.qko_bwd.restype = None
        fp, ip = ctypes.POINTER(ctypes.c_float), ctypes.POINTER(ctypes.c_int64)
        lib.qko_fwd_at.argtypes = [ctypes.POINTER(_Desc), fp, dp, dp, ip, ctypes.c_int64, dp]
        lib.qko_dx_at.argtypes = [ctypes.POINTER(_Desc), dp, fp, fp, ip, ctypes.c_int64, dp]
        lib.qko_dw_at.argtypes = [ctypes.POINTER(_Desc), fp, fp, fp, ip, ctypes.c_int64, dp]
        lib.qko_dbias.argtypes = [ctypes.POINTER(_Desc), fp, fp, dp]
        for f in (lib.qko_fwd_at, lib.qko_dx_at, lib.qko_dw_at, lib.qko_dbias):
            f.restype = None
        _lib = lib
    return _lib


# ---------------------------------------------------------------------------------------
# shape helpers (keras.utils.conv_utils.conv_output_length / TF padding arithmetic)
# ---------------------------------------------------------------------------------------
def conv_output_length(n, k, padding, stride, dilation=1):
    if n is None:
        return None
    dk = k + (k - 1) * (dilation - 1)
    if padding in ('same', 'causal'):
        out = n
    elif padding == 'valid':
        out = n - dk + 1
    else:
        raise ValueError(padding)
    return (out + stride - 1) // stride


def tf_pads(n, k, stride, dilation, padding):
    """(pad_lo, pad_hi) TensorFlow applies on one spatial axis."""
    if padding == 'valid':
        return 0, 0
    if padding == 'causal':
        return dilation * (k - 1), 0
    out = -(-n // stride)
    total = max((out - 1) * stride + (k - 1) * dilation + 1 - n, 0)
    return total // 2, total - total // 2


def _tup(v, rank):
    return (v,) * rank if isinstance(v, int) else tuple(v)


def make_desc(x_shape, w_shape, rank, strides=1, padding='valid', data_format='channels_last',
              dilation_rate=1, activation=None, use_bias=True, conj=None):
    """Geometry of one layer call.  rank 0 = QuaternionDense (conj defaults to 1)."""
    d = _Desc()
    d.rank = rank
    d.batch = x_shape[0]
    ch_first = data_format == 'channels_first' and rank > 0
    if rank == 0:
        assert len(x_shape) == 2
        ci = x_shape[1]
        sp = ()
    elif ch_first:
        ci, sp = x_shape[1], tuple(x_shape[2:])
    else:
        ci, sp = x_shape[-1], tuple(x_shape[1:-1])
    assert len(sp) == rank
    d.cq = ci // 4
    d.fq = w_shape[-1] // 4
    ks = tuple(w_shape[:rank])
    st, dl = _tup(strides, rank), _tup(dilation_rate, rank)
    out_sp = []
    for i in range(3):
        if i < rank:
            d.in_sp[i], d.kernel[i], d.stride[i], d.dil[i] = sp[i], ks[i], st[i], dl[i]
            d.pad_lo[i] = tf_pads(sp[i], ks[i], st[i], dl[i], padding)[0]
            d.out_sp[i] = conv_output_length(sp[i], ks[i], padding, st[i], dl[i])
            out_sp.append(d.out_sp[i])
        else:
            d.in_sp[i] = d.out_sp[i] = d.kernel[i] = d.stride[i] = d.dil[i] = 1
            d.pad_lo[i] = 0
    d.ch_first = int(ch_first)
    d.conj = int(rank == 0) if conj is None else int(conj)
    d.relu = int(activation == 'relu')
    assert activation in (None, 'linear', 'relu')
    d.has_bias = int(use_bias)
    co = 4 * d.fq
    if rank == 0:
        y_shape = (x_shape[0], co)
    elif ch_first:
        y_shape = (x_shape[0], co) + tuple(out_sp)
    else:
        y_shape = (x_shape[0],) + tuple(out_sp) + (co,)
    return d, y_shape


def _p(a):
    return None if a is None else a.ctypes.data_as(ctypes.POINTER(ctypes.c_double))


def _c(a):
    return None if a is None else np.ascontiguousarray(a, dtype=np.float64)


def forward(x, w, bias, rank, **kw):
    """y (float64) = activation(hamilton(x, w) + bias).  kw: strides, padding, data_format,
    dilation_rate, activation, conj."""
    x, w, bias = _c(x), _c(w), _c(bias)
    d, y_shape = make_desc(x.shape, w.shape, rank, use_bias=bias is not None, **kw)
    y = np.empty(y_shape, dtype=np.float64)
    _load().qko_fwd(ctypes.byref(d), _p(x), _p(w), _p(bias), _p(y))
    return y


def backward(x, w, bias, dy, rank, y=None, **kw):
    """(dx, dw, dbias) float64 of sum(y*dy); dbias is None without bias."""
    x, w, bias, dy = _c(x), _c(w), _c(bias), _c(dy)
    d, y_shape = make_desc(x.shape, w.shape, rank, use_bias=bias is not None, **kw)
    if y is None:
        y = np.empty(y_shape, dtype=np.float64)
        _load().qko_fwd(ctypes.byref(d), _p(x), _p(w), _p(bias), _p(y))
    y = _c(y)
    dx = np.empty(x.shape, dtype=np.float64)
    dw = np.empty(w.shape, dtype=np.float64)
    db = np.empty((w.shape[-1],), dtype=np.float64) if bias is not None else None
    _load().qko_bwd(ctypes.byref(d), _p(x), _p(w), _p(y), _p(dy), _p(dx), _p(dw), _p(db))
    return dx, dw, db


# ---------------------------------------------------------------------------------------
# sampled evaluation (qko_*_at): the oracle at BASELINE's full sizes, a few thousand
# entries at a time.  Activations are float32 arrays (exact images of bf16 / fp16 / fp32
# device tensors); kernel, bias and results float64.
# ---------------------------------------------------------------------------------------
def _f32(a):
    return None if a is None else np.ascontiguousarray(a, dtype=np.float32)


def _fp(a):
    return None if a is None else a.ctypes.data_as(ctypes.POINTER(ctypes.c_float))


def _idx(idx, limit):
    idx = np.ascontiguousarray(idx, dtype=np.int64).reshape(-1)
    assert idx.size == 0 or (idx.min() >= 0 and idx.max() < limit), 'sample index out of range'
    return idx


def forward_at(x, w, bias, idx, rank, **kw):
    """y.reshape(-1)[idx] of forward(x, w, bias, rank, **kw) -- only those outputs are computed."""
    x, w, bias = _f32(x), _c(w), _c(bias)
    d, y_shape = make_desc(x.shape, w.shape, rank, use_bias=bias is not None, **kw)
    idx = _idx(idx, int(np.prod(y_shape)))
    out = np.empty(idx.size, dtype=np.float64)
    _load().qko_fwd_at(ctypes.byref(d), _fp(x), _p(w), _p(bias), idx.ctypes.data_as(ctypes.POINTER(ctypes.c_int64)),
                       idx.size, _p(out))
    return out


def backward_at(x, w, dy, rank, y=None, dx_idx=None, dw_idx=None, want_dbias=False, **kw):
    """(dx.reshape(-1)[dx_idx], dw.reshape(-1)[dw_idx], dbias) of backward(...), sampled.  `y` (the forward output, for
    the relu mask) is required when activation='relu'; x may be None when dw_idx is None.  Entries not asked for are None."""
    w, dy, y, x = _c(w), _f32(dy), _f32(y), _f32(x)
    x_shape = x.shape if x is not None else kw.pop('x_shape')
    kw.pop('x_shape', None)
    d, y_shape = make_desc(x_shape, w.shape, rank, use_bias=True, **kw)
    assert tuple(dy.shape) == tuple(y_shape), (dy.shape, y_shape)
    assert not d.relu or y is not None, 'the relu mask needs the forward output'
    ip = ctypes.POINTER(ctypes.c_int64)
    dx = dw = db = None
    if dx_idx is not None:
        dx_idx = _idx(dx_idx, int(np.prod(x_shape)))
        dx = np.empty(dx_idx.size, dtype=np.float64)
        _load().qko_dx_at(ctypes.byref(d), _p(w), _fp(y), _fp(dy), dx_idx.ctypes.data_as(ip), dx_idx.size, _p(dx))
    if dw_idx is not None:
        dw_idx = _idx(dw_idx, int(np.prod(w.shape)))
        dw = np.empty(dw_idx.size, dtype=np.float64)
        _load().qko_dw_at(ctypes.byref(d), _fp(x), _fp(y), _fp(dy), dw_idx.ctypes.data_as(ip), dw_idx.size, _p(dw))
    if want_dbias:
        db = np.empty(w.shape[-1], dtype=np.float64)
        _load().qko_dbias(ctypes.byref(d), _fp(y), _fp(dy), _p(db))
    return dx, dw, db
