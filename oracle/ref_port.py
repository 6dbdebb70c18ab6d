"""TEST / BASELINE INFRASTRUCTURE -- the reference's CPU op sequence, restated in torch-CPU.

Used ONLY as (a) a second, independent checker in tests/ and (b) bench.py's
`cpu_baseline` leg (kind = "port"): the same op sequence the reference executes every
step on its Keras/TensorFlow CPU backend -- slice the compact kernel into r,i,j,k
(conv.py:294-307 / dense.py:131-134), build the 4x-expanded real kernel out of signed
copies (conv.py:327-331 / dense.py:139-143), run ONE real convolution / matmul
(conv.py:334 / dense.py:149), add the bias (conv.py:336-341 / dense.py:159-160), apply
the activation (conv.py:342-343 / dense.py:161-162); backward by autograd through that
graph, which is what TF autodiff does for the reference.  oneDNN/MKL stand in for
TF's Eigen/MKL kernels.  Never imported by the product path.
"""
import torch
import torch.nn.functional as F

from .oracle import tf_pads

# rows: input component a, cols: output component b -> (sign, part) with part = a ^ b
_CONV_SIGN = ((1, 1, 1, 1), (-1, 1, 1, -1), (-1, -1, 1, 1), (-1, 1, -1, 1))


def expand_kernel(w, conj=False):
    """(*k, Cq, 4F) compact -> (*k, 4Cq, 4F) real kernel (conv table, or its transpose)."""
    parts = torch.chunk(w, 4, dim=-1)
    cols = []
    for b in range(4):
        rows = []
        for a in range(4):
            s = _CONV_SIGN[b][a] if conj else _CONV_SIGN[a][b]
            rows.append(parts[a ^ b] if s > 0 else -parts[a ^ b])
        cols.append(torch.cat(rows, dim=-2))
    return torch.cat(cols, dim=-1)


def _tup(v, rank):
    return (v,) * rank if isinstance(v, int) else tuple(v)


def conv_forward(x, w, bias, rank, strides=1, padding='valid', data_format='channels_last',
                 dilation_rate=1, activation=None):
    st, dl = _tup(strides, rank), _tup(dilation_rate, rank)
    wk = expand_kernel(w)
    if data_format == 'channels_last':
        x = x.movedim(-1, 1)
    pads = []
    for ax in reversed(range(rank)):
        lo, hi = tf_pads(x.shape[2 + ax], w.shape[ax], st[ax], dl[ax], padding)
        pads += [lo, hi]
    if any(pads):
        x = F.pad(x, pads)
    wt = wk.permute(rank + 1, rank, *range(rank))
    y = (F.conv1d, F.conv2d, F.conv3d)[rank - 1](x, wt, bias, st, 0, dl)
    if data_format == 'channels_last':
        y = y.movedim(1, -1)
    if activation == 'relu':
        y = torch.relu(y)
    return y


def dense_forward(x, w, bias, activation=None):
    y = x @ expand_kernel(w, conj=True)
    if bias is not None:
        y = y + bias
    if activation == 'relu':
        y = torch.relu(y)
    return y
