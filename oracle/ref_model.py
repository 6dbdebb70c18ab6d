"""TEST / BASELINE INFRASTRUCTURE -- the reference's TIMIT model as the CPU op sequence it executes.

getTimitModel2D (models/interspeech_model.py:45-185) restated on torch-CPU through oracle/ref_port.py (expand the
compact kernel by signed concatenation every step -> ONE real convolution / matmul -> bias -> activation), with the
stock layers around it (MaxPooling2D((1,3),'same') over the frequency axis :103, PReLU(shared_axes=[1,0]) :99-101,
Permute + reshape :140-143, TimeDistributed dense :149-157, Dense(62, softmax) :173); backward by torch autograd,
which is what TF autodiff does for the reference.  Used by bench.py's `cpu_baseline` leg (kind = "port") and by
tests/ as an independent checker.  Never imported by the product path.
"""
import numpy as np
import torch

from . import ref_port


def init_params(num_layers=10, start_filter=32, seed=0, dtype=torch.float32, prelu=False):
    """Random parameters with the reference's shapes (values do not matter for timing / structure checks)."""
    rng = np.random.RandomState(seed)
    n, sf = num_layers, start_filter

    def P(*s):
        return torch.tensor(rng.randn(*s) / np.sqrt(max(np.prod(s[:-1]), 1) * 4.0), dtype=dtype, requires_grad=True)
    widths = [sf] * (n // 2) + [2 * sf] * (n // 2)
    p = {'conv': (P(3, 5, 1, 4 * sf), P(4 * sf)), 'convs': [], 'alphas': None}
    cin = sf
    for w in widths:
        p['convs'].append((P(3, 5, cin, 4 * w), P(4 * w)))
        cin = w
    p['dense'] = [(P(14 * cin, 256), P(256)), (P(64, 256), P(256)), (P(64, 256), P(256))]
    p['pred'] = (P(256, 62), P(62))
    if prelu:
        shapes = [(1, 41, 1)] + [(1, 14, 1)] * n + [(1, 1)] * 3
        p['alphas'] = [torch.tensor(0.05 + 0.3 * rng.rand(*s), dtype=dtype, requires_grad=True) for s in shapes]
    return p


def leaves(p):
    out = list(p['conv']) + [t for wb in p['convs'] for t in wb] + [t for wb in p['dense'] for t in wb] + list(p['pred'])
    return out + (list(p['alphas']) if p['alphas'] is not None else [])


def ctc_cost(pred, labels, input_length, label_length):
    """K.ctc_batch_cost (interspeech_model.py:37-39) restated on torch-CPU: Keras hands log(y_pred + 1e-7) to tf.nn.ctc_loss
    as logits, which normalises them again; blank = last class.  Per-sample cost (B,)."""
    logp = torch.log_softmax(torch.log(pred + 1e-7), dim=-1).transpose(0, 1)
    return torch.nn.functional.ctc_loss(logp, labels.long(), input_length.reshape(-1).long(), label_length.reshape(-1).long(),
                                        blank=pred.shape[-1] - 1, reduction='none', zero_infinity=False)


def timit_forward(x, p, act='relu', dropout=0.0):
    """x (B, 4, 41, T) channels_first -> posteriors (B, T, 62).  dropout: Dropout(d.dropout) behind every body convolution
    and the first two dense layers (interspeech_model.py:117-121,131-137,150-154), training mode."""
    drop = (lambda h: torch.nn.functional.dropout(h, dropout, True)) if dropout > 0 else (lambda h: h)
    alphas = p['alphas']
    a = None if alphas is not None else act
    pl = (lambda h, k: torch.relu(h) - alphas[k] * torch.relu(-h)) if alphas is not None else (lambda h, k: h)
    kw = dict(padding='same', data_format='channels_first', activation=a)
    h = pl(ref_port.conv_forward(x, p['conv'][0], p['conv'][1], 2, **kw), 0)
    h = torch.nn.functional.max_pool2d(h, (3, 1), (3, 1), ceil_mode=True)     # 'same': high-side padding only (41 -> 14)
    n = len(p['convs'])
    for i, (w, b) in enumerate(p['convs']):
        h = drop(pl(ref_port.conv_forward(h, w, b, 2, **kw), 1 + i))
    bsz, t = h.shape[0], h.shape[3]
    h = h.permute(0, 3, 1, 2).reshape(bsz, t, -1)
    for i, (w, b) in enumerate(p['dense']):
        h = pl(ref_port.dense_forward(h.reshape(bsz * t, -1), w, b, activation=a).reshape(bsz, t, -1), 1 + n + i)
        if i < 2:
            h = drop(h)
    return torch.softmax(h @ p['pred'][0] + p['pred'][1], dim=-1)
