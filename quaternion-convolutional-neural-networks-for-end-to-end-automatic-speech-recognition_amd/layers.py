"""Stock (real-valued) Keras layers the reference's model builders put AROUND the quaternion
layers (models/example_model.py, models/interspeech_model.py): pooling, Flatten, Dense, Dropout,
PReLU, TimeDistributed, Permute.  They are callers of the hot path, not part of it, and are plain
torch ops with the Keras/TensorFlow semantics the models rely on:

  * 'same' pooling pads like TensorFlow (extra cell on the high side); average pooling does NOT
    count the padding in its divisor;
  * MaxPooling2D/AveragePooling without `data_format` pool axes 1..rank of the tensor as given
    (the stock channels_last default the TIMIT model inherits, interspeech_model.py:103 -- for its
    (B, 4F, 41, T) tensor this pools the 41-bin frequency axis 41 -> 14).
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

from . import _lib as L
from . import functional as Fq
from ._shape import normalize_tuple, tf_pads
from .keras_like import Layer, activations, initializers, regularizers


class Flatten(Layer):
    def call(self, inputs):
        return inputs.reshape(inputs.shape[0], -1)

    def compute_output_shape(self, input_shape):
        return (input_shape[0], int(np.prod(input_shape[1:])))


class Dropout(Layer):
    def __init__(self, rate, **kwargs):
        super(Dropout, self).__init__(**kwargs)
        self.rate = rate

    def call(self, inputs):
        return F.dropout(inputs, self.rate, self.training)


class _TallDenseFn(torch.autograd.Function):
    """x @ W (+ b) for MANY rows of a 16-bit device tensor and an fp32 master kernel (the Dense(62) behind the TIMIT
    head: 51 200 rows x 256 -> 62).  Plain autograd leaves the kernel gradient to one 62 x 256 GEMM with a 51 200-long
    reduction, which hipBLASLt runs on a handful of workgroups (155 us); here the reduction is split into 32 batched
    slices with fp32 outputs that are summed (33 us, and accumulated in fp32 instead of 16 bits)."""

    SPLITS = 32

    @staticmethod
    def forward(ctx, x, w, b):
        w16 = w.to(x.dtype)
        ctx.save_for_backward(x, w16)
        ctx.has_bias = b is not None
        out = x @ w16
        return out + b.to(x.dtype) if b is not None else out

    @staticmethod
    def backward(ctx, dy):
        x, w16 = ctx.saved_tensors
        dy = dy.contiguous()
        dx = dy @ w16.t() if ctx.needs_input_grad[0] else None
        dw = db = None
        if ctx.needs_input_grad[1]:
            s = _TallDenseFn.SPLITS
            dw = torch.bmm(x.view(s, -1, x.shape[1]).transpose(1, 2), dy.view(s, -1, dy.shape[1]), out_dtype=torch.float32).sum(0)
        if ctx.has_bias and ctx.needs_input_grad[2]:
            db = dy.float().sum(0)
        return dx, dw, db


def _cast_cached(w, dtype):
    """w.to(dtype), kept on the parameter until it changes (version counters as in functional._Call._ws: the fused Adam bumps
    the flat buffer's, torch ops the parameter's own)."""
    base = getattr(w, '_qk_flat_base', None)
    ver = (w._version, -1 if base is None else base._version, w.data_ptr())
    hit = w.__dict__.get('_qk_cast16')
    if hit is None or hit[0] != ver or hit[1].dtype != dtype:
        hit = (ver, w.detach().to(dtype))
        w.__dict__['_qk_cast16'] = hit
        Fq._CAST_PARAMS[id(w)] = w                    # (functional.invalidate_cached_kernels: raw `.data` writes bump no counter)
    return hit[1]


class _DenseSoftmaxFn(torch.autograd.Function):
    """softmax(x @ W + b) for MANY rows of a 16-bit device tensor and fp32 master weights -- the model's output layer,
    TimeDistributed(Dense(62, activation='softmax')) on 51 200 rows (interspeech_model.py:171-175) -- as ONE hand-written launch per
    direction (round 6: qk_dense_softmax_fwd / _bwd, csrc/qk_out_layer.hip): forward reads the fp32 master kernel itself (no 16-bit
    copy, no cast launch) and the softmax sees fp32 logits; backward forms d logits, d x, and ADDS the kernel and bias gradients
    into fp32 buffers -- the parameters' own gradient views when they live in a dp.FlatParams buffer.  Until round 6: a library GEMM
    with fp32 logits + qk_softmax_rows_fwd forward, qk_softmax_rows_bwd + two library GEMMs + a reduction backward (7 launches,
    the last hipBLASLt kernels of the step); that composition remains for widths the kernel does not take."""

    @staticmethod
    def forward(ctx, x, w, b):
        ctx.fused = Fq.dense_softmax_supported(x, w.shape[1]) and w.is_contiguous() and (b is None or b.is_contiguous())
        ctx.params = (w, b)
        if ctx.fused:
            y = Fq.dense_softmax_fwd(x, w.detach(), b.detach() if b is not None else None)
            ctx.save_for_backward(x, w, y)
            return y
        w16 = _cast_cached(w, x.dtype)
        logits = torch.mm(x, w16, out_dtype=torch.float32)
        y = Fq.softmax_rows_fwd(logits, b, x.dtype)
        ctx.save_for_backward(x, w16, y)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, wk, y = ctx.saved_tensors
        w, b = ctx.params
        want_w = ctx.needs_input_grad[1]
        want_b = b is not None and ctx.needs_input_grad[2]
        if ctx.fused:
            direct = Fq._direct_grad(w, b, want_w, want_b) if want_w and (b is None or want_b) else None
            if direct is not None:
                dw, db = direct
            else:
                dw = torch.zeros(w.shape, dtype=torch.float32, device=y.device) if want_w else None
                db = torch.zeros(b.shape, dtype=torch.float32, device=y.device) if want_b else None
            dx = Fq.dense_softmax_bwd(x, wk.detach(), y, dy, dw, db)
            if direct is not None:
                Fq._grad_ready(w, b)              # the kernel added both gradients into the flat buffer itself
                dw = db = None
            return (dx if ctx.needs_input_grad[0] else None), dw, db
        w16 = wk
        direct_b = want_b and getattr(b, '_qk_direct_grad', False) and b.grad is not None and b.grad.dtype == torch.float32
        db = None
        if want_b:
            db = b.grad if direct_b else torch.zeros(b.shape, dtype=torch.float32, device=y.device)
        dl = Fq.softmax_rows_bwd(y, dy.contiguous(), db)
        dx = dl @ w16.t() if ctx.needs_input_grad[0] else None
        dw = None
        if want_w:
            s = _TallDenseFn.SPLITS
            dw = torch.bmm(x.view(s, -1, x.shape[1]).transpose(1, 2), dl.view(s, -1, dl.shape[1]), out_dtype=torch.float32).sum(0)
        if direct_b:
            Fq._grad_ready(b)                 # the kernel added the bias gradient into the flat buffer itself
            db = None
        return dx, dw, db


class _DenseSoftmaxCtcMeanFn(torch.autograd.Function):
    """mean over the batch of K.ctc_batch_cost(labels, softmax(x @ W + b), ...) -- the quantity training the reference model
    minimises (interspeech_model.py:37-39,171-178 under the usual `loss={'ctc': lambda y_true, y_pred: y_pred}` compile) -- as ONE
    autograd node: forward = qk_dense_softmax_fwd, qk_ctc_batch_cost (cost AND d cost / d y_pred), one reduction for the mean;
    backward = qk_dense_softmax_bwd on the stored gradient, the upstream scalar handed over as a DEVICE pointer and 1 / batch
    x loss_scale as a host factor.  As separate nodes the mean's backward, the broadcast and the 3.2 M-element product with the
    upstream gradient were four framework launches between the CTC kernel and the output layer's backward."""

    @staticmethod
    def forward(ctx, x, w, b, labels, input_length, label_length, batch, loss_scale):
        y = Fq.dense_softmax_fwd(x, w.detach(), b.detach() if b is not None else None)
        cost, dpred = Fq.ctc_cost_and_grad(y.view(batch, -1, y.shape[1]), labels, input_length, label_length)
        ctx.save_for_backward(x, w, y, dpred)
        ctx.params, ctx.scale = (w, b), float(loss_scale) / batch
        return cost.mean()

    @staticmethod
    def backward(ctx, g):
        x, wk, y, dpred = ctx.saved_tensors
        w, b = ctx.params
        want_w, want_b = ctx.needs_input_grad[1], b is not None and ctx.needs_input_grad[2]
        direct = Fq._direct_grad(w, b, want_w, want_b) if want_w and (b is None or want_b) else None
        if direct is not None:
            dw, db = direct
        else:
            dw = torch.zeros(w.shape, dtype=torch.float32, device=y.device) if want_w else None
            db = torch.zeros(b.shape, dtype=torch.float32, device=y.device) if want_b else None
        g = g.detach().reshape(1).float()
        dx = Fq.dense_softmax_bwd(x, wk.detach(), y, dpred.view(y.shape), dw, db, dy_scale_dev=g, dy_scale=ctx.scale)
        if direct is not None:
            Fq._grad_ready(w, b)
            dw = db = None
        return (dx if ctx.needs_input_grad[0] else None), dw, db, None, None, None, None, None


def dense_softmax_ctc_mean(features, dense, labels, input_length, label_length, loss_scale=1.0):
    """mean_b K.ctc_batch_cost(labels, TimeDistributed(dense)(features), input_length, label_length) for `features` (B, T, in_dim) and
    a built softmax `Dense` layer, through _DenseSoftmaxCtcMeanFn when the hand-written kernels take the shapes; None otherwise (the
    caller composes it from dense(features), ctc_batch_cost and .mean()).  loss_scale multiplies the gradient only."""
    if (not features.is_cuda or features.dim() != 3 or not dense.built or dense.kernel.dtype != torch.float32
            or activations.serialize(dense.activation) != 'softmax' or L.dbg(L.QK_DBG_NO_FUSED_SOFTMAX | L.QK_DBG_NO_FUSED_CTC)):
        return None
    b, t, k = features.shape
    x2 = features.reshape(b * t, k)
    if not (b > 0 and x2.is_contiguous() and Fq.dense_softmax_supported(x2, dense.units) and dense.kernel.is_contiguous()
            and (dense.bias is None or dense.bias.is_contiguous())):
        return None
    # (functional.ctc_supported on the posteriors this node will produce)
    if not (labels.dim() == 2 and labels.shape[1] <= 127 and (t + 8 * labels.shape[1] + 4 + 2 * dense.units + 4) * 4 <= 64 * 1024):
        return None
    return _DenseSoftmaxCtcMeanFn.apply(x2, dense.kernel, dense.bias, labels, input_length, label_length, b, float(loss_scale))


class Dense(Layer):
    """keras.layers.Dense on the last axis (kernel (in, units), glorot_uniform by default)."""

    def __init__(self, units, activation=None, use_bias=True, kernel_initializer='glorot_uniform',
                 bias_initializer='zeros', kernel_regularizer=None, bias_regularizer=None, **kwargs):
        super(Dense, self).__init__(**kwargs)
        self.units, self.use_bias = units, use_bias
        self.activation = activations.get(activation)
        self.kernel_initializer = initializers.get(kernel_initializer)
        self.bias_initializer = initializers.get(bias_initializer)
        self.kernel_regularizer = regularizers.get(kernel_regularizer)
        self.bias_regularizer = regularizers.get(bias_regularizer)

    def build(self, input_shape):
        self.add_weight('kernel', (input_shape[-1], self.units), initializer=self.kernel_initializer,
                        regularizer=self.kernel_regularizer)
        if self.use_bias:
            self.add_weight('bias', (self.units,), initializer=self.bias_initializer,
                            regularizer=self.bias_regularizer)
        else:
            self.bias = None
        self.built = True

    def call(self, inputs):
        rows = inputs.numel() // max(inputs.shape[-1], 1)
        tall = (inputs.is_cuda and inputs.dtype in (torch.bfloat16, torch.float16) and self.kernel.dtype == torch.float32
                and rows >= 8192 and rows % _TallDenseFn.SPLITS == 0 and inputs.is_contiguous())
        if (inputs.is_cuda and inputs.dtype in (torch.bfloat16, torch.float16) and self.kernel.dtype == torch.float32 and rows > 0
                and inputs.is_contiguous() and activations.serialize(self.activation) == 'softmax' and self.units <= 64
                and not L.dbg(L.QK_DBG_NO_FUSED_SOFTMAX)):
            x2 = inputs.reshape(rows, inputs.shape[-1])
            if tall or Fq.dense_softmax_supported(x2, self.units):       # (any row count on the hand-written kernels; the composition needs `tall`)
                y = _DenseSoftmaxFn.apply(x2, self.kernel, self.bias)
                return y.reshape(tuple(inputs.shape[:-1]) + (self.units,))
        if tall:
            out = _TallDenseFn.apply(inputs.reshape(rows, inputs.shape[-1]), self.kernel, self.bias)
            return self.activation(out.reshape(tuple(inputs.shape[:-1]) + (self.units,)))
        out = inputs @ self.kernel.to(inputs.dtype)
        if self.bias is not None:
            out = out + self.bias.to(inputs.dtype)
        return self.activation(out)

    def compute_output_shape(self, input_shape):
        return tuple(input_shape[:-1]) + (self.units,)


def _pool_nd(x, pool, strides, padding, axes, mode):
    """Pool `axes` of x with TensorFlow padding semantics."""
    rank = len(axes)
    if x.dim() == 4 and rank <= 2:
        # Pool a 4-D tensor where it lies: per-axis window / stride over dims 1..3 (1 = untouched); when
        # dim 1 is untouched torch pools dims (2, 3) of the (B, C, H, W) view directly -- with its NHWC
        # kernels when the buffer is channels-last, which is how the quaternion layers keep
        # channels_first tensors.  (The TIMIT model's MaxPooling2D((1, 3)) on a
        # channels_first tensor is this case: Keras' default data_format makes it pool axes (C, F) with a
        # window of 1 on C.)  TF padding must be all on the high side, which ceil_mode covers: a partial
        # last window ignores the missing elements (max) / leaves them out of the divisor (avg).
        win, step = [1, 1, 1], [1, 1, 1]
        for ax, pl, st in zip(axes, pool, strides):
            win[ax - 1], step[ax - 1] = pl, st
        if win[0] == 1 and step[0] == 1:
            lo_hi = [tf_pads(x.shape[d], win[d - 1], step[d - 1], 1, padding) for d in (2, 3)]
            if all(lo == 0 for lo, _ in lo_hi):
                ceil = any(hi > 0 for _, hi in lo_hi)
                if mode == 'max':
                    xl = x.permute(0, 2, 3, 1)                      # (B, H, W, C) view
                    if xl.is_contiguous() and Fq.maxpool2d_supported(xl, win[1:], step[1:]):
                        # channels-last buffer, non-overlapping windows: the engine's own HBM-bound kernels
                        out = [-(-x.shape[d] // step[d - 1]) if ceil else (x.shape[d] - win[d - 1]) // step[d - 1] + 1
                               for d in (2, 3)]
                        return Fq.maxpool2d_channels_last(xl, win[1:], out).permute(0, 3, 1, 2)
                    return F.max_pool2d(x, tuple(win[1:]), tuple(step[1:]), ceil_mode=ceil)
                return F.avg_pool2d(x, tuple(win[1:]), tuple(step[1:]), ceil_mode=ceil, count_include_pad=False)
    perm = [0] + [i for i in range(1, x.dim()) if i not in axes] + list(axes)
    xp = x.permute(perm)
    lead = xp.shape[:x.dim() - rank]
    xp = xp.reshape((-1, 1) + tuple(xp.shape[-rank:]))
    pads = []
    for ax in reversed(range(rank)):
        lo, hi = tf_pads(xp.shape[2 + ax], pool[ax], strides[ax], 1, padding)
        pads += [lo, hi]
    fn = {1: (F.max_pool1d, F.avg_pool1d), 2: (F.max_pool2d, F.avg_pool2d), 3: (F.max_pool3d, F.avg_pool3d)}[rank]
    if mode == 'max':
        if any(pads):
            xp = F.pad(xp, pads, value=float('-inf'))
        y = fn[0](xp, pool, strides)
    else:
        k = float(np.prod(pool))
        if any(pads):
            ones = F.pad(torch.ones_like(xp), pads)
            y = fn[1](F.pad(xp, pads), pool, strides) / fn[1](ones, pool, strides)   # divisor excludes padding
        else:
            y = fn[1](xp, pool, strides)
        del k
    y = y.reshape(tuple(lead) + tuple(y.shape[-rank:]))
    inv = [0] * x.dim()
    for i, p in enumerate(perm):
        inv[p] = i
    return y.permute(inv)


class _Pooling(Layer):
    rank, mode = 1, 'max'

    def __init__(self, pool_size=2, strides=None, padding='valid', data_format=None, **kwargs):
        super(_Pooling, self).__init__(**kwargs)
        self.pool_size = normalize_tuple(pool_size, self.rank, 'pool_size')
        self.strides = normalize_tuple(self.pool_size if strides is None else strides, self.rank, 'strides')
        self.padding = padding
        self.data_format = 'channels_last' if data_format is None else data_format

    def call(self, inputs):
        first = 2 if self.data_format == 'channels_first' else 1
        axes = tuple(range(first, first + self.rank))
        return _pool_nd(inputs, self.pool_size, self.strides, self.padding, axes, self.mode)


class AveragePooling1D(_Pooling):
    rank, mode = 1, 'avg'


class MaxPooling1D(_Pooling):
    rank, mode = 1, 'max'


class MaxPooling2D(_Pooling):
    rank, mode = 2, 'max'


class AveragePooling2D(_Pooling):
    rank, mode = 2, 'avg'


class PReLU(Layer):
    """keras.layers.PReLU: alpha has the input's shape without the batch axis, with size 1 on the shared
    axes.  Keras writes `param_shape[i - 1] = 1` for every i in `shared_axes`, so axis 0 -- which the TIMIT
    model passes, PReLU(shared_axes=[1, 0]), interspeech_model.py:99-101 -- lands on index -1 and shares the
    LAST axis: alpha is (1, F, 1) behind a (B, C, F, T) convolution (one slope per frequency row, shared over
    channels and over the variable-length time axis) and (1, 1) behind a TimeDistributed dense layer.
    An axis whose length is unknown (None) is shared as well -- it could not be sized."""

    def __init__(self, alpha_initializer='zeros', alpha_regularizer=None, alpha_constraint=None,
                 shared_axes=None, **kwargs):
        super(PReLU, self).__init__(**kwargs)
        if shared_axes is None:
            self.shared_axes = None
        elif not isinstance(shared_axes, (list, tuple)):
            self.shared_axes = [shared_axes]
        else:
            self.shared_axes = list(shared_axes)
        self.alpha_initializer = initializers.get(alpha_initializer)
        self.alpha_regularizer = regularizers.get(alpha_regularizer)
        self.alpha_constraint = alpha_constraint

    def build(self, input_shape):
        shape = [1 if d is None else d for d in input_shape[1:]]
        for i in self.shared_axes or []:
            shape[i - 1] = 1                                   # i == 0 -> index -1, as in Keras
        self.add_weight('alpha', tuple(shape), initializer=self.alpha_initializer,
                        regularizer=self.alpha_regularizer, constraint=self.alpha_constraint)
        self.built = True

    def call(self, inputs):
        a = self.alpha.to(inputs.dtype)
        return torch.relu(inputs) - a * torch.relu(-inputs)

    def get_config(self):
        cfg = super(PReLU, self).get_config()
        cfg.update(alpha_initializer=initializers.serialize(self.alpha_initializer),
                   alpha_regularizer=regularizers.serialize(self.alpha_regularizer),
                   shared_axes=self.shared_axes)
        return cfg


class Permute(Layer):
    def __init__(self, dims, **kwargs):
        super(Permute, self).__init__(**kwargs)
        self.dims = tuple(dims)

    def call(self, inputs):
        return inputs.permute((0,) + self.dims)


class TimeDistributed(Layer):
    """Applies `layer` to every step of (B, T, ...): folds T into the batch, as Keras does."""

    def __init__(self, layer, **kwargs):
        super(TimeDistributed, self).__init__(**kwargs)
        self.layer = layer

    def call(self, inputs):
        b, t = inputs.shape[0], inputs.shape[1]
        y = self.layer(inputs.reshape((b * t,) + tuple(inputs.shape[2:])))
        return y.reshape((b, t) + tuple(y.shape[1:]))


class _GradScale(torch.autograd.Function):
    """identity forward, gradient x s backward (static loss scaling on the torch path of ctc_batch_cost)"""

    @staticmethod
    def forward(ctx, x, s):
        ctx.s = s
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        return g * ctx.s, None


def ctc_batch_cost(y_pred, labels, input_length, label_length, blank=None, loss_scale=1.0):
    """K.ctc_batch_cost (interspeech_model.py:37-39): y_pred (B, T, C) softmax outputs, blank = last class; returns
    the per-sample negative log-likelihood (B, 1).  Keras 2.x hands log(y_pred + epsilon()) to tf.nn.ctc_loss as
    LOGITS, and that op normalises them again (softmax), so the log-probabilities are
    log_softmax(log(y_pred + 1e-7)); ctc_merge_repeated=True, no collapse of repeated labels (TF defaults).
    loss_scale: multiplies the gradient sent back (not the cost): float16 training, see functional.ctc_batch_cost."""
    if (blank is None and y_pred.is_cuda and Fq.ctc_supported(y_pred, labels) and not L.dbg(L.QK_DBG_NO_FUSED_CTC)):
        return Fq.ctc_batch_cost(y_pred, labels, input_length, label_length, loss_scale=loss_scale)        # one HIP launch: cost + gradient
    blank = y_pred.shape[-1] - 1 if blank is None else blank
    yp = y_pred.float()
    if loss_scale != 1.0:
        yp = _GradScale.apply(yp, float(loss_scale))          # (in fp32, in front of the cast back to y_pred's dtype)
    logp = torch.log_softmax(torch.log(yp + 1e-7), dim=-1).transpose(0, 1)
    loss = F.ctc_loss(logp, labels.long(), input_length.reshape(-1).long(), label_length.reshape(-1).long(),
                      blank=blank, reduction='none', zero_infinity=False)
    return loss.reshape(-1, 1)
