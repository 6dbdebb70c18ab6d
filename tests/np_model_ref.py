"""TEST INFRASTRUCTURE -- float64 numpy composition of the TIMIT quaternion CNN around the CPU oracle.

What models/interspeech_model.py:getTimitModel2D (:45-185) of the reference builds, restated layer by layer:
the quaternion layers are the oracle's (oracle/oracle.py: conv.py:288-345, dense.py:126-164 and their
autodiff), the stock Keras layers around them are a few lines of numpy each (the third-party semantics
are cited at each function).  Forward AND backward, so a GPU model can be compared gradient by gradient.
Only tests import this.
"""
import numpy as np

from oracle import oracle


def _ident(a):
    return a


def maxpool_freq_same(x, win=3):
    """keras MaxPooling2D((1, win), padding='same') with the default channels_last data_format applied to a
    channels_first (B, C, F, T) tensor (interspeech_model.py:103): pools axis 2.  TF 'same': out = ceil(F/win),
    total padding out*win - F with the LOW side getting total // 2 (F = 41: 0 low, 1 high; F = 40: 1 low, 1 high),
    padded cells never win."""
    b, c, f, t = x.shape
    out = -(-f // win)
    lo = (out * win - f) // 2
    y = np.empty((b, c, out, t))
    arg = np.empty((b, c, out, t), dtype=np.int64)
    for o in range(out):
        a, e = max(o * win - lo, 0), min((o + 1) * win - lo, f)
        seg = x[:, :, a:e, :]
        arg[:, :, o, :] = seg.argmax(2) + a                # first maximum wins (TF / torch)
        y[:, :, o, :] = seg.max(2)
    return y, arg


def maxpool_freq_same_bwd(dy, arg, f):
    b, c, out, t = dy.shape
    dx = np.zeros((b, c, f, t))
    bi, ci, oi, ti = np.meshgrid(np.arange(b), np.arange(c), np.arange(out), np.arange(t), indexing='ij')
    np.add.at(dx, (bi, ci, arg, ti), dy)
    return dx


def prelu(x, alpha):
    """keras.layers.PReLU: relu(x) - alpha * relu(-x), alpha broadcast over its size-1 (shared) axes."""
    return np.maximum(x, 0) - alpha[None] * np.maximum(-x, 0)


def prelu_bwd(x, alpha, dy):
    dx = dy * np.where(x > 0, 1.0, 0.0) + dy * alpha[None] * np.where(x < 0, 1.0, 0.0)
    full = dy * np.minimum(x, 0)                           # d/d alpha of -alpha * relu(-x) = min(x, 0)
    axes = tuple(i + 1 for i, n in enumerate(alpha.shape) if n == 1)
    dalpha = full.sum(axis=(0,) + axes, keepdims=True)[0]
    return dx, dalpha.reshape(alpha.shape)


def keras_prelu_alpha_shape(input_shape, shared_axes):
    """param_shape of keras.layers.PReLU.build: input_shape[1:] with `param_shape[i - 1] = 1` for every shared axis
    (so axis 0 lands on index -1)."""
    shape = [1 if d is None else d for d in input_shape[1:]]
    for i in shared_axes:
        shape[i - 1] = 1
    return tuple(shape)


class TimitRef(object):
    """Parameters pulled from a qcnn_amd.models.TimitQCNN instance (as float64 numpy); forward/backward in float64.

    `rnd` emulates 16-bit storage of activations (identity for fp32 runs); `rnd_w` the rounding of the kernels
    the 16-bit matrix-core path applies (identity for fp32 runs)."""

    def __init__(self, model, act='relu', rnd=_ident, rnd_w=_ident):
        g = lambda t: t.detach().cpu().double().numpy()
        self.act = act if model.prelu is None else None
        self.rnd, self.rnd_w = rnd, rnd_w
        self.conv = (g(model.conv.kernel), g(model.conv.bias))
        self.convs = [(g(c.kernel), g(c.bias)) for c in model.convs]
        self.dense = [(g(d.layer.r), g(d.layer.bias)) for d in model.dense]
        self.pred = (g(model.pred.layer.kernel), g(model.pred.layer.bias))
        self.alphas = [g(p.alpha) for p in model.prelu] if model.prelu is not None else None

    # ---- forward ----------------------------------------------------------------------------
    def forward(self, x, forced=None, keeps=None):
        """forced: {'pool': (B, C, F', T), 'y_c<i>': (B, C, F', T), 'y_d<i>': (B*T, units)} -- outputs of the GPU model.
        Where given, the layer's own output is checked against it (`forced_tol`, relative to its maximum) and then
        REPLACED by it, so that everything downstream -- the next layer's input, the relu masks and the operands of the
        backward -- is what the GPU had (16-bit runs: the gradients can then be compared element by element).
        keeps: {'c<i>' / 'd<i>': dropout factor (0 or 1 / (1 - rate)) per element of that layer's output} -- the
        Dropout layers of interspeech_model.py:117-121,131-137,150-154 with the masks the fused kernels generate."""
        rnd, rw = self.rnd, self.rnd_w
        self.saved = s = {}
        forced = forced or {}
        keeps = keeps or {}
        self.forced_err = {}
        kw = dict(padding='same', data_format='channels_first', activation=self.act)
        k = 0

        def activate(pre, k):
            if self.alphas is None:
                return pre
            s['pre%d' % k] = pre
            return rnd(prelu(pre, self.alphas[k]))

        def force(name, own):
            if name not in forced:
                return own
            f = np.asarray(forced[name], dtype=np.float64)
            assert f.shape == own.shape, (name, f.shape, own.shape)
            self.forced_err[name] = float(np.abs(f - own).max()) / max(float(np.abs(own).max()), 1e-30)
            return f

        def drop(name, h):
            if name not in keeps:
                return h
            s['keep_' + name] = keeps[name]
            return rnd(h * keeps[name])

        s['x'] = x
        h = rnd(oracle.forward(x, rw(self.conv[0]), self.conv[1], 2, **kw))
        s['y_conv'] = h
        h = activate(h, k); k += 1
        s['pool_in_f'] = h.shape[2]
        h, s['pool_arg'] = maxpool_freq_same(h, 3)
        h = force('pool', h)
        for i, (w, b) in enumerate(self.convs):
            s['x_c%d' % i] = h
            h = rnd(oracle.forward(h, rw(w), b, 2, **kw))
            h = activate(h, k); k += 1
            h = force('y_c%d' % i, drop('c%d' % i, h))
            s['y_c%d' % i] = h                            # relu runs: y > 0 <=> pre > 0 (and kept): the mask
        bsz, c, f, t = h.shape
        s['perm_shape'] = h.shape
        h = h.transpose(0, 3, 1, 2).reshape(bsz * t, c * f)        # Permute((3,1,2)) + reshape, TimeDistributed
        for i, (w, b) in enumerate(self.dense):
            s['x_d%d' % i] = h
            y = rnd(oracle.forward(h, rw(w), b, 0, activation=self.act))
            if self.alphas is not None:
                # TimeDistributed output is (B, T, units): alpha (1, 1) broadcasts over everything
                s['pre%d' % k] = y
                y = rnd(prelu(y.reshape(bsz, t, -1), self.alphas[k]).reshape(bsz * t, -1))
            y = force('y_d%d' % i, drop('d%d' % i, y))
            s['y_d%d' % i] = y
            k += 1
            h = y
        s['x_pred'] = h
        z = rnd(rnd(h @ rw(self.pred[0])) + rw(self.pred[1]))
        e = np.exp(z - z.max(1, keepdims=True))
        p = rnd(e / e.sum(1, keepdims=True))
        s['p'] = p
        return p.reshape(bsz, t, -1)

    # ---- backward of sum(pred * dpred) --------------------------------------------------------
    def backward(self, dpred):
        s = self.saved
        grads = {}
        bsz, c, f, t = s['perm_shape']
        p = s['p']
        dp = dpred.reshape(p.shape)
        dz = p * (dp - (dp * p).sum(1, keepdims=True))
        grads['pred.kernel'] = s['x_pred'].T @ dz
        grads['pred.bias'] = dz.sum(0)
        dh = dz @ self.rnd_w(self.pred[0]).T
        n_act = 1 + len(self.convs) + len(self.dense)
        k = n_act - 1
        for i in reversed(range(len(self.dense))):
            w, b = self.dense[i]
            if 'keep_d%d' % i in s:
                dh = dh * s['keep_d%d' % i]
            if self.alphas is not None:
                d3, da = prelu_bwd(s['pre%d' % k].reshape(bsz, t, -1), self.alphas[k], dh.reshape(bsz, t, -1))
                grads['alpha%d' % k] = da
                dh = d3.reshape(bsz * t, -1)
            k -= 1
            dh, dw, db = oracle.backward(s['x_d%d' % i], self.rnd_w(w), b, dh, 0, y=s['y_d%d' % i], activation=self.act)
            grads['dense%d.r' % i], grads['dense%d.bias' % i] = dw, db
        dh = dh.reshape(bsz, t, c, f).transpose(0, 2, 3, 1)
        kw = dict(padding='same', data_format='channels_first', activation=self.act)
        for i in reversed(range(len(self.convs))):
            w, b = self.convs[i]
            if 'keep_c%d' % i in s:
                dh = dh * s['keep_c%d' % i]
            if self.alphas is not None:
                dh, grads['alpha%d' % k] = prelu_bwd(s['pre%d' % k], self.alphas[k], dh)
            k -= 1
            dh, dw, db = oracle.backward(s['x_c%d' % i], self.rnd_w(w), b, dh, 2, y=s['y_c%d' % i], **kw)
            grads['conv%d.kernel' % i], grads['conv%d.bias' % i] = dw, db
        dh = maxpool_freq_same_bwd(dh, s['pool_arg'], s['pool_in_f'])
        if self.alphas is not None:
            dh, grads['alpha0'] = prelu_bwd(s['pre0'], self.alphas[0], dh)
        dx, dw, db = oracle.backward(s['x'], self.rnd_w(self.conv[0]), self.conv[1], dh, 2, y=s['y_conv'], **kw)
        grads['conv.kernel'], grads['conv.bias'], grads['x'] = dw, db, dx
        return grads


def model_grads(model):
    """The same dictionary keys from a TimitQCNN after backward()."""
    g = lambda t: None if t.grad is None else t.grad.detach().cpu().double().numpy()
    out = {'conv.kernel': g(model.conv.kernel), 'conv.bias': g(model.conv.bias),
           'pred.kernel': g(model.pred.layer.kernel), 'pred.bias': g(model.pred.layer.bias)}
    for i, c in enumerate(model.convs):
        out['conv%d.kernel' % i], out['conv%d.bias' % i] = g(c.kernel), g(c.bias)
    for i, d in enumerate(model.dense):
        out['dense%d.r' % i], out['dense%d.bias' % i] = g(d.layer.r), g(d.layer.bias)
    if model.prelu is not None:
        for i, pl in enumerate(model.prelu):
            out['alpha%d' % i] = g(pl.alpha)
    return out


# ---- fixtures captured from the reference's own model builders (oracle/make_golden.py: g17_*, g13_*) -----------
def load_timit_fixture(path):
    """tests/golden/g17_timit_*.npz -> dict(x, dpred, pred, ctc_cost, labels, input_length, label_length, d,
    weights {name: array}, grads {name: array}) with this module's names (conv.kernel, conv0.bias, dense1.r,
    alpha3, pred.kernel, ...).  The fixture lists the variables in the order getTimitModel2D created them."""
    import json
    z = np.load(path)
    cfg = json.loads(str(z['config']))
    names, n_conv, n_dense, n_alpha = [], 0, 0, 0
    for layer, wname, _ in cfg['weights']:
        if wname == 'alpha':
            names.append('alpha%d' % n_alpha)
            n_alpha += 1
        elif layer == 'conv':
            names.append('conv.' + wname)
        elif layer.startswith('conv'):
            names.append('conv%d.%s' % (n_conv, wname))
            n_conv += wname == 'bias'
        elif wname == 'r' or (layer.startswith('quaterniondense') and wname == 'bias'):
            names.append('dense%d.%s' % (n_dense, wname))
            n_dense += wname == 'bias'
        else:
            names.append('pred.' + wname)
    out = {k: z[k] for k in ('x', 'dpred', 'pred', 'ctc_cost', 'labels', 'input_length', 'label_length')}
    out['d'] = cfg['d']
    out['weights'] = {n: z['w%03d' % i].astype(np.float64) for i, n in enumerate(names)}
    out['grads'] = {n: z['g%03d' % i] for i, n in enumerate(names)}
    out['grads']['x'] = z['gx']
    return out


class _NS(object):
    def __init__(self, **kw):
        self.__dict__.update(kw)


def fake_model_from_weights(w):
    """A TimitQCNN-shaped attribute bag over numpy weights (for TimitRef)."""
    import torch
    t = lambda a: torch.tensor(a, dtype=torch.float64)
    n_conv = len([k for k in w if k.startswith('conv') and k.endswith('.kernel') and k != 'conv.kernel'])
    alphas = sorted((k for k in w if k.startswith('alpha')), key=lambda s: int(s[5:]))
    return _NS(conv=_NS(kernel=t(w['conv.kernel']), bias=t(w['conv.bias'])),
               convs=[_NS(kernel=t(w['conv%d.kernel' % i]), bias=t(w['conv%d.bias' % i])) for i in range(n_conv)],
               dense=[_NS(layer=_NS(r=t(w['dense%d.r' % i]), bias=t(w['dense%d.bias' % i]))) for i in range(3)],
               pred=_NS(layer=_NS(kernel=t(w['pred.kernel']), bias=t(w['pred.bias']))),
               prelu=[_NS(alpha=t(w[k])) for k in alphas] if alphas else None)


def load_weights_into_model(model, w):
    """Copy fixture weights into a built qcnn_amd.models.TimitQCNN."""
    import torch
    with torch.no_grad():
        def put(p, a):
            assert tuple(p.shape) == tuple(a.shape), (tuple(p.shape), a.shape)
            p.copy_(torch.tensor(a, dtype=torch.float32))
        put(model.conv.kernel, w['conv.kernel']); put(model.conv.bias, w['conv.bias'])
        for i, c in enumerate(model.convs):
            put(c.kernel, w['conv%d.kernel' % i]); put(c.bias, w['conv%d.bias' % i])
        for i, d in enumerate(model.dense):
            put(d.layer.r, w['dense%d.r' % i]); put(d.layer.bias, w['dense%d.bias' % i])
        put(model.pred.layer.kernel, w['pred.kernel']); put(model.pred.layer.bias, w['pred.bias'])
        if model.prelu is not None:
            for i, pl in enumerate(model.prelu):
                put(pl.alpha, w['alpha%d' % i])
