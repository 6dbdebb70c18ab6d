#!/usr/bin/env python
"""TEST INFRASTRUCTURE -- build-container only (needs /root/reference).

Generates tests/golden/*.npz by EXECUTING THE REFERENCE'S OWN LAYER CODE
(/root/reference/complexnn/{conv,dense,init}.py, imported where it lies) through the
torch-float64 keras stand-in of oracle/keras_standin.py.  torch autograd differentiates
through the reference's slice / negate / concatenate graph, so y, dx, dkernel, dbias are
what Keras/TF autodiff of conv.py:288-345 / dense.py:126-164 yields.

Run:  python oracle/make_golden.py          (rewrites tests/golden/)

Every fixture is DATA: float32-representable inputs (x, kernel, bias, dy) and float64
expected outputs (y, dx, dkernel, dbias) plus a JSON `config` string describing the layer.
"""
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import keras_standin  # noqa: E402

OUT = os.path.join(os.path.dirname(HERE), 'tests', 'golden')


def f32(t):
    """Round a float64 tensor to float32-representable values (kept in float64)."""
    return t.detach().to(torch.float32).to(torch.float64)


def run_layer(cn, kind, ctor_kwargs, x_shape, seed, use_bias=True):
    np.random.seed(seed)
    torch.manual_seed(seed)
    cls = getattr(cn, kind)
    layer = cls(use_bias=use_bias, **ctor_kwargs)
    x = f32(torch.randn(*x_shape, dtype=torch.float64)).requires_grad_(True)
    layer.build(tuple(x.shape))
    with torch.no_grad():
        layer.kernel.copy_(f32(layer.kernel))
        if use_bias:
            layer.bias.copy_(f32(0.1 * torch.randn(layer.bias.shape, dtype=torch.float64)))
    y = layer(x)
    dy = f32(torch.randn(*y.shape, dtype=torch.float64))
    (y * dy).sum().backward()
    rec = {
        'x': x.detach().numpy().astype(np.float32),
        'kernel': layer.kernel.detach().numpy().astype(np.float32),
        'dy': dy.numpy().astype(np.float32),
        'y': y.detach().numpy(),
        'dx': x.grad.numpy(),
        'dkernel': layer.kernel.grad.numpy(),
    }
    if use_bias:
        rec['bias'] = layer.bias.detach().numpy().astype(np.float32)
        rec['dbias'] = layer.bias.grad.numpy()
    cfg = dict(ctor_kwargs)
    cfg.update(kind=kind, use_bias=use_bias, seed=seed, x_shape=list(x_shape),
               kernel_name=layer.kernel.keras_name,
               output_shape=list(layer.compute_output_shape(tuple(x_shape))))
    rec['config'] = np.array(json.dumps(cfg))
    return rec


CASES = [
    # name, kind, ctor kwargs, x shape, seed, use_bias
    ('g01_conv1d_same_relu', 'QuaternionConv1D',
     dict(filters=6, kernel_size=3, padding='same', activation='relu'), (4, 20, 20), 101, True),
    ('g02_conv1d_valid_stride2', 'QuaternionConv1D',
     dict(filters=5, kernel_size=4, strides=2, padding='valid'), (3, 23, 12), 102, True),
    ('g02b_conv1d_same_stride2_even', 'QuaternionConv1D',
     dict(filters=3, kernel_size=4, strides=2, padding='same', activation='relu'),
     (2, 17, 8), 1021, True),
    ('g03_conv1d_valid_dil2', 'QuaternionConv1D',
     dict(filters=4, kernel_size=3, dilation_rate=2, padding='valid', activation='relu'),
     (2, 19, 16), 103, True),
    ('g04_conv1d_causal', 'QuaternionConv1D',
     dict(filters=4, kernel_size=3, dilation_rate=2, padding='causal'), (2, 15, 8), 104, True),
    ('g05_conv1d_chfirst', 'QuaternionConv1D',
     dict(filters=4, kernel_size=3, padding='same', data_format='channels_first',
          activation='relu'), (3, 12, 17), 105, True),
    ('g06_conv2d_chfirst_same', 'QuaternionConv2D',
     dict(filters=4, kernel_size=(3, 5), padding='same', data_format='channels_first',
          activation='relu'), (2, 12, 9, 11), 106, True),
    ('g07_conv2d_chlast_valid', 'QuaternionConv2D',
     dict(filters=3, kernel_size=(2, 3), strides=(2, 1), padding='valid'),
     (2, 8, 9, 8), 107, True),
    ('g07b_conv2d_chlast_same_dil', 'QuaternionConv2D',
     dict(filters=2, kernel_size=(3, 3), dilation_rate=(2, 1), padding='same',
          activation='relu'), (2, 7, 6, 12), 1071, False),
    ('g08_conv2d_first_layer', 'QuaternionConv2D',
     dict(filters=8, kernel_size=(3, 5), padding='same', data_format='channels_first',
          activation='relu'), (2, 4, 41, 13), 108, True),
    ('g09_conv3d_tiny', 'QuaternionConv3D',
     dict(filters=2, kernel_size=(3, 3, 3), padding='valid'), (1, 5, 5, 5, 8), 109, True),
    ('g09b_conv3d_chfirst_same', 'QuaternionConv3D',
     dict(filters=2, kernel_size=(2, 3, 1), padding='same', data_format='channels_first',
          activation='relu'), (2, 8, 4, 5, 3), 1091, True),
    ('g10_dense_relu', 'QuaternionDense',
     dict(units=12, activation='relu'), (5, 28), 110, True),
    ('g11_dense_nobias_linear', 'QuaternionDense',
     dict(units=16), (7, 8), 111, False),
    # MFMA-tile-sized cases (multiple of 8/16 channels, >1 tile of rows)
    ('g14_conv1d_mfma_tile', 'QuaternionConv1D',
     dict(filters=16, kernel_size=3, padding='same', activation='relu'), (3, 50, 32), 114, True),
    ('g15_conv2d_mfma_tile', 'QuaternionConv2D',
     dict(filters=16, kernel_size=(3, 5), padding='same', data_format='channels_first',
          activation='relu'), (2, 64, 6, 21), 115, True),
    ('g16_dense_mfma_tile', 'QuaternionDense',
     dict(units=64, activation='relu'), (70, 96), 116, True),
]


def init_goldens(cn):
    rec = {}
    meta = []
    specs = [
        ('conv1d_he', 'qconv', dict(kernel_size=(3,), input_dim=5, weight_dim=1, nb_filters=6,
                                    criterion='he')),
        ('conv2d_glorot', 'qconv', dict(kernel_size=(3, 5), input_dim=3, weight_dim=2,
                                        nb_filters=4, criterion='glorot')),
        ('conv3d_he', 'qconv', dict(kernel_size=(2, 3, 3), input_dim=2, weight_dim=3,
                                    nb_filters=3, criterion='he')),
        ('conv2d_he_seed7', 'qconv', dict(kernel_size=(3, 5), input_dim=2, weight_dim=2,
                                          nb_filters=4, criterion='he', seed=7)),
        ('dense_he', 'qdense', dict(shape=(7, 3), criterion='he')),
        ('dense_glorot', 'qdense', dict(shape=(16, 8), criterion='glorot')),
    ]
    for name, kind, kw in specs:
        np.random.seed(1234)
        if kind == 'qconv':
            w = cn.qconv_init(**kw)(None)
        else:
            w = cn.qdense_init(**kw)(None)
        rec[name] = np.asarray(w, dtype=np.float64)
        kw2 = {k: (list(v) if isinstance(v, tuple) else v) for k, v in kw.items()}
        meta.append(dict(name=name, kind=kind, kwargs=kw2, np_seed=1234))
    rec['config'] = np.array(json.dumps(meta))
    return rec


def api_goldens(cn):
    """Shapes / names / config keys of the reference classes (SURVEY.md 8a a2,a4,a5,a8,a10)."""
    out = {}
    np.random.seed(5)
    d = cn.QuaternionDense(12, activation='relu', seed=3)
    d.build((None, 28))
    cfg = d.get_config()
    out['dense_config_keys'] = sorted(cfg.keys())
    out['dense_config_scalars'] = {k: cfg[k] for k in
                                   ('units', 'activation', 'use_bias', 'init_criterion', 'seed')}
    out['dense_weight_names'] = [w.keras_name for w in d.weights]
    out['dense_weight_shapes'] = [list(w.shape) for w in d.weights]
    out['dense_output_shape'] = list(d.compute_output_shape((None, 28)))
    c = cn.QuaternionConv2D(4, (3, 5), padding='same', data_format='channels_first')
    c.build((None, 12, 41, None))
    out['conv2d_weight_names'] = [w.keras_name for w in c.weights]
    out['conv2d_weight_shapes'] = [list(w.shape) for w in c.weights]
    out['conv2d_kernel_shape_attr'] = list(c.kernel_shape)
    out['conv2d_output_shape'] = list(c.compute_output_shape((None, 12, 41, None)))
    try:
        c.get_config()
        out['conv2d_get_config_error'] = None
    except Exception as e:  # the reference raises NameError here (conv.py:809)
        out['conv2d_get_config_error'] = type(e).__name__
    shapes = []
    for kw, ishape in [
        (dict(filters=3, kernel_size=4, strides=2, padding='same'), (None, 17, 8)),
        (dict(filters=3, kernel_size=3, dilation_rate=2, padding='valid'), (None, 19, 8)),
        (dict(filters=3, kernel_size=3, dilation_rate=2, padding='causal'), (None, 15, 8)),
        (dict(filters=3, kernel_size=3, padding='valid', data_format='channels_first'),
         (None, 8, 15)),
    ]:
        l1 = cn.QuaternionConv1D(**kw)
        shapes.append(dict(kwargs=kw, input_shape=list(ishape),
                           output_shape=list(l1.compute_output_shape(ishape))))
    out['conv1d_output_shapes'] = shapes
    errs = {}
    try:
        cn.QuaternionConv1D(3, 3).build((None, 10, None))
    except Exception as e:
        errs['conv_none_channel'] = type(e).__name__
    try:
        cn.QuaternionConv1D(3, 3, kernel_initializer='glorot_uniform').build((None, 10, 8))
    except Exception as e:
        errs['conv_bad_initializer'] = type(e).__name__
    try:
        cn.QuaternionDense(8).build((None, 3, 8))
    except Exception as e:
        errs['dense_rank3'] = type(e).__name__
    try:
        cn.QuaternionDense(8, init_criterion='foo').build((None, 8))
    except Exception as e:
        errs['dense_bad_criterion'] = type(e).__name__
    try:
        cn.qconv_init(kernel_size=(3,), input_dim=2, weight_dim=2, nb_filters=2)
    except Exception as e:
        errs['qconv_init_dim_mismatch'] = type(e).__name__
    out['errors'] = errs
    # component getters (complexnn/utils.py:17-115): which slice of an arange tensor each one returns, per rank
    import torch
    getters = []
    for shape in [(3, 8), (2, 5, 8), (2, 8, 3, 5), (2, 8, 3, 2, 2)]:
        x = torch.arange(float(np.prod(shape)), dtype=torch.float64).reshape(shape)
        for part in 'rijk':
            for y in (getattr(cn.utils, 'get_%spart_first' % part)(x), getattr(cn.utils, 'Get%sFirst' % part.upper())().call(x)):
                getters.append(dict(input_shape=list(shape), part=part, output_shape=list(y.shape),
                                    values=[float(v) for v in y.reshape(-1).tolist()]))
        getters.append(dict(input_shape=list(shape), part='shape',
                            output_shape=list(cn.utils.getpart_quaternion_output_shape_first((None,) + shape[1:]))))
    out['getters'] = getters
    return out


def _capture(weights):
    """[(layer name, weight name, tensor)] -> arrays w000.. (float32-representable values) + a JSON manifest."""
    rec, names = {}, []
    for i, t in enumerate(weights):
        rec['w%03d' % i] = t.detach().numpy().astype(np.float32)
        names.append([t.keras_layer, t.keras_name, list(t.shape)])
    return rec, names


def _randomise_small_weights(rng):
    """Post-build hook: biases and PReLU slopes start at zero in the reference; give them float32-representable
    non-zero values so that the fixtures exercise their paths; round every kernel to float32."""
    def hook(layer):
        with torch.no_grad():
            for t in layer.weights:
                if t.keras_name == 'bias':
                    t.copy_(f32(torch.tensor(0.1 * rng.randn(*t.shape))))
                elif t.keras_name == 'alpha':
                    t.copy_(f32(torch.tensor(0.05 + 0.3 * rng.rand(*t.shape))))
                else:
                    t.copy_(f32(t))
    return hook


def example_net_goldens():
    """G13: the reference's OWN model builders (models/example_model.py:15-81, imported where they
    lie) on the first 8 documents of the bundled DECODA DEV set, parsed by the reference's OWN reader
    (working_example.py:19-66): class posteriors + the gradient of sum(p * dp) w.r.t. every weight."""
    import ast
    import types
    import models.example_model as em
    # working_example.py runs its whole training script at import time (and needs the missing TRAIN file), so
    # only its reader function is taken -- compiled from the file where it lies, nothing is copied
    src = open('/root/reference/working_example.py').read()
    fn = [n for n in ast.parse(src).body if isinstance(n, ast.FunctionDef) and n.name == 'dataPrepDecodaQuaternion']
    ns = {'np': np}
    exec(compile(ast.Module(body=fn, type_ignores=[]), '/root/reference/working_example.py', 'exec'), ns)
    x_all, y_all = ns['dataPrepDecodaQuaternion']('/root/reference/decoda/250_DEV_Q.data', isquat=True)
    x = torch.tensor(x_all[:8], dtype=torch.float64)
    out = {'x': x_all[:8].astype(np.float32), 'labels': y_all[:8].astype(np.float32),
           'dev_shape': np.array(x_all.shape), 'dev_label_counts': y_all.sum(0)}
    manifest = {}
    for tag, builder, params in (('qcnn', em.CNN, 'QCNN'), ('qdnn', em.DNN, 'QDNN')):
        np.random.seed(1300 + len(manifest))
        rng = np.random.RandomState(77 + len(manifest))
        keras_standin.ALL_WEIGHTS[:] = []
        keras_standin.WEIGHT_HOOK[0] = _randomise_small_weights(rng)
        keras_standin.feed_inputs([f32(x)])
        model = builder(types.SimpleNamespace(model=params))
        keras_standin.WEIGHT_HOOK[0] = None
        p = model.outputs
        dp = f32(torch.tensor(rng.randn(*p.shape)))
        weights = list(keras_standin.ALL_WEIGHTS)
        grads = torch.autograd.grad((p * dp).sum(), weights)
        wrec, names = _capture(weights)
        for k, v in wrec.items():
            out['%s_%s' % (tag, k)] = v
        for i, gt in enumerate(grads):
            g = gt.numpy()
            out['%s_g%03d' % (tag, i)] = g.astype(np.float32) if g.size > 100000 else g    # (keeps the file small)
        out[tag + '_p'] = p.detach().numpy()
        out[tag + '_dp'] = dp.numpy().astype(np.float32)
        manifest[tag] = names
    out['config'] = np.array(json.dumps(manifest))
    return out


def timit_goldens(aact):
    """G17: the reference's OWN getTimitModel2D (models/interspeech_model.py:45-185, imported where it lies; it is
    Python-2 code: `xrange` and `n/2` are served by an int-casting xrange) run eagerly on a small batch:
    posteriors, CTC cost, and the gradient of sum(pred * dpred) w.r.t. the input and every weight."""
    import builtins
    import types
    builtins.xrange = lambda a, b=None: range(int(a)) if b is None else range(int(a), int(b))
    import models.interspeech_model as im
    seed = 1700 + (1 if aact == 'prelu' else 0)
    np.random.seed(seed)
    rng = np.random.RandomState(seed + 50)
    bsz, t = 2, 10
    x = f32(torch.tensor(rng.randn(bsz, 4, 41, t))).requires_grad_(True)
    labels = torch.tensor(rng.randint(0, 61, (bsz, 4)).astype(np.float32))
    input_length = torch.tensor([[t], [t - 2]], dtype=torch.int64)
    label_length = torch.tensor([[4], [3]], dtype=torch.int64)
    d = types.SimpleNamespace(num_layers=4, start_filter=4, act='relu', aact=aact, dropout=0.0, l2=1e-4,
                              model='quaternion', quat_init='quaternion')
    keras_standin.ALL_WEIGHTS[:] = []
    keras_standin.WEIGHT_HOOK[0] = _randomise_small_weights(rng)
    keras_standin.feed_inputs([x, labels, input_length, label_length])     # order of the Input() calls (:81-90)
    model, val_function = im.getTimitModel2D(d)
    keras_standin.WEIGHT_HOOK[0] = None
    pred = val_function([x])[0]
    cost = model.outputs
    dpred = f32(torch.tensor(rng.randn(*pred.shape)))
    weights = list(keras_standin.ALL_WEIGHTS)
    grads = torch.autograd.grad((pred * dpred).sum(), [x] + weights)
    out, names = _capture(weights)
    out.update(x=x.detach().numpy().astype(np.float32), labels=labels.numpy(), input_length=input_length.numpy(),
               label_length=label_length.numpy(), pred=pred.detach().numpy(), dpred=dpred.numpy().astype(np.float32),
               ctc_cost=cost.detach().numpy(), gx=grads[0].numpy())
    for i, gt in enumerate(grads[1:]):
        out['g%03d' % i] = gt.numpy()
    out['config'] = np.array(json.dumps(dict(weights=names, d=dict(num_layers=4, start_filter=4, act='relu', aact=aact,
                                                                  dropout=0.0, l2=1e-4), seed=seed)))
    return out


def main():
    cn = keras_standin.import_reference()
    os.makedirs(OUT, exist_ok=True)
    for name, kind, kw, xs, seed, ub in CASES:
        rec = run_layer(cn, kind, kw, xs, seed, ub)
        np.savez_compressed(os.path.join(OUT, name + '.npz'), **rec)
        print('%-34s y%s' % (name, tuple(rec['y'].shape)))
    np.savez_compressed(os.path.join(OUT, 'g12_init.npz'), **init_goldens(cn))
    with open(os.path.join(OUT, 'g00_api.json'), 'w') as f:
        json.dump(api_goldens(cn), f, indent=1, sort_keys=True)
    np.savez_compressed(os.path.join(OUT, 'g13_example_nets.npz'), **example_net_goldens())
    for aact in ('none', 'prelu'):
        np.savez_compressed(os.path.join(OUT, 'g17_timit_%s.npz' % ('relu' if aact == 'none' else 'prelu')),
                            **timit_goldens(aact))
    print('wrote', OUT)


if __name__ == '__main__':
    main()
