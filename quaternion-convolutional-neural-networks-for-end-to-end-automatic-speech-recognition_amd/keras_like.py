"""The slice of the Keras-2 `Layer` protocol the reference's layers sit behind, on torch.

The reference's drop-in surface is `keras.layers.Layer`:
`__init__(**kwargs)`, `build(input_shape)` creating weights through `add_weight`,
`call(inputs)`, `compute_output_shape`, `get_config`, and instances used as callables
(`QuaternionConv1D(32, 3, ...)(x)`, models/example_model.py:25).  Here a Layer is a
`torch.nn.Module` whose `forward` runs build-once-then-call, so torch optimizers,
`.to(device)` and `state_dict` work, while the Keras names (`kernel`, `bias`, `r`) and
semantics are kept.  `activations` / `initializers` / `regularizers` / `constraints`
mirror the `keras.*.get / serialize` helpers the layer constructors call
(conv.py:129-147, dense.py:76-85).
"""
import math
import re

import numpy as np
import torch

_uid = {}


class InputSpec(object):
    def __init__(self, dtype=None, shape=None, ndim=None, max_ndim=None, min_ndim=None, axes=None):
        self.dtype, self.shape, self.ndim = dtype, shape, ndim
        self.max_ndim, self.min_ndim, self.axes = max_ndim, min_ndim, axes or {}


# ---- activations ----------------------------------------------------------------------
class activations(object):
    @staticmethod
    def linear(x):
        return x

    @staticmethod
    def relu(x):
        return torch.relu(x)

    @staticmethod
    def tanh(x):
        return torch.tanh(x)

    @staticmethod
    def sigmoid(x):
        return torch.sigmoid(x)

    @staticmethod
    def softmax(x):
        return torch.softmax(x, dim=-1)

    @staticmethod
    def get(identifier):
        if identifier is None:
            return activations.linear
        if isinstance(identifier, str):
            fn = getattr(activations, identifier, None)
            if fn is None or identifier in ('get', 'serialize'):
                raise ValueError('Unknown activation function:' + identifier)
            return fn
        if callable(identifier):
            return identifier
        raise ValueError('Could not interpret activation function identifier:', identifier)

    @staticmethod
    def serialize(fn):
        return fn.__name__


# ---- initializers ---------------------------------------------------------------------
class Initializer(object):
    def __call__(self, shape, dtype=None):
        raise NotImplementedError

    def get_config(self):
        return {}


class Zeros(Initializer):
    def __call__(self, shape, dtype=None):
        return np.zeros(shape)


class Ones(Initializer):
    def __call__(self, shape, dtype=None):
        return np.ones(shape)


class Constant(Initializer):
    def __init__(self, value=0):
        self.value = value

    def __call__(self, shape, dtype=None):
        return np.full(shape, self.value, dtype=np.float64)

    def get_config(self):
        return {'value': self.value}


def _compute_fans(shape, data_format='channels_last'):
    """keras.initializers._compute_fans for kernels stored (*k, in, out) (init.py:55-57)."""
    if len(shape) == 2:
        return shape[0], shape[1]
    receptive_field_size = int(np.prod(shape[:-2]))
    return shape[-2] * receptive_field_size, shape[-1] * receptive_field_size


def _rng(seed):
    """Keras backends draw with `seed = np.random.randint(10e6)` when an initializer has seed=None, i.e. from the GLOBAL
    numpy state: `np.random.seed(...)` makes a whole model reproducible (what data-parallel replicas rely on)."""
    return np.random.RandomState(np.random.randint(int(10e6)) if seed is None else seed)


class RandomNormal(Initializer):
    def __init__(self, mean=0., stddev=0.05, seed=None):
        self.mean, self.stddev, self.seed = mean, stddev, seed

    def __call__(self, shape, dtype=None):
        return _rng(self.seed).normal(self.mean, self.stddev, shape)

    def get_config(self):
        return {'mean': self.mean, 'stddev': self.stddev, 'seed': self.seed}


class RandomUniform(Initializer):
    def __init__(self, minval=-0.05, maxval=0.05, seed=None):
        self.minval, self.maxval, self.seed = minval, maxval, seed

    def __call__(self, shape, dtype=None):
        return _rng(self.seed).uniform(self.minval, self.maxval, shape)

    def get_config(self):
        return {'minval': self.minval, 'maxval': self.maxval, 'seed': self.seed}


class VarianceScaling(Initializer):
    """keras.initializers.VarianceScaling (glorot / he / lecun families) for the real-valued
    weights that sit next to the quaternion layers (e.g. a bias or a softmax Dense)."""

    def __init__(self, scale=1.0, mode='fan_in', distribution='normal', seed=None):
        self.scale, self.mode, self.distribution, self.seed = scale, mode, distribution, seed

    def __call__(self, shape, dtype=None):
        fan_in, fan_out = _compute_fans(shape) if len(shape) >= 2 else (shape[0], shape[0])
        n = {'fan_in': fan_in, 'fan_out': fan_out, 'fan_avg': (fan_in + fan_out) / 2.}[self.mode]
        scale = self.scale / max(1., n)
        rng = _rng(self.seed)
        if self.distribution == 'normal':
            return rng.normal(0., np.sqrt(scale), shape)
        limit = np.sqrt(3. * scale)
        return rng.uniform(-limit, limit, shape)

    def get_config(self):
        return {'scale': self.scale, 'mode': self.mode, 'distribution': self.distribution,
                'seed': self.seed}


def _vs(scale, mode, distribution):
    return lambda seed=None: VarianceScaling(scale, mode, distribution, seed)


class initializers(object):
    Initializer, Zeros, Ones, Constant = Initializer, Zeros, Ones, Constant
    RandomNormal, RandomUniform, VarianceScaling = RandomNormal, RandomUniform, VarianceScaling
    _compute_fans = staticmethod(_compute_fans)
    _by_name = {'zeros': Zeros, 'ones': Ones, 'constant': Constant,
                'Zeros': Zeros, 'Ones': Ones, 'Constant': Constant,
                'random_normal': RandomNormal, 'RandomNormal': RandomNormal, 'normal': RandomNormal,
                'random_uniform': RandomUniform, 'RandomUniform': RandomUniform, 'uniform': RandomUniform,
                'VarianceScaling': VarianceScaling,
                'glorot_uniform': _vs(1., 'fan_avg', 'uniform'), 'glorot_normal': _vs(1., 'fan_avg', 'normal'),
                'he_uniform': _vs(2., 'fan_in', 'uniform'), 'he_normal': _vs(2., 'fan_in', 'normal'),
                'lecun_uniform': _vs(1., 'fan_in', 'uniform'), 'lecun_normal': _vs(1., 'fan_in', 'normal')}

    @staticmethod
    def get(identifier):
        if identifier is None:
            return None
        if isinstance(identifier, dict):
            return initializers._by_name[identifier['class_name']](**identifier.get('config', {}))
        if isinstance(identifier, str):
            if identifier not in initializers._by_name:
                raise ValueError('Unknown initializer: ' + identifier)
            return initializers._by_name[identifier]()
        if callable(identifier):
            return identifier
        raise ValueError('Could not interpret initializer identifier: ' + str(identifier))

    @staticmethod
    def serialize(initializer):
        if initializer is None:
            return None
        if isinstance(initializer, Initializer):
            return {'class_name': initializer.__class__.__name__, 'config': initializer.get_config()}
        return getattr(initializer, '__name__', str(initializer))


# ---- regularizers / constraints ---------------------------------------------------------
class L1L2(object):
    """keras.regularizers.L1L2: loss = l1*sum|w| + l2*sum w^2 (interspeech_model.py:63,68 uses l2)."""

    def __init__(self, l1=0., l2=0.):
        self.l1, self.l2 = float(l1), float(l2)

    def __call__(self, w):
        out = w.new_zeros(())
        if self.l1:
            out = out + self.l1 * w.abs().sum()
        if self.l2:
            out = out + self.l2 * (w * w).sum()
        return out

    def get_config(self):
        return {'l1': self.l1, 'l2': self.l2}


def l1(l=0.01):
    return L1L2(l1=l)


def l2(l=0.01):
    return L1L2(l2=l)


class _Passthrough(object):
    kind = 'object'

    @classmethod
    def get(cls, identifier):
        if identifier is None:
            return None
        if isinstance(identifier, dict) and identifier.get('class_name') == 'L1L2':
            return L1L2(**identifier.get('config', {}))
        if callable(identifier):
            return identifier
        raise ValueError('Could not interpret %s identifier: %r' % (cls.kind, identifier))

    @staticmethod
    def serialize(obj):
        if obj is None:
            return None
        if hasattr(obj, 'get_config'):
            return {'class_name': obj.__class__.__name__, 'config': obj.get_config()}
        return getattr(obj, '__name__', str(obj))


class regularizers(_Passthrough):
    kind = 'regularizer'
    L1L2, l1, l2 = L1L2, staticmethod(l1), staticmethod(l2)


class constraints(_Passthrough):
    kind = 'constraint'


# ---- Layer ------------------------------------------------------------------------------
class Layer(torch.nn.Module):
    _allowed_kwargs = {'input_shape', 'batch_input_shape', 'batch_size', 'dtype', 'name',
                       'trainable', 'weights', 'input_dtype'}

    def __init__(self, **kwargs):
        super(Layer, self).__init__()
        for k in kwargs:
            if k not in self._allowed_kwargs:
                raise TypeError('Keyword argument not understood:', k)
        name = kwargs.get('name')
        if not name:
            prefix = _to_snake_case(self.__class__.__name__)
            _uid[prefix] = _uid.get(prefix, 0) + 1
            name = prefix + '_' + str(_uid[prefix])
        self.name = name
        self.trainable = kwargs.get('trainable', True)
        if 'batch_input_shape' in kwargs:
            self.batch_input_shape = tuple(kwargs['batch_input_shape'])
        elif 'input_shape' in kwargs:
            self.batch_input_shape = (kwargs.get('batch_size'),) + tuple(kwargs['input_shape'])
        self.built = False
        self.input_spec = None
        self.supports_masking = False
        self._weight_names = []
        self._regularizers = {}
        self._constraints = {}
        self._build_device = None

    # Keras: add_weight(name, shape, dtype, initializer, regularizer, trainable, constraint)
    def add_weight(self, name=None, shape=None, dtype=None, initializer=None, regularizer=None,
                   trainable=True, constraint=None):
        if isinstance(name, (tuple, list)):       # legacy positional (shape, ...) form, conv.py:175
            name, shape = shape if isinstance(shape, str) else None, tuple(name)
        init = initializers.get(initializer) if not callable(initializer) else initializer
        value = np.asarray(init(tuple(shape)), dtype=np.float64)
        # Keras<=2.2 semantics: the variable takes the shape of what the initializer returns
        # (conv.py:165-181 requests (*k,Cq,F); qconv_init returns (*k,Cq,4F)).
        p = torch.nn.Parameter(torch.tensor(value, dtype=torch.float32, device=self._build_device),
                               requires_grad=bool(trainable and self.trainable))
        self.register_parameter(name, p)
        self._weight_names.append(name)
        if regularizer is not None:
            self._regularizers[name] = regularizer
            p._qk_regularized = regularizer     # dp.FlatParams: this gradient has a second source (or folds into Adam)
        if constraint is not None:
            self._constraints[name] = constraint
        return p

    @property
    def weights(self):
        """[(keras name, parameter)] in creation order."""
        return [(n, getattr(self, n)) for n in self._weight_names]

    def regularization_losses(self):
        """Keras collects `regularizer(weight)` into model.losses; add these to the loss."""
        return [r(getattr(self, n)) for n, r in self._regularizers.items()]

    @torch.no_grad()
    def apply_constraints(self):
        """Keras applies `constraint(weight)` after every optimizer update."""
        for n, c in self._constraints.items():
            p = getattr(self, n)
            p.copy_(c(p))

    def build(self, input_shape):
        self.built = True

    def call(self, inputs):
        return inputs

    def forward(self, inputs):
        if not self.built:
            self._build_device = inputs.device
            self.build(tuple(inputs.shape))
        return self.call(inputs)

    def compute_output_shape(self, input_shape):
        return input_shape

    def get_config(self):
        config = {'name': self.name, 'trainable': self.trainable}
        if hasattr(self, 'batch_input_shape'):
            config['batch_input_shape'] = self.batch_input_shape
        return config

    @classmethod
    def from_config(cls, config):
        return cls(**config)


def _to_snake_case(name):
    """keras.engine.base_layer._to_snake_case ('QuaternionConv1D' -> 'quaternion_conv1d')."""
    intermediate = re.sub('(.)([A-Z][a-z0-9]+)', r'\1_\2', name)
    insecure = re.sub('([a-z])([A-Z])', r'\1_\2', intermediate).lower()
    return 'private' + insecure if insecure[0] == '_' else insecure


__all__ = ['Layer', 'InputSpec', 'activations', 'initializers', 'regularizers', 'constraints',
           'Initializer', 'math']
