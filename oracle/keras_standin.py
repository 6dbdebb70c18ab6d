"""TEST INFRASTRUCTURE -- build-container only.  Never imported by the product path.

A minimal ``keras`` namespace (Keras 2.2.x surface) backed by torch-CPU float64, just
large enough that ``import complexnn`` from /root/reference succeeds and the reference's
OWN ``build()`` / ``call()`` / initialiser code executes.  The
reference's slicing / sign / concat code (complexnn/conv.py:294-331,
complexnn/dense.py:131-143) then runs unchanged and torch autograd differentiates
through it, which is how tests/golden/*.npz were produced (oracle/make_golden.py).

What is restated here (third-party arithmetic that is NOT under /root/reference, i.e.
keras + tensorflow, unpinned in setup.py:15-16):
  K.conv1d/2d/3d  cross-correlation, kernel layout (*k, in, out), TF 'same'/'valid'/
                  'causal' padding, channels_first/last
  K.dot, K.concatenate, K.bias_add, activations relu/linear/tanh/sigmoid/softmax
  Layer.add_weight with Keras<=2.2 semantics: the variable takes the shape of the
                  array the initializer RETURNS (needed by conv.py:165-181 where the
                  requested shape is (*k,Cq,F) and qconv_init returns (*k,Cq,4F)).

Round 2: the stock layers the reference's MODEL BUILDERS put around the quaternion layers
(models/example_model.py, models/interspeech_model.py) are restated too -- Input (hands out fed tensors: the
builders then run eagerly on concrete data), Dense, Flatten, Dropout (identity: inference / rate 0),
AveragePooling1D / MaxPooling2D with TensorFlow 'same' semantics, PReLU(shared_axes) exactly as
keras.layers.PReLU.build writes it, Permute, Lambda, TimeDistributed, Model, K.function, K.reshape,
K.ctc_batch_cost, regularizers.l2 -- so that getTimitModel2D / CNN / DNN execute as written and torch
autograd differentiates through them (fixtures g13_*, g17_*).

This file cannot travel to the GPU box in any useful way (the reference does not exist
there); it is committed so the goldens are reproducible here.
"""
import sys
import types

import numpy as np
import torch
import torch.nn.functional as F

DTYPE = torch.float64


# --------------------------------------------------------------------------------------
# keras.utils.conv_utils
# --------------------------------------------------------------------------------------
def normalize_tuple(value, n, name):
    if isinstance(value, int):
        return (value,) * n
    value_tuple = tuple(value)
    if len(value_tuple) != n:
        raise ValueError('The `%s` argument must be a tuple of %d integers. Received: %s'
                         % (name, n, value))
    for v in value_tuple:
        int(v)
    return value_tuple


def normalize_padding(value):
    padding = value.lower()
    if padding not in {'valid', 'same', 'causal'}:
        raise ValueError('The `padding` argument must be one of "valid", "same" '
                         '(or "causal"). Received: ' + str(padding))
    return padding


def conv_output_length(input_length, filter_size, padding, stride, dilation=1):
    if input_length is None:
        return None
    assert padding in {'same', 'valid', 'full', 'causal'}
    dilated = filter_size + (filter_size - 1) * (dilation - 1)
    if padding == 'same':
        out = input_length
    elif padding == 'valid':
        out = input_length - dilated + 1
    elif padding == 'causal':
        out = input_length
    else:
        out = input_length + dilated - 1
    return (out + stride - 1) // stride


# --------------------------------------------------------------------------------------
# keras.backend
# --------------------------------------------------------------------------------------
def _normalize_data_format(value):
    if value is None:
        value = 'channels_last'
    data_format = value.lower()
    if data_format not in {'channels_first', 'channels_last'}:
        raise ValueError('The `data_format` argument must be one of '
                         '"channels_first", "channels_last". Received: ' + str(value))
    return data_format


def _tf_pads(n, k, s, d, padding):
    """(lo, hi) zero padding TensorFlow applies on one spatial axis."""
    if padding == 'valid':
        return 0, 0
    if padding == 'causal':
        return d * (k - 1), 0
    out = -(-n // s)
    total = max((out - 1) * s + (k - 1) * d + 1 - n, 0)
    return total // 2, total - total // 2


def _conv_nd(x, kernel, rank, strides, padding, data_format, dilation_rate):
    data_format = _normalize_data_format(data_format)
    strides = normalize_tuple(strides, rank, 'strides')
    dil = normalize_tuple(dilation_rate, rank, 'dilation_rate')
    if data_format == 'channels_last':
        x = x.movedim(-1, 1)
    ksz = kernel.shape[:rank]
    pads = []
    for ax in range(rank):
        lo, hi = _tf_pads(x.shape[2 + ax], ksz[ax], strides[ax], dil[ax], padding)
        pads.append((lo, hi))
    flat = []
    for lo, hi in reversed(pads):
        flat += [lo, hi]
    x = F.pad(x, flat)
    # (*k, in, out) -> (out, in, *k)
    perm = (rank + 1, rank) + tuple(range(rank))
    w = kernel.permute(*perm)
    fn = {1: F.conv1d, 2: F.conv2d, 3: F.conv3d}[rank]
    y = fn(x, w, None, strides, 0, dil)
    if data_format == 'channels_last':
        y = y.movedim(1, -1)
    return y


def _build_backend():
    K = types.ModuleType('keras.backend')
    K.normalize_data_format = _normalize_data_format
    K.image_data_format = lambda: 'channels_last'
    K.floatx = lambda: 'float64'
    K.shape = lambda x: tuple(x.shape)
    K.int_shape = lambda x: tuple(x.shape)
    K.ndim = lambda x: x.dim()
    K.concatenate = lambda tensors, axis=-1: torch.cat(list(tensors), dim=axis)
    K.dot = lambda a, b: a @ b
    K.sqrt = lambda v: np.sqrt(v)
    K.constant = lambda value, dtype=None, shape=None, name=None: torch.full(
        tuple(shape), float(value), dtype=DTYPE)

    def bias_add(x, bias, data_format=None):
        data_format = _normalize_data_format(data_format)
        if data_format == 'channels_first' and x.dim() > 2:
            return x + bias.reshape((1, -1) + (1,) * (x.dim() - 2))
        return x + bias
    K.bias_add = bias_add

    def conv1d(x, kernel, strides=1, padding='valid', data_format=None, dilation_rate=1):
        return _conv_nd(x, kernel, 1, strides, padding, data_format, dilation_rate)

    def conv2d(x, kernel, strides=(1, 1), padding='valid', data_format=None,
               dilation_rate=(1, 1)):
        return _conv_nd(x, kernel, 2, strides, padding, data_format, dilation_rate)

    def conv3d(x, kernel, strides=(1, 1, 1), padding='valid', data_format=None,
               dilation_rate=(1, 1, 1)):
        return _conv_nd(x, kernel, 3, strides, padding, data_format, dilation_rate)
    K.conv1d, K.conv2d, K.conv3d = conv1d, conv2d, conv3d
    K.reshape = lambda x, shape: x.reshape(tuple(int(v) for v in shape))
    K.epsilon = lambda: 1e-7
    K.function = lambda inputs, outputs, **kw: (lambda *a, **k: list(outputs))
    K.learning_phase = lambda: 0

    def ctc_batch_cost(y_true, y_pred, input_length, label_length):
        """keras.backend.ctc_batch_cost (Keras 2.x, TF backend): tf.nn.ctc_loss on inputs = log(y_pred + epsilon)
        taken as LOGITS (the op normalises them again), time-major, blank = last class, ctc_merge_repeated=True;
        returns (batch, 1)."""
        logits = torch.log(y_pred + 1e-7).transpose(0, 1)
        logp = torch.log_softmax(logits, dim=-1)
        il = torch.as_tensor(input_length).reshape(-1).long()
        ll = torch.as_tensor(label_length).reshape(-1).long()
        loss = F.ctc_loss(logp, torch.as_tensor(y_true).long(), il, ll, blank=y_pred.shape[-1] - 1,
                          reduction='none', zero_infinity=False)
        return loss.reshape(-1, 1)
    K.ctc_batch_cost = ctc_batch_cost
    return K


# --------------------------------------------------------------------------------------
# keras.activations / initializers / regularizers / constraints
# --------------------------------------------------------------------------------------
def _build_activations():
    m = types.ModuleType('keras.activations')

    def linear(x):
        return x

    def relu(x):
        return torch.relu(x)

    def tanh(x):
        return torch.tanh(x)

    def sigmoid(x):
        return torch.sigmoid(x)

    def softmax(x):
        return torch.softmax(x, dim=-1)
    table = {f.__name__: f for f in (linear, relu, tanh, sigmoid, softmax)}

    def get(identifier):
        if identifier is None:
            return linear
        if callable(identifier):
            return identifier
        return table[identifier]

    def serialize(fn):
        return fn.__name__
    m.get, m.serialize = get, serialize
    for k, v in table.items():
        setattr(m, k, v)
    return m


def _build_initializers():
    m = types.ModuleType('keras.initializers')

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

    def _compute_fans(shape, data_format='channels_last'):
        if len(shape) == 2:
            return shape[0], shape[1]
        rfs = int(np.prod(shape[:-2]))
        return shape[-2] * rfs, shape[-1] * rfs

    class RandomUniform(Initializer):
        def __init__(self, minval=-0.05, maxval=0.05, seed=None):
            self.minval, self.maxval = minval, maxval

        def __call__(self, shape, dtype=None):
            return np.random.uniform(self.minval, self.maxval, shape)

    class GlorotUniform(Initializer):
        def __call__(self, shape, dtype=None):
            fan_in, fan_out = _compute_fans(shape)
            limit = np.sqrt(6.0 / (fan_in + fan_out))
            return np.random.uniform(-limit, limit, shape)
    m.RandomUniform, m.Orthogonal = RandomUniform, Initializer

    def get(identifier):
        if identifier is None:
            return None
        if isinstance(identifier, str):
            return {'zeros': Zeros, 'ones': Ones, 'random_uniform': RandomUniform,
                    'glorot_uniform': GlorotUniform}[identifier]()
        if callable(identifier):
            return identifier
        raise ValueError('Could not interpret initializer identifier: ' + str(identifier))

    def serialize(init):
        return {'class_name': init.__class__.__name__, 'config': init.get_config()}
    m.Initializer, m.Zeros, m.Ones = Initializer, Zeros, Ones
    m._compute_fans, m.get, m.serialize = _compute_fans, get, serialize
    return m


def _build_passthrough(name):
    m = types.ModuleType(name)

    def get(identifier):
        if identifier is None:
            return None
        if callable(identifier):
            return identifier
        raise ValueError('stand-in supports only None/callable: ' + str(identifier))

    def serialize(obj):
        return None if obj is None else getattr(obj, '__name__', str(obj))
    m.get, m.serialize = get, serialize
    return m


# --------------------------------------------------------------------------------------
# keras.layers
# --------------------------------------------------------------------------------------
class InputSpec(object):
    def __init__(self, dtype=None, shape=None, ndim=None, max_ndim=None, min_ndim=None,
                 axes=None):
        self.dtype, self.shape, self.ndim = dtype, shape, ndim
        self.max_ndim, self.min_ndim, self.axes = max_ndim, min_ndim, axes or {}


_uid = {}
ALL_WEIGHTS = []          # every variable created through add_weight, in creation order (reset by the caller)
WEIGHT_HOOK = [None]      # optional callable(layer): runs once after build(), before the first call()
_FEED = []                # tensors handed out by Input(), in call order


def feed_inputs(tensors):
    _FEED[:] = list(tensors)


class Layer(object):
    def __init__(self, **kwargs):
        allowed = {'input_shape', 'batch_input_shape', 'batch_size', 'dtype', 'name',
                   'trainable', 'weights', 'input_dtype'}
        for k in kwargs:
            if k not in allowed:
                raise TypeError('Keyword argument not understood:', k)
        name = kwargs.get('name')
        if not name:
            prefix = self.__class__.__name__.lower()
            _uid[prefix] = _uid.get(prefix, 0) + 1
            name = prefix + '_' + str(_uid[prefix])
        self.name = name
        self.trainable = kwargs.get('trainable', True)
        self.built = False
        self.input_spec = None
        self.supports_masking = False
        self._weights = []

    def add_weight(self, *args, **kwargs):
        # Keras 2.2 `legacy_add_weight_support`: (shape, initializer=..., name=...)
        # and the modern (name, shape, ...) / all-keyword forms.
        args = list(args)
        if args and isinstance(args[0], (tuple, list)):
            kwargs['shape'] = tuple(args.pop(0))
        elif args:
            kwargs['name'] = args.pop(0)
            if args:
                kwargs['shape'] = tuple(args.pop(0))
        init = kwargs.get('initializer')
        if isinstance(init, str):
            init = sys.modules['keras.initializers'].get(init)
        value = init(kwargs['shape'])
        # Keras<=2.2: K.variable(initializer(shape)) -> shape of the RETURNED array.
        if isinstance(value, torch.Tensor):
            t = value.detach().clone().to(DTYPE)
        else:
            t = torch.tensor(np.asarray(value), dtype=DTYPE)
        t.requires_grad_(True)
        t.keras_name = kwargs.get('name')
        t.keras_layer = self.name
        self._weights.append(t)
        ALL_WEIGHTS.append(t)
        return t

    def build(self, input_shape):
        self.built = True

    def call(self, inputs):
        return inputs

    def __call__(self, inputs):
        if not self.built:
            self.build(tuple(inputs.shape))
            if WEIGHT_HOOK[0] is not None:
                WEIGHT_HOOK[0](self)
        return self.call(inputs)

    def get_config(self):
        return {'name': self.name, 'trainable': self.trainable}

    @property
    def weights(self):
        return list(self._weights)


# ---- stock layers used by the reference's model builders (restated Keras 2.x / TF semantics) ----------
def Input(shape=None, batch_shape=None, name=None, dtype=None, **kwargs):
    """keras.layers.Input: the stand-in runs eagerly, so an Input IS the next fed tensor."""
    if not _FEED:
        raise RuntimeError('keras stand-in: Input(%r) called with no fed tensor left (feed_inputs)' % (name,))
    return _FEED.pop(0)


class Dense(Layer):
    def __init__(self, units, activation=None, use_bias=True, kernel_initializer='glorot_uniform',
                 bias_initializer='zeros', kernel_regularizer=None, bias_regularizer=None,
                 activity_regularizer=None, kernel_constraint=None, bias_constraint=None, **kwargs):
        super(Dense, self).__init__(**kwargs)
        self.units, self.use_bias = units, use_bias
        self.activation = sys.modules['keras.activations'].get(activation)
        self.kernel_initializer, self.bias_initializer = kernel_initializer, bias_initializer

    def build(self, input_shape):
        self.kernel = self.add_weight(shape=(input_shape[-1], self.units), initializer=self.kernel_initializer,
                                      name='kernel')
        self.bias = self.add_weight(shape=(self.units,), initializer=self.bias_initializer, name='bias') \
            if self.use_bias else None
        self.built = True

    def call(self, inputs):
        out = inputs @ self.kernel
        if self.bias is not None:
            out = out + self.bias
        return self.activation(out)


class Flatten(Layer):
    def call(self, inputs):
        return inputs.reshape(inputs.shape[0], -1)


class Dropout(Layer):
    """Identity: the fixtures are taken at inference / with rate 0 (K.in_train_phase(dropped, inputs))."""

    def __init__(self, rate, noise_shape=None, seed=None, **kwargs):
        super(Dropout, self).__init__(**kwargs)
        self.rate = rate


class Permute(Layer):
    def __init__(self, dims, **kwargs):
        super(Permute, self).__init__(**kwargs)
        self.dims = tuple(dims)

    def call(self, inputs):
        return inputs.permute((0,) + self.dims)


class Lambda(Layer):
    def __init__(self, function, output_shape=None, mask=None, arguments=None, **kwargs):
        super(Lambda, self).__init__(**kwargs)
        self.function, self.arguments = function, arguments or {}

    def __call__(self, inputs):
        return self.function(inputs, **self.arguments)


class TimeDistributed(Layer):
    """keras.layers.TimeDistributed: (B, T, ...) -> reshape to (B*T, ...), apply the layer, reshape back."""

    def __init__(self, layer, **kwargs):
        super(TimeDistributed, self).__init__(**kwargs)
        self.layer = layer

    def __call__(self, inputs):
        b, t = inputs.shape[0], inputs.shape[1]
        y = self.layer(inputs.reshape((b * t,) + tuple(inputs.shape[2:])))
        return y.reshape((b, t) + tuple(y.shape[1:]))


class PReLU(Layer):
    """keras.layers.PReLU: param_shape = input_shape[1:], `param_shape[i - 1] = 1` for i in shared_axes
    (axis 0 therefore shares the LAST axis); f(x) = relu(x) - alpha * relu(-x)."""

    def __init__(self, alpha_initializer='zeros', alpha_regularizer=None, alpha_constraint=None,
                 shared_axes=None, **kwargs):
        super(PReLU, self).__init__(**kwargs)
        self.alpha_initializer = alpha_initializer
        if shared_axes is None:
            self.shared_axes = None
        elif not isinstance(shared_axes, (list, tuple)):
            self.shared_axes = [shared_axes]
        else:
            self.shared_axes = list(shared_axes)

    def build(self, input_shape):
        param_shape = list(input_shape[1:])
        if self.shared_axes is not None:
            for i in self.shared_axes:
                param_shape[i - 1] = 1
        self.alpha = self.add_weight(shape=tuple(param_shape), name='alpha', initializer=self.alpha_initializer)
        self.built = True

    def call(self, inputs):
        return torch.relu(inputs) - self.alpha * torch.relu(-inputs)


def _pool_same_valid(x, pool, strides, padding, mode):
    """Pool the LAST len(pool) axes of x (any leading axes) with TensorFlow padding: 'same' pads
    total = max((ceil(n/s) - 1) * s + k - n, 0) as (total // 2, rest); padded cells never win a max and are
    left out of an average's divisor."""
    rank = len(pool)
    lead = x.shape[:-rank]
    xp = x.reshape((-1, 1) + tuple(x.shape[-rank:]))
    pads = []
    for ax in reversed(range(rank)):
        lo, hi = _tf_pads(xp.shape[2 + ax], pool[ax], strides[ax], 1, padding)
        pads += [lo, hi]
    fmax = {1: F.max_pool1d, 2: F.max_pool2d}[rank]
    favg = {1: F.avg_pool1d, 2: F.avg_pool2d}[rank]
    if mode == 'max':
        y = fmax(F.pad(xp, pads, value=float('-inf')) if any(pads) else xp, pool, strides)
    elif any(pads):
        y = favg(F.pad(xp, pads), pool, strides) / favg(F.pad(torch.ones_like(xp), pads), pool, strides)
    else:
        y = favg(xp, pool, strides)
    return y.reshape(tuple(lead) + tuple(y.shape[-rank:]))


class _Pool(Layer):
    rank, mode = 1, 'max'

    def __init__(self, pool_size=2, strides=None, padding='valid', data_format=None, **kwargs):
        super(_Pool, self).__init__(**kwargs)
        self.pool_size = normalize_tuple(pool_size, self.rank, 'pool_size')
        self.strides = normalize_tuple(self.pool_size if strides is None else strides, self.rank, 'strides')
        self.padding = normalize_padding(padding)
        # keras: data_format None -> K.image_data_format() == 'channels_last' (1-D pooling is always (B, steps, C))
        self.data_format = _normalize_data_format(data_format)

    def call(self, inputs):
        if self.data_format == 'channels_last':             # spatial axes 1..rank, channels last
            x = inputs.movedim(-1, 1)
            return _pool_same_valid(x, self.pool_size, self.strides, self.padding, self.mode).movedim(1, -1)
        return _pool_same_valid(inputs, self.pool_size, self.strides, self.padding, self.mode)


class AveragePooling1D(_Pool):
    rank, mode = 1, 'avg'


class MaxPooling1D(_Pool):
    rank, mode = 1, 'max'


class MaxPooling2D(_Pool):
    rank, mode = 2, 'max'


class AveragePooling2D(_Pool):
    rank, mode = 2, 'avg'


class Model(object):
    def __init__(self, inputs=None, outputs=None, **kwargs):
        self.inputs, self.outputs = inputs, outputs


class _Placeholder(object):
    def __init__(self, *a, **k):
        raise NotImplementedError('keras stand-in placeholder')


def install():
    """Put the stand-in into sys.modules (idempotent) and return the backend module."""
    if 'keras' in sys.modules and getattr(sys.modules['keras'], '_qk_standin', False):
        return sys.modules['keras.backend']
    keras = types.ModuleType('keras')
    keras._qk_standin = True
    K = _build_backend()
    acts = _build_activations()
    inits = _build_initializers()
    regs = _build_passthrough('keras.regularizers')
    cons = _build_passthrough('keras.constraints')

    layers = types.ModuleType('keras.layers')
    layers.Layer, layers.InputSpec = Layer, InputSpec
    for n in ('Convolution1D', 'Convolution2D', 'add', 'multiply', 'Activation', 'concatenate', 'Conv1D', 'Conv2D',
              'AveragePooling3D', 'Add', 'Concatenate', 'BatchNormalization', 'Reshape', 'ConvLSTM2D',
              'SpatialDropout1D'):
        setattr(layers, n, _Placeholder)
    for cls in (Dense, Flatten, Dropout, Permute, Lambda, TimeDistributed, PReLU, AveragePooling1D, MaxPooling1D,
                MaxPooling2D, AveragePooling2D):
        setattr(layers, cls.__name__, cls)
    layers.Input = Input
    convolutional = types.ModuleType('keras.layers.convolutional')
    convolutional._Conv = _Placeholder
    merge = types.ModuleType('keras.layers.merge')
    merge._Merge = _Placeholder
    recurrent = types.ModuleType('keras.layers.recurrent')
    recurrent.Recurrent = _Placeholder
    layers.convolutional, layers.merge, layers.recurrent = convolutional, merge, recurrent

    models = types.ModuleType('keras.models')
    models.Model = Model
    models.load_model = models.save_model = _Placeholder
    regs.l2 = lambda l=0.01: (lambda w: l * (w * w).sum())
    extra = {}
    for modname, names in (('keras.callbacks', ('Callback', 'ModelCheckpoint', 'LearningRateScheduler')),
                           ('keras.datasets', ('cifar10', 'cifar100')),
                           ('keras.optimizers', ('SGD', 'Adam', 'RMSprop')),
                           ('keras.preprocessing', ()), ('keras.preprocessing.image', ('ImageDataGenerator',)),
                           ('keras.utils.np_utils', ('to_categorical',)),
                           ('keras.utils.training_utils', ('multi_gpu_model',)),
                           ('keras.backend.tensorflow_backend', ('set_session',)),
                           ('tensorflow', ())):
        mod = types.ModuleType(modname)
        for n in names:
            setattr(mod, n, _Placeholder)
        extra[modname] = mod
    extra['keras.preprocessing'].image = extra['keras.preprocessing.image']
    keras.callbacks, keras.datasets, keras.optimizers = extra['keras.callbacks'], extra['keras.datasets'], extra['keras.optimizers']
    keras.preprocessing = extra['keras.preprocessing']
    K.tensorflow_backend = extra['keras.backend.tensorflow_backend']

    utils = types.ModuleType('keras.utils')
    conv_utils = types.ModuleType('keras.utils.conv_utils')
    conv_utils.normalize_tuple = normalize_tuple
    conv_utils.normalize_padding = normalize_padding
    conv_utils.conv_output_length = conv_output_length
    generic_utils = types.ModuleType('keras.utils.generic_utils')
    generic_utils.serialize_keras_object = lambda o: o
    generic_utils.deserialize_keras_object = lambda o, **k: o
    utils.conv_utils, utils.generic_utils = conv_utils, generic_utils
    utils.np_utils, utils.training_utils = extra['keras.utils.np_utils'], extra['keras.utils.training_utils']

    keras.backend, keras.activations, keras.initializers = K, acts, inits
    keras.regularizers, keras.constraints = regs, cons
    keras.layers, keras.models, keras.utils = layers, models, utils
    mods = {
        'keras': keras, 'keras.backend': K, 'keras.activations': acts,
        'keras.initializers': inits, 'keras.regularizers': regs, 'keras.constraints': cons,
        'keras.layers': layers, 'keras.layers.convolutional': convolutional,
        'keras.layers.merge': merge, 'keras.layers.recurrent': recurrent,
        'keras.models': models, 'keras.utils': utils,
        'keras.utils.conv_utils': conv_utils, 'keras.utils.generic_utils': generic_utils,
    }
    mods.update(extra)
    sys.modules.update(mods)
    return K


def import_reference(path='/root/reference'):
    """Import the reference's complexnn package through the stand-in."""
    install()
    if path not in sys.path:
        sys.path.insert(0, path)
    import complexnn  # noqa: the reference package, imported where it lies
    return complexnn
