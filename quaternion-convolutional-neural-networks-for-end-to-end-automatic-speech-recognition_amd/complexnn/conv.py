"""Quaternion convolution layers on the MI355X HIP path -- drop-in for complexnn/conv.py.

Same classes, constructor signatures, weight names / shapes, `compute_output_shape` and
`get_config` keys as the reference (QuaternionConv conv.py:24-403, QuaternionConv1D :407-524,
QuaternionConv2D :527-658, QuaternionConv3D :661-794, aliases :818-820).  `call()` is where
the two differ: the reference slices the compact kernel, builds the 4x-expanded real kernel
with 6 negations + 5 concats and runs one real K.conv (conv.py:294-343) every step; here one
fused HIP launch (qk_conv_fwd, include/qk.h) consumes the compact kernel directly and
applies bias + activation in its epilogue; backward is two more launches.

Documented deviations from the reference (all are reference defects):
  * get_config() works.  The reference's raises NameError as soon as it serialises the bias
    initializer (sanitizedInitSer references undefined names, conv.py:806-814).
  * normalize_weight=True creates the ten gamma vectors like conv.py:183-256 does (the
    reference crashes there, passing the sqrt_init CLASS to add_weight); as in the reference
    they are never read by call().
"""
import numpy as np

from .. import functional as F
from .._shape import conv_output_length, normalize_data_format, normalize_padding, normalize_tuple
from ..keras_like import InputSpec, Initializer, Layer, activations, constraints, initializers, regularizers
from .init import qconv_init, sqrt_init

_QUATERNION_INIT_NAMES = ["complex", "complex_independent", "glorot_complex", "he_complex",
                          "quaternion", "quaternion_independent"]


def sanitizedInitGet(init):
    """Initializer lookup of conv.py:796-804: 'sqrt_init' and the quaternion-family names pass through,
    everything else goes to the Keras registry."""
    if init == "sqrt_init":
        return sqrt_init
    if isinstance(init, str) and init in _QUATERNION_INIT_NAMES:
        return init
    return initializers.get(init)


def sanitizedInitSer(init):
    """What conv.py:806-814 intends (its isinstance checks name undefined classes)."""
    if init is sqrt_init or isinstance(init, sqrt_init):
        return "sqrt_init"
    if isinstance(init, qconv_init):
        return "quaternion"
    return init if isinstance(init, str) else initializers.serialize(init)


def _instantiate(init):
    return init() if isinstance(init, type) and issubclass(init, Initializer) else init


class QuaternionConv(Layer):
    """Abstract N-D quaternion convolution layer (rank 1, 2 or 3).

    y = activation(W (x) x + bias), W (x) x the Hamilton product with the weight on the left,
    every scalar product a real cross-correlation (conv.py:327-334).  Channels are
    component-planar: input channel a*Cq+c, output channel b*filters+f, a,b in (r,i,j,k).
    `filters` counts QUATERNION filters: the layer emits 4*filters real channels.
    """

    # constructor arguments resolved through a registry, grouped by how (get, serialize)
    _INITIALIZERS = ('kernel_initializer', 'bias_initializer', 'gamma_diag_initializer', 'gamma_off_initializer')
    _REGULARIZERS = ('kernel_regularizer', 'bias_regularizer', 'gamma_diag_regularizer', 'gamma_off_regularizer',
                     'activity_regularizer')
    _CONSTRAINTS = ('kernel_constraint', 'bias_constraint', 'gamma_diag_constraint', 'gamma_off_constraint')
    _PLAIN = ('use_bias', 'normalize_weight', 'init_criterion', 'spectral_parametrization', 'epsilon')

    def __init__(self, rank, filters, kernel_size, strides=1, padding='valid', data_format='channels_last',
                 dilation_rate=1, activation=None, use_bias=True, normalize_weight=False,
                 kernel_initializer='quaternion', bias_initializer='zeros', gamma_diag_initializer=sqrt_init,
                 gamma_off_initializer='zeros', kernel_regularizer=None, bias_regularizer=None,
                 gamma_diag_regularizer=None, gamma_off_regularizer=None, activity_regularizer=None,
                 kernel_constraint=None, bias_constraint=None, gamma_diag_constraint=None,
                 gamma_off_constraint=None, init_criterion='he', seed=None, spectral_parametrization=False,
                 epsilon=1e-7, internal_layout='channels_last', **kwargs):
        given = dict(locals())
        super(QuaternionConv, self).__init__(**kwargs)
        self.rank, self.filters = rank, filters
        for attr in ('kernel_size', 'strides', 'dilation_rate'):          # int or tuple -> rank-tuple
            setattr(self, attr, normalize_tuple(given[attr], rank, attr))
        self.padding = normalize_padding(padding)
        self.data_format = normalize_data_format(data_format)
        self.activation = activations.get(activation)
        for attr in self._PLAIN:
            setattr(self, attr, given[attr])
        for attr in self._INITIALIZERS:
            setattr(self, attr, sanitizedInitGet(given[attr]))
        for attr in self._REGULARIZERS:
            setattr(self, attr, regularizers.get(given[attr]))
        for attr in self._CONSTRAINTS:
            setattr(self, attr, constraints.get(given[attr]))
        self.seed = int(np.random.randint(1, 10e6)) if seed is None else seed    # never used afterwards (conv.py:148-151)
        # where channels_first data physically lives (not a reference option): 'channels_last'
        # keeps it NHWC-style in HBM behind a channels_first-shaped view; 'native' does not.
        if internal_layout not in ('channels_last', 'native'):
            raise ValueError('internal_layout must be "channels_last" or "native"')
        self.internal_layout = internal_layout
        self.input_spec = InputSpec(ndim=rank + 2)

    def build(self, input_shape):
        channel_axis = 1 if self.data_format == 'channels_first' else -1
        if input_shape[channel_axis] is None:                      # same error as conv.py:161-163
            raise ValueError('The channel dimension of the inputs should be defined. Found `None`.')
        input_dim = input_shape[channel_axis] // 4
        # the attribute keeps the reference's (misleading) value; the variable is (*k, Cq, 4F)
        self.kernel_shape = self.kernel_size + (input_dim, self.filters)

        kls = {'quaternion': qconv_init}[self.kernel_initializer]     # KeyError otherwise (conv.py:167)
        kern_init = kls(kernel_size=self.kernel_size, input_dim=input_dim, weight_dim=self.rank,
                        nb_filters=self.filters, criterion=self.init_criterion)
        self.add_weight('kernel', self.kernel_shape, initializer=kern_init,
                        regularizer=self.kernel_regularizer, constraint=self.kernel_constraint)

        if self.normalize_weight:
            gamma_shape = (input_dim * self.filters,)
            diag = ('rr', 'ii', 'jj', 'jk')        # conv.py:188,217,239,246 use the diag initializer
            for tag in ('rr', 'ri', 'rj', 'rk', 'ii', 'ij', 'ik', 'jj', 'jk', 'kk'):
                d = tag in diag
                self.add_weight('gamma_' + tag, gamma_shape,
                                initializer=_instantiate(self.gamma_diag_initializer if d else self.gamma_off_initializer),
                                regularizer=self.gamma_diag_regularizer if d else self.gamma_off_regularizer,
                                constraint=self.gamma_diag_constraint if d else self.gamma_off_constraint)
        else:
            for tag in ('rr', 'ri', 'rj', 'rk', 'ii', 'ij', 'ik', 'jj', 'jk', 'kk'):
                setattr(self, 'gamma_' + tag, None)

        if self.use_bias:
            self.add_weight('bias', (4 * self.filters,), initializer=_instantiate(self.bias_initializer),
                            regularizer=self.bias_regularizer, constraint=self.bias_constraint)
        else:
            self.bias = None

        self.input_spec = InputSpec(ndim=self.rank + 2, axes={channel_axis: input_dim * 4})
        self.built = True

    def call(self, inputs):
        if inputs.dim() != self.rank + 2:
            raise ValueError('%s expects %d-D input, got shape %s'
                             % (self.name, self.rank + 2, tuple(inputs.shape)))
        name = activations.serialize(self.activation)
        fused = name if name in ('linear', 'relu') else 'linear'
        out = F.quaternion_conv(inputs, self.kernel, self.bias, strides=self.strides,
                                padding=self.padding, data_format=self.data_format,
                                dilation_rate=self.dilation_rate, activation=fused,
                                internal_layout=self.internal_layout)
        if fused != name:
            out = self.activation(out)
        return out

    def compute_output_shape(self, input_shape):
        first = self.data_format == 'channels_first'
        space = input_shape[2:] if first else input_shape[1:-1]
        out = tuple(conv_output_length(n, k, padding=self.padding, stride=st, dilation=dl)
                    for n, k, st, dl in zip(space, self.kernel_size, self.strides, self.dilation_rate))
        channels = (4 * self.filters,)
        return (input_shape[0],) + (channels + out if first else out + channels)

    def get_config(self):
        cfg = super(QuaternionConv, self).get_config()
        for attr in ('rank', 'filters', 'kernel_size', 'strides', 'padding', 'data_format', 'dilation_rate'):
            cfg[attr] = getattr(self, attr)
        cfg['activation'] = activations.serialize(self.activation)
        for attr in ('use_bias', 'normalize_weight'):
            cfg[attr] = getattr(self, attr)
        for attr in self._INITIALIZERS:
            cfg[attr] = sanitizedInitSer(getattr(self, attr))
        for attr in self._REGULARIZERS:
            cfg[attr] = regularizers.serialize(getattr(self, attr))
        for attr in self._CONSTRAINTS:
            cfg[attr] = constraints.serialize(getattr(self, attr))
        for attr in ('init_criterion', 'spectral_parametrization'):
            cfg[attr] = getattr(self, attr)
        return cfg


def _subclass_kwargs(local):
    """kwargs the 1D/2D/3D constructors forward to QuaternionConv (they drop `seed`:
    conv.py:495,630,766 accept it and never pass it on)."""
    names = ('filters', 'kernel_size', 'strides', 'padding', 'dilation_rate', 'activation',
             'use_bias', 'kernel_initializer', 'bias_initializer', 'kernel_regularizer',
             'bias_regularizer', 'activity_regularizer', 'kernel_constraint', 'bias_constraint',
             'init_criterion', 'spectral_parametrization')
    return {k: local[k] for k in names}


class QuaternionConv1D(QuaternionConv):
    """1-D quaternion convolution (temporal).  Input (batch, steps, 4*Cq) for channels_last;
    padding may also be 'causal' (conv.py:432-436).  Constructor == conv.py:480-518."""

    def __init__(self, filters, kernel_size, strides=1, padding='valid', data_format='channels_last',
                 dilation_rate=1, activation=None, use_bias=True, kernel_initializer='quaternion',
                 bias_initializer='zeros', kernel_regularizer=None, bias_regularizer=None,
                 activity_regularizer=None, kernel_constraint=None, bias_constraint=None, seed=None,
                 init_criterion='he', spectral_parametrization=False, **kwargs):
        super(QuaternionConv1D, self).__init__(rank=1, data_format=data_format,
                                               **_subclass_kwargs(locals()), **kwargs)

    def get_config(self):
        config = super(QuaternionConv1D, self).get_config()
        config.pop('rank')
        config.pop('data_format')       # as conv.py:520-524
        return config


class QuaternionConv2D(QuaternionConv):
    """2-D quaternion convolution.  Input (batch, rows, cols, 4*Cq) or (batch, 4*Cq, rows, cols).
    Constructor == conv.py:615-653."""

    def __init__(self, filters, kernel_size, strides=(1, 1), padding='valid', data_format='channels_last',
                 dilation_rate=(1, 1), activation=None, use_bias=True, kernel_initializer='quaternion',
                 bias_initializer='zeros', kernel_regularizer=None, bias_regularizer=None,
                 activity_regularizer=None, kernel_constraint=None, bias_constraint=None, seed=None,
                 init_criterion='he', spectral_parametrization=False, **kwargs):
        super(QuaternionConv2D, self).__init__(rank=2, data_format=data_format,
                                               **_subclass_kwargs(locals()), **kwargs)

    def get_config(self):
        config = super(QuaternionConv2D, self).get_config()
        config.pop('rank')
        return config


class QuaternionConv3D(QuaternionConv):
    """3-D quaternion convolution.  Constructor == conv.py:751-789."""

    def __init__(self, filters, kernel_size, strides=(1, 1, 1), padding='valid', data_format='channels_last',
                 dilation_rate=(1, 1, 1), activation=None, use_bias=True, kernel_initializer='quaternion',
                 bias_initializer='zeros', kernel_regularizer=None, bias_regularizer=None,
                 activity_regularizer=None, kernel_constraint=None, bias_constraint=None, seed=None,
                 init_criterion='he', spectral_parametrization=False, **kwargs):
        super(QuaternionConv3D, self).__init__(rank=3, data_format=data_format,
                                               **_subclass_kwargs(locals()), **kwargs)

    def get_config(self):
        config = super(QuaternionConv3D, self).get_config()
        config.pop('rank')
        return config


# Aliases (conv.py:818-820)
QuaternionConvolution1D = QuaternionConv1D
QuaternionConvolution2D = QuaternionConv2D
QuaternionConvolution3D = QuaternionConv3D
