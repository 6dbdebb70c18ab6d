"""Quaternion fully-connected layer on the MI355X HIP path -- drop-in for complexnn/dense.py.

Same constructor, weight names ('r', 'bias'), shapes, `compute_output_shape`, `get_config`
keys and assertion behaviour as the reference (QuaternionDense, dense.py:16-193).
`call()` is one fused HIP launch (qk_dense_fwd) instead of slice / 6 neg / 5 concat /
K.dot / bias_add / activation (dense.py:131-162).

NB the block table dense.py:139-143 builds is the TRANSPOSE of the convolution's, i.e. the
layer computes conj(W) (x) x, not W (x) x.  That asymmetry is reproduced, not "fixed".
"""
import numpy as np

from .. import functional as F
from ..keras_like import InputSpec, Layer, activations, constraints, initializers, regularizers
from .init import qdense_init


class QuaternionDense(Layer):
    """`units` is the TOTAL real output width (4 * quaternion units, dense.py:74-75).

    Input (batch, input_dim) with input_dim = 4*in_q laid out r|i|j|k; output (batch, units).
    As in the reference: `kernel_initializer` is stored but the kernel is always drawn by
    qdense_init (dense.py:101) and the bias always starts at zero (dense.py:115).
    """

    # (getter module, attribute) of the Keras-style arguments that are resolved through a registry
    _RESOLVED = (('bias_initializer', initializers), ('kernel_regularizer', regularizers),
                 ('bias_regularizer', regularizers), ('activity_regularizer', regularizers),
                 ('kernel_constraint', constraints), ('bias_constraint', constraints))

    def __init__(self, units, activation=None, use_bias=True, init_criterion='he',
                 kernel_initializer='quaternion', bias_initializer='zeros', kernel_regularizer=None,
                 bias_regularizer=None, activity_regularizer=None, kernel_constraint=None,
                 bias_constraint=None, seed=None, **kwargs):
        given = dict(locals())
        if 'input_dim' in kwargs and 'input_shape' not in kwargs:       # Keras shorthand (dense.py:71-72)
            kwargs['input_shape'] = (kwargs.pop('input_dim'),)
        super(QuaternionDense, self).__init__(**kwargs)
        self.units, self.q_units = units, units // 4
        self.use_bias, self.init_criterion = use_bias, init_criterion
        self.activation = activations.get(activation)
        self.kernel_initializer = kernel_initializer                      # stored, never used (dense.py:101)
        for attr, registry in self._RESOLVED:
            setattr(self, attr, registry.get(given[attr]))
        self.seed = int(np.random.randint(1, 10e6)) if seed is None else seed   # drawn, unused (dense.py:86-89)
        self.input_spec = InputSpec(ndim=2)
        self.supports_masking = True

    @property
    def kernel(self):
        return self.r          # the reference names the variable 'r' (dense.py:103-109)

    def build(self, input_shape):
        width = input_shape[-1] if len(input_shape) == 2 else None
        assert len(input_shape) == 2                       # same assertions as dense.py:94-95
        assert width % 2 == 0
        if width % 4 or self.units % 4:
            # the reference passes its assert for width % 4 == 2 and then dies inside K.dot with a
            # shape mismatch; fail here with a message instead
            raise ValueError('QuaternionDense needs input width and units divisible by 4, got %d / %d'
                             % (width, self.units))
        in_q = width // 4
        self.kernel_init = qdense_init((in_q, self.q_units), self.init_criterion)
        self.add_weight('r', (in_q, self.units), initializer=self.kernel_init,
                        regularizer=self.kernel_regularizer, constraint=self.kernel_constraint)
        if self.use_bias:
            self.add_weight('bias', (self.units,), initializer='zeros',
                            regularizer=self.bias_regularizer, constraint=self.bias_constraint)
        else:
            self.bias = None
        self.input_spec = InputSpec(ndim=2, axes={-1: width})
        self.built = True

    def call(self, inputs):
        if inputs.dim() != 2:
            raise ValueError('%s expects 2-D input (wrap it in TimeDistributed for sequences), got %s'
                             % (self.name, tuple(inputs.shape)))
        name = activations.serialize(self.activation)
        fused = name if name in ('linear', 'relu') else 'linear'
        out = F.quaternion_dense(inputs, self.r, self.bias, activation=fused)
        if fused != name:
            out = self.activation(out)
        return out

    def compute_output_shape(self, input_shape):
        assert input_shape and len(input_shape) == 2 and input_shape[-1]
        return tuple(input_shape[:-1]) + (self.units,)

    def get_config(self):
        cfg = super(QuaternionDense, self).get_config()
        cfg.update(units=self.units, activation=activations.serialize(self.activation), use_bias=self.use_bias,
                   init_criterion=self.init_criterion,
                   kernel_initializer=(self.kernel_init if self.kernel_initializer == 'quaternion'
                                       else initializers.serialize(self.kernel_initializer)))
        for attr, registry in self._RESOLVED:
            cfg[attr] = registry.serialize(getattr(self, attr))
        cfg['seed'] = self.seed
        return cfg
