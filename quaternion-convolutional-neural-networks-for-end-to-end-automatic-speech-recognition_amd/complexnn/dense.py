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

    def __init__(self, units,
                 activation=None,
                 use_bias=True,
                 init_criterion='he',
                 kernel_initializer='quaternion',
                 bias_initializer='zeros',
                 kernel_regularizer=None,
                 bias_regularizer=None,
                 activity_regularizer=None,
                 kernel_constraint=None,
                 bias_constraint=None,
                 seed=None,
                 **kwargs):
        if 'input_shape' not in kwargs and 'input_dim' in kwargs:
            kwargs['input_shape'] = (kwargs.pop('input_dim'),)
        super(QuaternionDense, self).__init__(**kwargs)
        self.units = units
        self.q_units = units // 4
        self.activation = activations.get(activation)
        self.use_bias = use_bias
        self.init_criterion = init_criterion
        self.kernel_initializer = kernel_initializer
        self.bias_initializer = initializers.get(bias_initializer)
        self.kernel_regularizer = regularizers.get(kernel_regularizer)
        self.bias_regularizer = regularizers.get(bias_regularizer)
        self.activity_regularizer = regularizers.get(activity_regularizer)
        self.kernel_constraint = constraints.get(kernel_constraint)
        self.bias_constraint = constraints.get(bias_constraint)
        if seed is None:
            self.seed = np.random.randint(1, 10e6)
        else:
            self.seed = seed
        self.input_spec = InputSpec(ndim=2)
        self.supports_masking = True

    @property
    def kernel(self):
        return self.r          # the reference names the variable 'r' (dense.py:103-109)

    def build(self, input_shape):
        assert len(input_shape) == 2
        assert input_shape[-1] % 2 == 0
        if input_shape[-1] % 4 or self.units % 4:
            # the reference passes its assert for in%4==2 and then dies inside K.dot with a
            # shape mismatch; fail here with a message instead
            raise ValueError('QuaternionDense needs input width and units divisible by 4, got %d / %d'
                             % (input_shape[-1], self.units))
        input_dim = input_shape[-1] // 4
        kernel_shape = (input_dim, self.units)
        init_shape = (input_dim, self.q_units)
        self.kernel_init = qdense_init(init_shape, self.init_criterion)
        self.add_weight('r', kernel_shape, initializer=self.kernel_init,
                        regularizer=self.kernel_regularizer, constraint=self.kernel_constraint)
        if self.use_bias:
            self.add_weight('bias', (self.units,), initializer='zeros',
                            regularizer=self.bias_regularizer, constraint=self.bias_constraint)
        else:
            self.bias = None
        self.input_spec = InputSpec(ndim=2, axes={-1: 4 * input_dim})
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
        assert input_shape and len(input_shape) == 2
        assert input_shape[-1]
        output_shape = list(input_shape)
        output_shape[-1] = self.units
        return tuple(output_shape)

    def get_config(self):
        if self.kernel_initializer == 'quaternion':
            ki = self.kernel_init
        else:
            ki = initializers.serialize(self.kernel_initializer)
        config = {
            'units': self.units,
            'activation': activations.serialize(self.activation),
            'use_bias': self.use_bias,
            'init_criterion': self.init_criterion,
            'kernel_initializer': ki,
            'bias_initializer': initializers.serialize(self.bias_initializer),
            'kernel_regularizer': regularizers.serialize(self.kernel_regularizer),
            'bias_regularizer': regularizers.serialize(self.bias_regularizer),
            'activity_regularizer': regularizers.serialize(self.activity_regularizer),
            'kernel_constraint': constraints.serialize(self.kernel_constraint),
            'bias_constraint': constraints.serialize(self.bias_constraint),
            'seed': self.seed,
        }
        base_config = super(QuaternionDense, self).get_config()
        return dict(list(base_config.items()) + list(config.items()))
