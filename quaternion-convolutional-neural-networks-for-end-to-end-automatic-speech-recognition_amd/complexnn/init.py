"""Quaternion weight initialisers -- host-side numpy, same draws as the reference.

Mirrors complexnn/init.py of the reference (qconv_init :22-93, qdense_init :95-155,
sqrt_init :157-159).  The recipe (polar form of a quaternion, He / Glorot scale):

  s        = 1/sqrt(2*fan_in)                (he)      or 1/sqrt(2*(fan_in+fan_out)) (glorot)
  v_i,j,k ~ U(0,1) from the GLOBAL numpy RNG  (init.py:70-72), normalised by
             sqrt(v_i^2+v_j^2+v_k^2) + 1e-4   (init.py:74-78; all-positive octant)
  modulus ~ Rayleigh(s), phase ~ U(-pi, pi)   from RandomState(seed), seed 1337 unless given
                                               (init.py:46,83-85)
  w_r = m cos(phase),  w_{i,j,k} = m v_{i,j,k} sin(phase)        (init.py:87-90)
  result = concatenate([w_r, w_i, w_j, w_k], axis=-1), float64   (init.py:91-93)

The only intentional difference: the per-weight Python normalisation loop (O(N) interpreted,
seconds for >= 1M weights) is a vectorised numpy expression producing the same float64 bits.
"""
import numpy as np
from numpy.random import RandomState

from ..keras_like import Initializer, _compute_fans


def _scale(criterion, fan_in, fan_out):
    if criterion == 'glorot':
        return 1. / np.sqrt(2 * (fan_in + fan_out))
    if criterion == 'he':
        return 1. / np.sqrt(2 * fan_in)
    raise ValueError('Invalid criterion: ' + criterion)


def _polar_quaternion_weights(shape, s, seed):
    n = int(np.prod(shape))
    # three global-RNG draws in the reference's order: i, then j, then k
    v_i = np.random.uniform(0.0, 1.0, n)
    v_j = np.random.uniform(0.0, 1.0, n)
    v_k = np.random.uniform(0.0, 1.0, n)
    norm = np.sqrt(v_i ** 2 + v_j ** 2 + v_k ** 2) + 0.0001
    v_i = (v_i / norm).reshape(shape)
    v_j = (v_j / norm).reshape(shape)
    v_k = (v_k / norm).reshape(shape)
    rng = RandomState(seed)
    modulus = rng.rayleigh(scale=s, size=shape)
    phase = rng.uniform(low=-np.pi, high=np.pi, size=shape)
    sin = np.sin(phase)
    return np.concatenate([modulus * np.cos(phase), modulus * v_i * sin, modulus * v_j * sin,
                           modulus * v_k * sin], axis=-1)


class qconv_init(Initializer):
    """Initialiser of a compact quaternion conv kernel (*kernel_size, input_dim, 4*nb_filters)."""

    def __init__(self, kernel_size, input_dim, weight_dim, nb_filters=None, criterion='he', seed=None):
        assert len(kernel_size) == weight_dim and weight_dim in {0, 1, 2, 3}
        self.nb_filters = nb_filters
        self.kernel_size = kernel_size
        self.input_dim = input_dim
        self.weight_dim = weight_dim
        self.criterion = criterion
        self.seed = 1337 if seed is None else seed

    def __call__(self, shape=None, dtype=None):
        # the requested `shape` is ignored, as in the reference (init.py:48-54)
        if self.nb_filters is not None:
            kernel_shape = tuple(self.kernel_size) + (int(self.input_dim), self.nb_filters)
        else:
            kernel_shape = (int(self.input_dim), self.kernel_size[-1])
        fan_in, fan_out = _compute_fans(tuple(self.kernel_size) + (self.input_dim, self.nb_filters))
        s = _scale(self.criterion, fan_in, fan_out)
        return _polar_quaternion_weights(kernel_shape, s, self.seed)

    def get_config(self):
        return {'kernel_size': tuple(self.kernel_size), 'input_dim': self.input_dim,
                'weight_dim': self.weight_dim, 'nb_filters': self.nb_filters,
                'criterion': self.criterion, 'seed': self.seed}


class qdense_init(Initializer):
    """Initialiser of a compact quaternion dense kernel (in_q, 4*q_units); shape=(in_q, q_units)."""

    def __init__(self, shape, criterion='he', seed=None):
        self.shape = tuple(shape)
        self.criterion = criterion
        self.seed = 1337 if seed is None else seed

    def __call__(self, shape=None, dtype=None):
        fan_in, fan_out = self.shape[0], self.shape[1]
        s = _scale(self.criterion, fan_in, fan_out)
        return _polar_quaternion_weights(self.shape, s, self.seed)

    def get_config(self):
        return {'shape': self.shape, 'criterion': self.criterion, 'seed': self.seed}


class sqrt_init(Initializer):
    """Constant 1/sqrt(2) -- only for the (never read) gamma weights, init.py:157-159."""

    def __call__(self, shape, dtype=None):
        return np.full(tuple(shape), 1.0 / np.sqrt(2.0))
