"""Quaternion component getters -- drop-in for complexnn/utils.py (pure views, no kernel).

`get_{r,i,j,k}part_first` slice one component block out of a component-planar tensor
(utils.py:17-79); like the reference they hard-wire data_format='channels_first', so every
tensor that is not 3-D is sliced on axis 1 and 3-D tensors on the last axis.
"""
from ..keras_like import Layer


def _part(x, index):
    ndim = x.dim()
    if ndim != 3 or ndim == 2:
        input_dim = x.shape[1] // 4
        return x[:, index * input_dim:(index + 1) * input_dim]
    input_dim = x.shape[-1] // 4
    return x[..., index * input_dim:(index + 1) * input_dim]


def get_rpart_first(x):
    return _part(x, 0)


def get_ipart_first(x):
    return _part(x, 1)


def get_jpart_first(x):
    return _part(x, 2)


def get_kpart_first(x):
    return _part(x, 3)


def getpart_quaternion_output_shape_first(input_shape):
    returned_shape = list(input_shape[:])
    axis = 1 if len(returned_shape) != 3 else -1
    returned_shape[axis] = returned_shape[axis] // 4
    return tuple(returned_shape)


class _GetPart(Layer):
    _index = 0

    def call(self, inputs):
        return _part(inputs, self._index)

    def compute_output_shape(self, input_shape):
        return getpart_quaternion_output_shape_first(input_shape)


class GetRFirst(_GetPart):
    _index = 0


class GetIFirst(_GetPart):
    _index = 1


class GetJFirst(_GetPart):
    _index = 2


class GetKFirst(_GetPart):
    _index = 3
