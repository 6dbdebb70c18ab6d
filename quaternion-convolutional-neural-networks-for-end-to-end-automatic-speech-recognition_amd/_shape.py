"""Host-side shape arithmetic the reference delegates to Keras / TensorFlow.

`conv_output_length`, `normalize_*` follow keras.utils.conv_utils as used at
complexnn/conv.py:124-128,347-372; `tf_pads` is TensorFlow's SAME rule (extra zero goes to
the high side) plus Keras' 'causal' left padding (conv.py:432-436, conv1d only).
"""


def normalize_tuple(value, n, name):
    if isinstance(value, int):
        return (value,) * n
    try:
        value_tuple = tuple(value)
    except TypeError:
        raise ValueError('The `' + name + '` argument must be a tuple of ' + str(n) +
                         ' integers. Received: ' + str(value))
    if len(value_tuple) != n:
        raise ValueError('The `' + name + '` argument must be a tuple of ' + str(n) +
                         ' integers. Received: ' + str(value))
    for v in value_tuple:
        try:
            int(v)
        except ValueError:
            raise ValueError('The `' + name + '` argument must be a tuple of ' + str(n) +
                             ' integers. Received: ' + str(value))
    return tuple(int(v) for v in value_tuple)


def normalize_padding(value):
    padding = value.lower()
    if padding not in {'valid', 'same', 'causal'}:
        raise ValueError('The `padding` argument must be one of "valid", "same" (or "causal" '
                         'for 1D convolutions). Received: ' + str(padding))
    return padding


def normalize_data_format(value):
    if value is None:
        value = 'channels_last'
    data_format = value.lower()
    if data_format not in {'channels_first', 'channels_last'}:
        raise ValueError('The `data_format` argument must be one of "channels_first", '
                         '"channels_last". Received: ' + str(value))
    return data_format


def conv_output_length(input_length, filter_size, padding, stride, dilation=1):
    if input_length is None:
        return None
    assert padding in {'same', 'valid', 'causal'}
    dilated = filter_size + (filter_size - 1) * (dilation - 1)
    if padding in ('same', 'causal'):
        output_length = input_length
    else:
        output_length = input_length - dilated + 1
    return (output_length + stride - 1) // stride


def tf_pads(n, k, stride, dilation, padding):
    """(zeros before, zeros after) on one spatial axis."""
    if padding == 'valid':
        return 0, 0
    if padding == 'causal':
        return dilation * (k - 1), 0
    out = -(-n // stride)
    total = max((out - 1) * stride + (k - 1) * dilation + 1 - n, 0)
    return total // 2, total - total // 2
