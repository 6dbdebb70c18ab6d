"""Export surface of the reference's `complexnn` package (complexnn/__init__.py:9-17)."""
from .conv import (QuaternionConv, QuaternionConv1D, QuaternionConv2D, QuaternionConv3D,
                   QuaternionConvolution1D, QuaternionConvolution2D, QuaternionConvolution3D)
from .dense import QuaternionDense
from .init import sqrt_init, qdense_init, qconv_init
from .utils import (GetRFirst, GetIFirst, GetJFirst, GetKFirst,
                    getpart_quaternion_output_shape_first, get_rpart_first, get_ipart_first,
                    get_jpart_first, get_kpart_first)
