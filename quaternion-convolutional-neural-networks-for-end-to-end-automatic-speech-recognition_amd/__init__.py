"""MI355X-native quaternion-layer engine (gfx950 HIP kernels behind the Keras-style layer API
of Orkis-Research/Quaternion-CNN-for-E2E-ASR's `complexnn`).

    import qcnn_amd                      # root shim; the directory name is not an identifier
    from qcnn_amd.complexnn import QuaternionConv1D, QuaternionDense

The layers run only through libqk_hip.so (include/qk.h); there is no CPU fallback.
"""
from . import _lib, functional, keras_like          # noqa: F401
from . import complexnn                             # noqa: F401
from . import layers, data, dp                      # noqa: F401
from . import models                                # noqa: F401
from .complexnn import *                            # noqa: F401,F403
from .functional import invalidate_cached_kernels   # noqa: F401  (after raw `.data` writes to layer weights)

__version__ = '0.1.0'


def library_path():
    return _lib.LIB_PATH


def build_library(force=False, verbose=False):
    from ._build import build
    return build(force=force, verbose=verbose)
