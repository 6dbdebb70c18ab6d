"""ctypes binding of libqk_hip.so -- the C-ABI declared in include/qk.h.

The product path has NO fallback: if the library is missing or a call fails, a RuntimeError
is raised.  (The CPU oracle under oracle/ is test infrastructure and is never imported here.)
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.realpath(__file__))
# QK_LIB: diagnostic override (A/B runs of two builds of the library on one GPU box, tools/ab_layers.py)
LIB_PATH = os.environ.get('QK_LIB') or os.path.join(_HERE, 'libqk_hip.so')

QK_F32, QK_BF16, QK_F16 = 0, 1, 2
QK_CH_LAST, QK_CH_FIRST = 0, 1
QK_ACT_LINEAR, QK_ACT_RELU = 0, 1
QK_OP_FWD, QK_OP_BWD_DATA, QK_OP_BWD_WEIGHT, QK_OP_BWD = 0, 1, 2, 3
QK_KERNEL_TAPS_MAJOR, QK_KERNEL_CHANNEL_MAJOR = 0, 1            # qk_conv_desc_t.kernel_order
QK_BWD_MASK_DX, QK_BWD_DY_PREMASKED, QK_BWD_ACCUMULATE = 1, 2, 4      # flags of qk_*_bwd_chain
QK_DBG_NO_MFMA16, QK_DBG_NO_BAND16, QK_DBG_NO_BAND32, QK_DBG_WGRAD16_ONE_TAP, QK_DBG_BAND16_8WAVES = 1, 2, 4, 8, 16   # qk_set_debug_flags
QK_DBG_NO_WGRAD_BAND = 32
QK_DBG_NO_POINT16 = 64
QK_DBG_CTC_TWO_SWEEPS = 128
QK_DBG_DETERMINISTIC = 0x10000
QK_DBG_WGRAD_BAND_V1 = 0x20000
QK_DBG_NO_SMALL16 = 0x40000
# graph-level A/B switches (include/qk.h): kept in the library's mask, acted on by models/interspeech_model.py and layers.py
QK_DBG_NO_CONV_CHAIN, QK_DBG_NO_FUSED_PRELU, QK_DBG_NO_FUSED_DROPOUT, QK_DBG_NO_FUSED_CTC = 0x100000, 0x200000, 0x400000, 0x800000
QK_DBG_NO_FUSED_FIRST, QK_DBG_NO_DENSE_IN_CHAIN, QK_DBG_NO_FUSED_SOFTMAX = 0x1000000, 0x2000000, 0x4000000
QK_ERR_INVALID_ARG, QK_ERR_UNSUPPORTED, QK_ERR_WORKSPACE, QK_ERR_LAUNCH = -1, -2, -3, -4
QK_PATH_NAMES = {0: 'none', 1: 'mfma16', 2: 'mfma16_band', 3: 'fp32_mfma', 4: 'mfma16_point', 5: 'mfma16_small'}               # qk_last_path

I32 = ctypes.c_int32


class ConvDesc(ctypes.Structure):
    _fields_ = [('rank', I32), ('batch', I32), ('in_spatial', I32 * 3), ('out_spatial', I32 * 3),
                ('cq', I32), ('fq', I32), ('kernel', I32 * 3), ('stride', I32 * 3),
                ('dilation', I32 * 3), ('pad_lo', I32 * 3), ('layout', I32), ('dtype', I32),
                ('activation', I32), ('has_bias', I32), ('conj', I32), ('ws_has_kernel', I32), ('kernel_order', I32)]


class DenseDesc(ctypes.Structure):
    _fields_ = [('rows', I32), ('in_q', I32), ('q_units', I32), ('dtype', I32),
                ('activation', I32), ('has_bias', I32), ('ws_has_kernel', I32)]


# every symbol include/qk.h declares: name -> (restype, argtypes)
_VP, _FP, _SZ = ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t
_CD, _DD = ctypes.POINTER(ConvDesc), ctypes.POINTER(DenseDesc)
class PoolDesc(ctypes.Structure):
    """qk_pool_desc_t (include/qk.h)."""
    _fields_ = [('batch', ctypes.c_int32), ('in_h', ctypes.c_int32), ('in_w', ctypes.c_int32),
                ('channels', ctypes.c_int32), ('win_h', ctypes.c_int32), ('win_w', ctypes.c_int32),
                ('out_h', ctypes.c_int32), ('out_w', ctypes.c_int32), ('dtype', ctypes.c_int32)]


_PD = ctypes.POINTER(PoolDesc)


class PostOp(ctypes.Structure):
    """qk_postop_t (include/qk.h): PReLU (+ Dropout) behind a layer."""
    _fields_ = [('alpha_axis', ctypes.c_int32), ('alpha_len', ctypes.c_int32), ('alpha', ctypes.c_void_p),
                ('drop_rate', ctypes.c_float), ('drop_seed', ctypes.c_uint32), ('drop_seed_dev', ctypes.c_void_p)]


_PO = ctypes.POINTER(PostOp)

class ProfRec(ctypes.Structure):
    """qk_prof_rec_t (include/qk.h)"""
    _fields_ = [('op', I32), ('dtype', I32), ('path', I32), ('rows', ctypes.c_int64), ('n', I32), ('k', I32),
                ('ms', ctypes.c_float)]


SYMBOLS = {
    'qk_version': (ctypes.c_int, []),
    'qk_prof_enable': (ctypes.c_int, [ctypes.c_int]),
    'qk_prof_count': (ctypes.c_int, []),
    'qk_prof_get': (ctypes.c_int, [ctypes.c_int, ctypes.POINTER(ProfRec)]),
    'qk_last_error': (ctypes.c_char_p, []),
    'qk_set_debug_flags': (ctypes.c_uint, [ctypes.c_uint]),
    'qk_get_debug_flags': (ctypes.c_uint, []),
    'qk_set_debug_buffer': (None, [_VP, _SZ]),
    'qk_last_path': (ctypes.c_int, []),
    'qk_conv_workspace_bytes': (_SZ, [_CD, ctypes.c_int]),
    'qk_dense_workspace_bytes': (_SZ, [_DD, ctypes.c_int]),
    'qk_conv_fwd': (ctypes.c_int, [_CD, _VP, _FP, _FP, _VP, _VP, _SZ, _VP]),
    'qk_conv_bwd_data': (ctypes.c_int, [_CD, _VP, _VP, _FP, _VP, _VP, _SZ, _VP]),
    'qk_conv_fwd_post': (ctypes.c_int, [_CD, _PO, _VP, _FP, _FP, _VP, _VP, _VP, _SZ, _VP]),
    'qk_conv_bwd_post': (ctypes.c_int, [_CD, _VP, _VP, _FP, _VP, _FP, _FP, _PO, _VP, _FP, I32, _VP, _SZ, _VP]),
    'qk_postop_fwd': (ctypes.c_int, [_CD, _PO, _VP, _VP, _VP]),
    'qk_postop_bwd': (ctypes.c_int, [_CD, _PO, _VP, _VP, _VP, _FP, _VP]),
    'qk_conv_bwd_weight': (ctypes.c_int, [_CD, _VP, _VP, _VP, _FP, _FP, _VP, _SZ, _VP]),
    'qk_conv_bwd_chain': (ctypes.c_int, [_CD, _VP, _VP, _VP, _FP, _VP, _FP, _FP, I32, _VP, _SZ, _VP]),
    'qk_dense_bwd_chain': (ctypes.c_int, [_DD, _VP, _VP, _VP, _FP, _VP, _FP, _FP, I32, _VP, _SZ, _VP]),
    'qk_conv_bwd_weight_acc': (ctypes.c_int, [_CD, _VP, _VP, _VP, _FP, _FP, _VP, _SZ, _VP]),
    'qk_dense_bwd_weight_acc': (ctypes.c_int, [_DD, _VP, _VP, _VP, _FP, _FP, _VP, _SZ, _VP]),
    'qk_conv_bwd': (ctypes.c_int, [_CD, _VP, _VP, _VP, _FP, _VP, _FP, _FP, _VP, _SZ, _VP]),
    'qk_dense_bwd': (ctypes.c_int, [_DD, _VP, _VP, _VP, _FP, _VP, _FP, _FP, _VP, _SZ, _VP]),
    'qk_conv_fold_taps': (ctypes.c_int, [_CD, _VP, _VP, I32, _VP]),
    'qk_conv_relu_pool_aux_bytes': (_SZ, [_CD, I32]),
    'qk_conv_relu_pool_fwd': (ctypes.c_int, [_CD, I32, _VP, _FP, _FP, _VP, _VP, _VP]),
    'qk_conv_relu_pool_bwd': (ctypes.c_int, [_CD, I32, _VP, _VP, _VP, _FP, _FP, I32, _VP]),
    'qk_conv_prelu_pool_fwd': (ctypes.c_int, [_CD, I32, _PO, _VP, _FP, _FP, _VP, _VP, _VP, _VP]),
    'qk_conv_prelu_pool_bwd': (ctypes.c_int, [_CD, I32, _PO, _VP, _VP, _VP, _VP, _FP, _FP, _FP, I32, _VP]),
    'qk_dense_fwd': (ctypes.c_int, [_DD, _VP, _FP, _FP, _VP, _VP, _SZ, _VP]),
    'qk_dense_bwd_data': (ctypes.c_int, [_DD, _VP, _VP, _FP, _VP, _VP, _SZ, _VP]),
    'qk_dense_bwd_weight': (ctypes.c_int, [_DD, _VP, _VP, _VP, _FP, _FP, _VP, _SZ, _VP]),
    'qk_adam_step_zero_grad': (ctypes.c_int, [_FP, _FP, _FP, _FP, _SZ, ctypes.c_float, ctypes.c_float,
                                              ctypes.c_float, ctypes.c_float, I32, ctypes.c_float, _VP]),
    'qk_ctc_workspace_bytes': (_SZ, [I32, I32, I32]),
    'qk_ctc_batch_cost': (ctypes.c_int, [I32, I32, I32, I32, _VP, _VP, I32, _VP, _VP, _FP, _VP, _VP, _SZ, _VP]),
    'qk_conv_prep_kernels': (ctypes.c_int, [I32, ctypes.POINTER(_CD), ctypes.POINTER(I32), ctypes.POINTER(ctypes.c_void_p),
                                            ctypes.POINTER(ctypes.c_void_p), _VP]),
    'qk_adam_step_l2': (ctypes.c_int, [_FP, _FP, _FP, _FP, _FP, _SZ, ctypes.c_float, ctypes.c_float,
                                       ctypes.c_float, ctypes.c_float, I32, ctypes.c_float, I32, _VP]),
    'qk_adam_step_dev': (ctypes.c_int, [_FP, _FP, _FP, _FP, _FP, _SZ, ctypes.c_float, ctypes.c_float,
                                        ctypes.c_float, ctypes.c_float, _VP, ctypes.c_float, I32, _VP]),
    'qk_softmax_rows_fwd': (ctypes.c_int, [I32, ctypes.c_int64, I32, _VP, _VP, _VP, _VP]),
    'qk_softmax_rows_bwd': (ctypes.c_int, [I32, ctypes.c_int64, I32, _VP, _VP, _VP, _VP, _VP]),
    'qk_dense_softmax_supported': (ctypes.c_int, [I32, ctypes.c_int64, I32, I32]),
    'qk_dense_softmax_fwd': (ctypes.c_int, [I32, ctypes.c_int64, I32, I32, _VP, _VP, _VP, _VP, _VP]),
    'qk_dense_softmax_bwd_workspace_bytes': (ctypes.c_size_t, [I32, ctypes.c_int64, I32, I32]),
    'qk_dense_softmax_bwd': (ctypes.c_int, [I32, ctypes.c_int64, I32, I32, _VP, _VP, _VP, _VP, _VP, _VP, _VP, _VP, ctypes.c_float, _VP, _SZ, _VP]),
    'qk_weighted_sum': (ctypes.c_int, [I32, ctypes.c_int64, _VP, _VP, _VP, _VP]),
    'qk_maxpool2d_fwd': (ctypes.c_int, [_PD, _VP, _VP, _VP]),
    'qk_maxpool2d_bwd': (ctypes.c_int, [_PD, _VP, _VP, _VP, _VP]),
    'qk_adam_step': (ctypes.c_int, [_FP, _FP, _FP, _FP, _SZ, ctypes.c_float, ctypes.c_float,
                                    ctypes.c_float, ctypes.c_float, I32, ctypes.c_float, _VP]),
}

_lib = None


def lib():
    """The loaded library; raises RuntimeError (never falls back) when it is not built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                'libqk_hip.so is not built (%s). Run `python -c "import __graft_entry__ as g; '
                'g.build()"` -- the quaternion layers have no CPU or eager fallback.' % LIB_PATH)
        handle = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SYMBOLS.items():
            fn = getattr(handle, name)
            fn.restype, fn.argtypes = res, args
        _lib = handle
    return _lib


def check(rc, what):
    if rc != 0:
        msg = lib().qk_last_error()
        raise RuntimeError('%s failed (status %d): %s' % (what, rc, msg.decode() if msg else ''))


def dbg(bit):
    """True when diagnostic bit `bit` of the library's process-wide mask is set (qk_get_debug_flags: one relaxed atomic load; the
    mask's initial value was read from the environment ONCE, when the library initialised it)."""
    return bool(lib().qk_get_debug_flags() & bit)


class debug_flags(object):
    """`with debug_flags(QK_DBG_NO_MFMA16): ...` -- OR the given diagnostic bits into the library's process-wide
    mask (qk_set_debug_flags) for the duration of the block.  Tests and profiling only."""

    def __init__(self, bits, ablate=0):
        self.bits = int(bits) | (int(ablate) << 8)

    def __enter__(self):
        self.prev = lib().qk_get_debug_flags()
        lib().qk_set_debug_flags(self.prev | self.bits)
        return self

    def __exit__(self, *exc):
        lib().qk_set_debug_flags(self.prev)
        return False


class profile(object):
    """`with profile() as p: step()` then `p.records()`: every forward / backward-data / backward-weight call made
    inside the block (any thread, any stream) with its GEMM view and the milliseconds its launches took on their
    stream (qk_prof_*, HIP events recorded by the library).  records() synchronises."""

    OPS = {QK_OP_FWD: 'fwd', QK_OP_BWD_DATA: 'bwd_data', QK_OP_BWD_WEIGHT: 'bwd_weight'}

    def __enter__(self):
        lib().qk_prof_enable(1)
        return self

    def __exit__(self, *exc):
        lib().qk_prof_enable(0)
        return False

    def records(self):
        out = []
        rec = ProfRec()
        for i in range(lib().qk_prof_count()):
            check(lib().qk_prof_get(i, ctypes.byref(rec)), 'qk_prof_get')
            out.append(dict(op=self.OPS.get(rec.op, str(rec.op)), dtype=rec.dtype, path=QK_PATH_NAMES.get(rec.path, 'none'),
                            rows=int(rec.rows), n=int(rec.n), k=int(rec.k), ms=float(rec.ms)))
        return out


def last_path():
    """Kernel family that served this thread's most recent compute call: 'mfma16', 'mfma16_band', 'fp32_mfma'."""
    return QK_PATH_NAMES.get(lib().qk_last_path(), 'none')
