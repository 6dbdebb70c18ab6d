"""torch.autograd bindings of the HIP Hamilton-product kernels (through the C-ABI only).

`quaternion_conv` / `quaternion_dense` compute what QuaternionConv.call
(complexnn/conv.py:288-345) and QuaternionDense.call (complexnn/dense.py:126-164) of the
reference compute, with the backward TF autodiff would produce -- on MI355X, from the
compact kernel, without materialising the 4x-expanded real kernel.

PyTorch is plumbing here: it owns the device buffers (caching allocator), supplies the
current HIP stream and chains the backward calls.  There is NO CPU or eager fallback:
a CPU tensor or a missing libqk_hip.so raises.
"""
import ctypes
import math

import torch

from . import _lib as L
from ._shape import conv_output_length, normalize_tuple, tf_pads

_DTYPES = {torch.float32: L.QK_F32, torch.bfloat16: L.QK_BF16, torch.float16: L.QK_F16}
_PREP_CACHE_ON = not __import__('os').environ.get('QK_NO_PREP_CACHE')     # diagnostic: re-lay the 16-bit kernels out on every call
_PREP_PARAMS = __import__('weakref').WeakValueDictionary()   # id -> parameter that carries cached 16-bit re-layouts (_Call._ws)
_CAST_PARAMS = __import__('weakref').WeakValueDictionary()   # id -> tensor that carries a cached 16-bit cast (layers._cast_cached, _WeightedSumFn)
_PREP_WARNED = False


def invalidate_cached_kernels(params=None):
    """Drop every cached 16-bit image of the given parameters (default: of all parameters that carry one): the re-laid-out
    kernels of the quaternion layers and the 16-bit casts of the output layer / weighted_sum weights.

    The caches are keyed on the tensor's version counter, its flat buffer's version counter and its data pointer.  Every torch
    op, `FlatParams.adam_step`, `dp.broadcast_params` and `load_state_dict` bump one of them; a RAW write through `.data`
    (`p.data.copy_()`, `p.data.clamp_()`, an EMA / weight-averaging loop over `.data`, a c10d collective on `.data`) bumps none --
    call this afterwards (or use `with torch.no_grad(): p.copy_(...)`, which does bump the counter).  The next call of each
    layer rebuilds its image; returns the number of parameters touched."""
    if params is None:
        params = list(_PREP_PARAMS.values()) + list(_CAST_PARAMS.values())
    n = 0
    for p in params:
        hit = False
        for key in ('_qk_prep', '_qk_cast16', '_qk_cast'):
            if p.__dict__.pop(key, None) is not None:
                hit = True
        n += hit
    return n


def refresh_prepped_kernels(base=None):
    """Bring every cached 16-bit kernel re-layout up to date in ONE launch per device (qk_conv_prep_kernels): called by
    adam_step behind the update, so that the next training step's forward / backward calls find their workspace ready
    (desc.ws_has_kernel = 1) and launch nothing but the GEMM kernel -- 26 small launches per TIMIT step become one.
    base: restrict to the parameters that live in this flat buffer (or are this tensor).  Returns the number of jobs.
    A refresh that fails is not fatal (the optimiser step that called it has already happened): the device's entries
    are dropped from the cache and the next call of each layer re-lays its kernel out itself (ws_has_kernel = 0)."""
    per_dev = {}
    for p in list(_PREP_PARAMS.values()):
        cache = p.__dict__.get('_qk_prep')
        if not cache or not p.is_cuda:
            continue
        pbase = getattr(p, '_qk_flat_base', None)
        if base is not None and pbase is not base and p is not base:
            continue
        ver = (p._version, -1 if pbase is None else pbase._version)
        for key, ent in cache.items():
            if ent[0] == ver or ent[1] != p.data_ptr():
                continue
            per_dev.setdefault(p.device, []).append((p, key, ent, ver))
    total = 0
    for dev, jobs in per_dev.items():
        n = len(jobs)
        dp = (ctypes.POINTER(L.ConvDesc) * n)(*[ctypes.pointer(j[2][3][0]) for j in jobs])
        op_arr = (ctypes.c_int32 * n)(*[j[2][3][1] for j in jobs])
        w_arr = (ctypes.c_void_p * n)(*[j[0].data_ptr() for j in jobs])
        ws_arr = (ctypes.c_void_p * n)(*[j[2][2].data_ptr() for j in jobs])
        with torch.cuda.device(dev):
            rc = L.lib().qk_conv_prep_kernels(n, dp, op_arr, w_arr, ws_arr, _raw_stream(dev.index) if _raw_stream is not None
                                              else torch.cuda.current_stream(dev).cuda_stream)
        if rc != 0:
            # non-fatal (the optimiser step has already happened) but never silent: a sticky HIP fault or an argument bug would
            # otherwise show up only as 26 extra launches per step, or as a later unrelated failure
            msg = L.lib().qk_last_error()
            msg = msg.decode() if isinstance(msg, bytes) else str(msg)
            for p, key, ent, ver in jobs:
                p.__dict__.get('_qk_prep', {}).pop(key, None)
            if rc == L.QK_ERR_LAUNCH:
                raise RuntimeError('qk_conv_prep_kernels failed on %s: %s' % (dev, msg))
            global _PREP_WARNED
            if not _PREP_WARNED:
                _PREP_WARNED = True
                __import__('warnings').warn('qk_conv_prep_kernels refused the batched 16-bit kernel refresh on %s (rc %d: %s); the layers '
                                            're-lay their kernels out per call from now on' % (dev, rc, msg), RuntimeWarning)
            continue
        for p, key, ent, ver in jobs:
            ent[0] = ver
        total += n
    return total


def _require_device(t, what):
    if not t.is_cuda:
        raise RuntimeError('%s: got a CPU tensor. The quaternion layers run only on the MI355X HIP '
                           'path (libqk_hip.so); there is no CPU fallback.' % what)
    if t.dtype not in _DTYPES:
        raise TypeError('%s: unsupported dtype %s (float32, bfloat16, float16)' % (what, t.dtype))


def _ptr(t):
    return t.data_ptr() if t is not None else None      # argtypes are c_void_p: ints convert


# Host cost matters: the config-2 layer step is ~150 us of GPU time over 7 launches, so every
# microsecond of Python per launch shows up once a collective joins the step.
_raw_stream = getattr(torch._C, '_cuda_getCurrentRawStream', None)


def _stream(t):
    if _raw_stream is not None:
        return _raw_stream(t.device.index)
    return torch.cuda.current_stream(t.device).cuda_stream


class _on_device(object):
    """torch.cuda.device(d) guard that is skipped when d is already the current device (the usual case)."""
    __slots__ = ('guard',)

    def __init__(self, dev):
        self.guard = None if torch.cuda.current_device() == dev.index else torch.cuda.device(dev)

    def __enter__(self):
        if self.guard is not None:
            self.guard.__enter__()

    def __exit__(self, *exc):
        if self.guard is not None:
            self.guard.__exit__(*exc)
        return False


class _Call(object):
    """One layer invocation: descriptor + the three C-ABI entry points it uses."""

    def __init__(self, desc, names, ws_fn, x_shape, y_shape, w_shape, relu):
        self.desc, self.names, self.ws_fn = desc, names, ws_fn
        self.x_shape, self.y_shape, self.w_shape, self.relu = x_shape, y_shape, w_shape, relu
        self.static_buffers = False      # True: keep workspaces (graph capture / bench loops)
        self._ws_cache = {}
        self._ws_bytes = {}              # the descriptor never changes after construction

    def _prep_job(self, op):
        """(conv descriptor, operation) that reproduces this call's kernel re-layout through qk_conv_prep_kernels."""
        d = self.desc
        if isinstance(d, L.DenseDesc):
            c = L.ConvDesc()
            c.rank, c.batch, c.cq, c.fq, c.dtype, c.conj, c.layout = 0, d.rows, d.in_q, d.q_units, d.dtype, 1, L.QK_CH_LAST
            for i in range(3):
                c.in_spatial[i] = c.out_spatial[i] = c.kernel[i] = c.stride[i] = c.dilation[i] = 1
        else:
            c = L.ConvDesc.from_buffer_copy(d)
        return c, (L.QK_OP_FWD if op == L.QK_OP_FWD else L.QK_OP_BWD_DATA)

    def _kernel_only_bytes(self, op):
        """Size of operation `op`'s workspace when it holds NOTHING but the re-laid-out 16-bit kernel (band layout + zero line,
        and for 16 / 32-channel layers the small-channel kernel's fragment layout behind it); 0 for fp32 and for
        channels_first descriptors (their workspace also carries re-laid-out operands).  The library is asked, never a formula
        restated here: the forward / backward-data figure IS the kernel-only size for such descriptors."""
        d = self.desc
        if d.dtype == L.QK_F32 or getattr(d, 'layout', L.QK_CH_LAST) != L.QK_CH_LAST:
            return 0
        cq, fq = (d.in_q, d.q_units) if isinstance(d, L.DenseDesc) else (d.cq, d.fq)
        if cq % 16 or fq % 16:
            return 0            # off the matrix-core path: the fp32-MFMA kernels never read the workspace -- nothing to cache or refresh
        kop = L.QK_OP_FWD if op == L.QK_OP_FWD else L.QK_OP_BWD_DATA
        n = self._ws_bytes.get(('k', kop))
        if n is None:
            n = self._ws_bytes[('k', kop)] = int(getattr(L.lib(), self.ws_fn)(ctypes.byref(self.desc), kop))
        return n

    def _ws(self, op, like, wparam=None):
        """Workspace of operation `op`.  `wparam`: the PARAMETER the kernel argument is (a long-lived leaf tensor).  When the
        workspace holds only the kernel's 16-bit re-layout, it is kept ON the parameter together with the tensor version it
        was made from, and handed back with desc.ws_has_kernel = 1 as long as the weights have not changed -- the C side then
        skips the re-layout launch (26 launches per TIMIT training step).  Every in-place update bumps the version
        (torch ops do it themselves, adam_step through torch.autograd.graph.increment_version); a re-homed `.data` changes
        the pointer; a parameter re-homed into a dp.FlatParams buffer is also checked against THAT buffer's version (writes through
        the flat buffer -- the fused Adam, a broadcast -- do not touch the views' own counters); the cache dies with the parameter."""
        n = self._ws_bytes.get(op)
        if n is None:
            n = self._ws_bytes[op] = int(getattr(L.lib(), self.ws_fn)(ctypes.byref(self.desc), op))
        self.desc.ws_has_kernel = 0
        if n == 0:
            return None, 0
        if (wparam is not None and not self.static_buffers and n == self._kernel_only_bytes(op) and wparam.is_leaf
                and wparam.requires_grad and _PREP_CACHE_ON):
            key = ('f' if op == L.QK_OP_FWD else 't', int(self.desc.conj) if hasattr(self.desc, 'conj') else 1, int(self.desc.dtype))
            cache = wparam.__dict__.setdefault('_qk_prep', {})
            base = getattr(wparam, '_qk_flat_base', None)         # dp.FlatParams: the flat buffer this parameter is a view of --
            ver = (wparam._version, -1 if base is None else base._version)    # `.data` views do not share its version counter
            hit = cache.get(key)
            if hit is not None and hit[0] == ver and hit[1] == wparam.data_ptr() and hit[2].numel() == n \
                    and hit[2].device == like.device:
                self.desc.ws_has_kernel = 1
                return hit[2], n
            buf = torch.empty(n, dtype=torch.uint8, device=like.device)
            cache[key] = [ver, wparam.data_ptr(), buf, self._prep_job(op)]
            _PREP_PARAMS[id(wparam)] = wparam
            return buf, n
        if self.static_buffers:
            buf = self._ws_cache.get(op)
            if buf is None:
                buf = self._ws_cache[op] = torch.empty(n, dtype=torch.uint8, device=like.device)
            return buf, n
        return torch.empty(n, dtype=torch.uint8, device=like.device), n

    def fwd(self, x, w, bias, out=None, wparam=None):
        y = out if out is not None else torch.empty(self.y_shape, dtype=x.dtype, device=x.device)
        ws, n = self._ws(L.QK_OP_FWD, x, wparam)
        with _on_device(x.device):
            rc = getattr(L.lib(), self.names[0])(ctypes.byref(self.desc), _ptr(x), _ptr(w), _ptr(bias),
                                                 _ptr(y), _ptr(ws), n, _stream(x))
        L.check(rc, self.names[0])
        return y

    def bwd_data(self, dy, y, w, out=None, wparam=None):
        dx = out if out is not None else torch.empty(self.x_shape, dtype=dy.dtype, device=dy.device)
        ws, n = self._ws(L.QK_OP_BWD_DATA, dy, wparam)
        with _on_device(dy.device):
            rc = getattr(L.lib(), self.names[1])(ctypes.byref(self.desc), _ptr(dy), _ptr(y), _ptr(w),
                                                 _ptr(dx), _ptr(ws), n, _stream(dy))
        L.check(rc, self.names[1])
        return dx

    def bwd_weight(self, x, dy, y, has_bias, out=None, masked_dy_out=None, accumulate=False):
        """dw, db.  `masked_dy_out` (a tensor like dy) receives dy*(y>0) for RELU layers.
        accumulate=True adds into `out` instead of overwriting it (qk_*_bwd_weight_acc)."""
        if out is not None:
            dw, db = out
        else:
            dw = torch.empty(self.w_shape, dtype=torch.float32, device=x.device)
            db = torch.empty((self.w_shape[-1],), dtype=torch.float32, device=x.device) if has_bias else None
        ws, n = self._ws(L.QK_OP_BWD_WEIGHT, x)
        if masked_dy_out is not None:
            ws, n = masked_dy_out, masked_dy_out.numel() * masked_dy_out.element_size()
        with _on_device(x.device):
            name = self.names[2] + ('_acc' if accumulate else '')
            if accumulate and out is None:
                raise ValueError('accumulate=True needs the buffers to add into (out=)')
            rc = getattr(L.lib(), name)(ctypes.byref(self.desc), _ptr(x), _ptr(dy), _ptr(y),
                                        _ptr(dw), _ptr(db), _ptr(ws), n, _stream(x))
        L.check(rc, name)
        return dw, db


    def fwd_post(self, x, w, bias, post, wparam=None):
        """(pre, y) = qk_conv_fwd_post: the LINEAR convolution and post(pre) (PReLU / dropout) from one launch.  The relu
        form of the post-op (post.alpha is None) writes y only: pre is None."""
        pre = torch.empty(self.y_shape, dtype=x.dtype, device=x.device) if post.alpha is not None else None
        y = torch.empty(self.y_shape, dtype=x.dtype, device=x.device)
        ws, n = self._ws(L.QK_OP_FWD, x, wparam)
        with _on_device(x.device):
            rc = L.lib().qk_conv_fwd_post(ctypes.byref(self.desc), ctypes.byref(post.struct), _ptr(x), _ptr(w), _ptr(bias),
                                          _ptr(pre), _ptr(y), _ptr(ws), n, _stream(x))
        L.check(rc, 'qk_conv_fwd_post')
        return pre, y

    def bwd_post(self, x, dy, w, has_bias, post_x, x_pre, dalpha_x, direct=None, wparam=None):
        """Fused backward of a LINEAR layer whose input x = post_x(x_pre): returns (d x_pre, dw, db) and accumulates
        the slope gradient of post_x into dalpha_x (qk_conv_bwd_post).  direct = (dw, db) buffers to ADD into
        (QK_BWD_ACCUMULATE; the returned dw / db are then None).  Relu form of post_x: x_pre / dalpha_x are None."""
        dx = torch.empty(self.x_shape, dtype=dy.dtype, device=dy.device)
        if direct is not None:
            dw, db = direct
        else:
            dw = torch.empty(self.w_shape, dtype=torch.float32, device=x.device)
            db = torch.empty((self.w_shape[-1],), dtype=torch.float32, device=x.device) if has_bias else None
        ws, n = self._ws(L.QK_OP_BWD, x, wparam)
        with _on_device(x.device):
            rc = L.lib().qk_conv_bwd_post(ctypes.byref(self.desc), _ptr(x), _ptr(dy), _ptr(w), _ptr(dx), _ptr(dw), _ptr(db),
                                          ctypes.byref(post_x.struct), _ptr(x_pre), _ptr(dalpha_x),
                                          L.QK_BWD_ACCUMULATE if direct is not None else 0, _ptr(ws), n, _stream(x))
        L.check(rc, 'qk_conv_bwd_post')
        return (dx, None, None) if direct is not None else (dx, dw, db)

    def bwd(self, x, dy, y, w, has_bias, out=None, flags=0, wparam=None):
        """Fused backward (qk_*_bwd, or qk_*_bwd_chain with L.QK_BWD_* flags): returns (dx, dw, db)."""
        if out is not None:
            dx, dw, db = out
        else:
            dx = torch.empty(self.x_shape, dtype=dy.dtype, device=dy.device)
            dw = torch.empty(self.w_shape, dtype=torch.float32, device=x.device)
            db = torch.empty((self.w_shape[-1],), dtype=torch.float32, device=x.device) if has_bias else None
        ws, n = self._ws(L.QK_OP_BWD, x, wparam)
        name = self.names[1].replace('_bwd_data', '_bwd')
        with _on_device(x.device):
            if flags:
                name += '_chain'
                rc = getattr(L.lib(), name)(ctypes.byref(self.desc), _ptr(x), _ptr(dy), _ptr(y), _ptr(w), _ptr(dx),
                                            _ptr(dw), _ptr(db), int(flags), _ptr(ws), n, _stream(x))
            else:
                rc = getattr(L.lib(), name)(ctypes.byref(self.desc), _ptr(x), _ptr(dy), _ptr(y), _ptr(w), _ptr(dx),
                                            _ptr(dw), _ptr(db), _ptr(ws), n, _stream(x))
        L.check(rc, name)
        return dx, dw, db


class PostOp(object):
    """PReLU (+ Dropout) behind a quaternion layer (include/qk.h: qk_postop_t).  `alpha`: float32 device tensor, one
    slope (alpha_axis = -1) or one per position along spatial axis `alpha_axis` of the channels_last activation;
    `rate`: dropout rate (0 = off); `seed`: 32-bit seed of the counter-based mask (a new one every step);
    `seed_dev`: optional one-element int32 DEVICE tensor whose current value the kernels mix into the seed -- the step counter of
    `adam_step(step=<that tensor>)`: the launch arguments are then the same every step (graph replay) and the masks still change.
    alpha=None is the relu form y = dropout(relu(pre)): one output tensor, the backward reads only y.
    The rate the kernels apply is round(rate * 256) / 256 (8 random bits per element): `applied_rate`."""

    def __init__(self, alpha, alpha_axis=-1, rate=0.0, seed=0, seed_dev=None):
        if alpha is not None and (alpha.dtype != torch.float32 or not alpha.is_cuda):
            raise TypeError('PReLU slopes must be float32 device tensors')
        if seed_dev is not None and (seed_dev.dtype != torch.int32 or not seed_dev.is_cuda or seed_dev.numel() != 1):
            raise TypeError('seed_dev must be a one-element int32 device tensor (the training step counter of adam_step)')
        self.seed_dev = seed_dev                      # kept alive: the struct below holds its address
        if not 0.0 <= float(rate) < 1.0:
            raise ValueError('dropout rate must be in [0, 1), got %r' % (rate,))
        self.alpha, self.alpha_axis, self.rate, self.seed = alpha, int(alpha_axis), float(rate), int(seed) & 0xffffffff
        self.applied_rate = min(255, int(self.rate * 256.0 + 0.5)) / 256.0
        self.flat = alpha.detach().reshape(-1).contiguous() if alpha is not None else None
        self.struct = L.PostOp(self.alpha_axis if alpha is not None else -1, self.flat.numel() if alpha is not None else 0,
                               self.flat.data_ptr() if alpha is not None else None, self.rate, self.seed,
                               seed_dev.data_ptr() if seed_dev is not None else None)


def _tensor_desc(t):
    """ConvDesc fields the post-op entry points read, for a channels_last (N, *spatial, 4F) or (M, 4F) tensor."""
    d = L.ConvDesc()
    d.rank, d.batch, d.fq, d.dtype = t.dim() - 2, t.shape[0], t.shape[-1] // 4, _DTYPES[t.dtype]
    for i in range(3):
        d.out_spatial[i] = t.shape[1 + i] if i < t.dim() - 2 else 1
    return d


def postop_fwd(pre, post):
    y = torch.empty_like(pre)
    d = _tensor_desc(pre)
    with _on_device(pre.device):
        rc = L.lib().qk_postop_fwd(ctypes.byref(d), ctypes.byref(post.struct), _ptr(pre), _ptr(y), _stream(pre))
    L.check(rc, 'qk_postop_fwd')
    return y


def postop_bwd(pre, dy, post, dalpha):
    """d pre (returned) and the slope gradient (accumulated into the float32 buffer `dalpha`).  Relu form of the
    post-op: pass the forward OUTPUT y as `pre`, dalpha=None."""
    dpre = torch.empty_like(pre)
    d = _tensor_desc(pre)
    with _on_device(pre.device):
        rc = L.lib().qk_postop_bwd(ctypes.byref(d), ctypes.byref(post.struct), _ptr(pre), _ptr(dy), _ptr(dpre), _ptr(dalpha),
                                   _stream(pre))
    L.check(rc, 'qk_postop_bwd')
    return dpre


class _PostOpFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pre, alpha, post):
        ctx.post = post
        y = postop_fwd(pre, post)
        ctx.save_for_backward(pre if post.alpha is not None else y)      # relu form: the output is its own mask
        return y

    @staticmethod
    def backward(ctx, dy):
        pre, = ctx.saved_tensors
        post = ctx.post
        if post.alpha is None:
            return postop_bwd(pre, dy.contiguous(), post, None), None, None
        dalpha = torch.zeros(post.flat.numel(), dtype=torch.float32, device=pre.device)
        dpre = postop_bwd(pre, dy.contiguous(), post, dalpha)
        return dpre, dalpha.reshape(post.alpha.shape), None


def prelu_dropout(x, alpha, alpha_axis=-1, rate=0.0, seed=0):
    """y = dropout(prelu(x)) in ONE pass over a contiguous channels_last (N, *spatial, C) / (M, C) device tensor
    (keras PReLU + Dropout, interspeech_model.py:99-101,117-121); alpha: one slope or one per position along spatial
    axis `alpha_axis`.  The mask is a hash of (seed, element index): nothing is stored, the backward regenerates it."""
    _require_device(x, 'prelu_dropout')
    xc = x.contiguous()
    if xc.shape[-1] % (4 if xc.dtype == torch.float32 else 8) or xc.dim() < 2 or xc.dim() > 5:
        raise ValueError('prelu_dropout: unsupported shape %s' % (tuple(x.shape),))
    return _PostOpFn.apply(xc, alpha, PostOp(alpha, alpha_axis, rate, seed))


def relu_dropout(x, rate=0.0, seed=0):
    """y = dropout(relu(x)) in one pass (the relu form of prelu_dropout: no slopes; the backward reads y only)."""
    return prelu_dropout(x, None, -1, rate, seed)


def _direct_grad(w, b, want_w, want_b):
    """(dw, db) buffers to ACCUMULATE into, or None.  Parameters re-homed by dp.FlatParams carry `.grad` views of one
    flat, zeroed-after-the-step gradient buffer: the backward kernels add into them directly (qk_*_bwd_weight_acc /
    QK_BWD_ACCUMULATE) instead of filling a temporary that autograd then adds -- one memset and one elementwise pass
    less per parameter and step.  The caller returns None as these parameters' gradients and calls _grad_ready."""
    if not (want_w and getattr(w, '_qk_direct_grad', False) and w.grad is not None and w.grad.is_contiguous()
            and w.grad.dtype == torch.float32):
        return None
    if b is None:
        return w.grad, None
    if not (want_b and getattr(b, '_qk_direct_grad', False) and b.grad is not None and b.grad.is_contiguous()
            and b.grad.dtype == torch.float32):
        return None
    return w.grad, b.grad


def _grad_ready(*params):
    for p in params:
        cb = getattr(p, '_qk_grad_ready', None) if p is not None else None
        if cb is not None:
            cb(p)                         # dp.BucketedAllReduce counts its bucket down (autograd's hook will not fire)


class _HamiltonFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, bias, call):
        y = call.fwd(x, w, bias, wparam=w)
        ctx.call = call
        ctx.has_bias = bias is not None
        ctx.params = (w, bias)
        ctx.save_for_backward(x, w, y if call.relu else None)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w, y = ctx.saved_tensors
        call = ctx.call
        dy = dy.contiguous()
        dx = dw = db = None
        want_w = ctx.needs_input_grad[1] or (ctx.has_bias and ctx.needs_input_grad[2])
        direct = _direct_grad(ctx.params[0], ctx.params[1], ctx.needs_input_grad[1], ctx.has_bias and ctx.needs_input_grad[2])
        if ctx.needs_input_grad[0] and want_w:
            if direct is not None:
                dx = torch.empty(call.x_shape, dtype=dy.dtype, device=dy.device)
                call.bwd(x, dy, y, w, ctx.has_bias, out=(dx,) + direct, flags=L.QK_BWD_ACCUMULATE, wparam=ctx.params[0])
            else:
                dx, dw, db = call.bwd(x, dy, y, w, ctx.has_bias, wparam=ctx.params[0])        # fused: one pass over (dy, y)
        elif ctx.needs_input_grad[0]:
            dx = call.bwd_data(dy, y, w, wparam=ctx.params[0])
        elif want_w:
            if direct is not None:
                call.bwd_weight(x, dy, y, ctx.has_bias, out=direct, accumulate=True)
            else:
                dw, db = call.bwd_weight(x, dy, y, ctx.has_bias)
        if direct is not None and want_w:
            _grad_ready(*ctx.params)
        return dx, dw, db, None


def _act_code(activation):
    if activation in (None, 'linear'):
        return L.QK_ACT_LINEAR
    if activation == 'relu':
        return L.QK_ACT_RELU
    raise ValueError('fused activation must be None/"linear"/"relu", got %r' % (activation,))


def _check_weights(w, bias, n_out):
    if w.dtype != torch.float32 or (bias is not None and bias.dtype != torch.float32):
        raise TypeError('kernel / bias must be float32 (Keras floatx); activations may be 16-bit')
    if bias is not None and tuple(bias.shape) != (n_out,):
        raise ValueError('bias must have shape (%d,), got %s' % (n_out, tuple(bias.shape)))


def conv_call(x_shape, w_shape, dtype, rank, strides=1, padding='valid', layout='channels_last',
              dilation_rate=1, activation=None, use_bias=True, conj=False, kernel_order=0):
    """Descriptor for one conv call on a PHYSICAL layout (`layout` describes the buffer).  kernel_order =
    L.QK_KERNEL_CHANNEL_MAJOR: the kernel buffer lies as (cq, *kernel_size, 4 fq) -- a QuaternionDense weight read in place as
    the kernel of the equivalent convolution (include/qk.h); w_shape stays the logical (*kernel_size, cq, 4 fq)."""
    st = normalize_tuple(strides, rank, 'strides')
    dl = normalize_tuple(dilation_rate, rank, 'dilation_rate')
    if len(x_shape) != rank + 2 or len(w_shape) != rank + 2:
        raise ValueError('rank %d conv needs %d-D input and kernel' % (rank, rank + 2))
    if layout == 'channels_first':
        ci, sp = x_shape[1], tuple(x_shape[2:])
    else:
        ci, sp = x_shape[-1], tuple(x_shape[1:-1])
    cq, fq = w_shape[-2], w_shape[-1] // 4
    if ci != 4 * cq or w_shape[-1] != 4 * fq:
        raise ValueError('input channels %d / kernel shape %s are not quaternion-consistent'
                         % (ci, tuple(w_shape)))
    d = L.ConvDesc()
    d.rank, d.batch, d.cq, d.fq = rank, x_shape[0], cq, fq
    out_sp = []
    for i in range(3):
        if i < rank:
            k = w_shape[i]
            d.in_spatial[i], d.kernel[i], d.stride[i], d.dilation[i] = sp[i], k, st[i], dl[i]
            d.pad_lo[i] = tf_pads(sp[i], k, st[i], dl[i], padding)[0]
            d.out_spatial[i] = conv_output_length(sp[i], k, padding, st[i], dl[i])
            if d.out_spatial[i] <= 0:
                raise ValueError('convolution output would be empty on axis %d' % i)
            out_sp.append(d.out_spatial[i])
        else:
            d.in_spatial[i] = d.out_spatial[i] = d.kernel[i] = d.stride[i] = d.dilation[i] = 1
            d.pad_lo[i] = 0
    d.layout = L.QK_CH_FIRST if layout == 'channels_first' else L.QK_CH_LAST
    d.dtype = _DTYPES[dtype]
    d.activation = _act_code(activation)
    d.has_bias = int(bool(use_bias))
    d.conj = int(bool(conj))
    if layout == 'channels_first':
        y_shape = (x_shape[0], 4 * fq) + tuple(out_sp)
    else:
        y_shape = (x_shape[0],) + tuple(out_sp) + (4 * fq,)
    d.kernel_order = int(kernel_order)
    return _Call(d, ('qk_conv_fwd', 'qk_conv_bwd_data', 'qk_conv_bwd_weight'),
                 'qk_conv_workspace_bytes', tuple(x_shape), y_shape, tuple(w_shape),
                 d.activation == L.QK_ACT_RELU)


def dense_call(x_shape, w_shape, dtype, activation=None, use_bias=True):
    if len(x_shape) != 2 or len(w_shape) != 2:
        raise ValueError('quaternion dense needs 2-D input and kernel')
    in_q, units = w_shape
    if x_shape[1] != 4 * in_q or units % 4:
        raise ValueError('input width %d / kernel shape %s are not quaternion-consistent'
                         % (x_shape[1], tuple(w_shape)))
    d = L.DenseDesc()
    d.rows, d.in_q, d.q_units = x_shape[0], in_q, units // 4
    d.dtype = _DTYPES[dtype]
    d.activation = _act_code(activation)
    d.has_bias = int(bool(use_bias))
    return _Call(d, ('qk_dense_fwd', 'qk_dense_bwd_data', 'qk_dense_bwd_weight'),
                 'qk_dense_workspace_bytes', tuple(x_shape), (x_shape[0], units), tuple(w_shape),
                 d.activation == L.QK_ACT_RELU)


def quaternion_conv(x, kernel, bias=None, strides=1, padding='valid', data_format='channels_last',
                    dilation_rate=1, activation=None, conj=False, internal_layout='channels_last',
                    fold_small_cq=True, post=None):
    """y = act(W (x) x + b): Hamilton-product convolution of rank kernel.dim()-2.

    x       (N, *spatial, 4Cq) or (N, 4Cq, *spatial) -- component-planar channels (r|i|j|k)
    kernel  (*kernel_size, Cq, 4F) float32 compact kernel (conv.py:165, init.py:91)
    For data_format='channels_first' and internal_layout='channels_last' (default) the data is
    kept PHYSICALLY channels-last (a strided view with the logical channels_first shape is
    returned, torch.channels_last style): MFMA operands want the reduction axis contiguous.
    internal_layout='native' runs the channels_first buffers as they are.
    fold_small_cq: layers with 1-2 quaternion input channels whose input needs no gradient (the first
    layer of a network) are run as a 1x1 convolution on a tap-folded copy of x (qk_conv_fold_taps).
    post: dict(alpha=, alpha_axis=, rate=, seed=) -- PReLU (+ dropout) behind a LINEAR layer, fused into the
    kernel epilogues (see quaternion_conv_chain); alpha_axis counts the spatial axes of the layer output.
    """
    _require_device(x, 'quaternion_conv')
    rank = kernel.dim() - 2
    _check_weights(kernel, bias, kernel.shape[-1])
    ch_first = data_format == 'channels_first'
    taps, cq = int(math.prod(kernel.shape[:rank])), kernel.shape[-2]
    if x.shape[0] == 0:
        return _empty_batch(x, kernel, bias, rank, strides, padding, data_format, dilation_rate)
    if post is not None and internal_layout != 'channels_last':
        raise ValueError('a post-op needs internal_layout="channels_last"')
    if fold_small_cq and cq <= 2 and taps > 1 and taps * cq <= 64 and not x.requires_grad and not conj:
        return _folded_conv(x, kernel, bias, rank, strides, padding, data_format, dilation_rate, activation,
                            internal_layout, post)
    if post is not None:
        xl = x.movedim(1, -1) if ch_first else x
        y = quaternion_conv_chain(xl, [(kernel, bias, dict(strides=strides, padding=padding, dilation_rate=dilation_rate,
                                                           activation=activation, conj=conj, post=post))])
        return y.movedim(-1, 1) if ch_first else y
    if ch_first and internal_layout == 'channels_last':
        xp = x.movedim(1, -1).contiguous()
        layout = 'channels_last'
    else:
        xp = x.contiguous()
        layout = data_format
    call = conv_call(tuple(xp.shape), tuple(kernel.shape), xp.dtype, rank, strides, padding, layout,
                     dilation_rate, activation, bias is not None, conj)
    y = _HamiltonFn.apply(xp, kernel.contiguous(), bias, call)
    if ch_first and internal_layout == 'channels_last':
        y = y.movedim(-1, 1)
    return y


def _empty_batch(x, kernel, bias, rank, strides, padding, data_format, dilation_rate):
    """Zero samples: Keras returns an empty tensor of the right shape (and zero gradients); there is
    nothing to launch."""
    ch_first = data_format == 'channels_first'
    ks = tuple(kernel.shape[:rank])
    st, dl = normalize_tuple(strides, rank, 'strides'), normalize_tuple(dilation_rate, rank, 'dilation_rate')
    sp = x.shape[2:] if ch_first else x.shape[1:-1]
    out = tuple(conv_output_length(sp[i], ks[i], padding, st[i], dl[i]) for i in range(rank))
    shape = (0, kernel.shape[-1]) + out if ch_first else (0,) + out + (kernel.shape[-1],)
    y = torch.zeros(shape, dtype=x.dtype, device=x.device)
    tie = x.sum() * 0 + (kernel.sum() * 0).to(x.dtype)          # keeps autograd connected: zero gradients
    if bias is not None:
        tie = tie + (bias.sum() * 0).to(x.dtype)
    return y + tie


def _folded_conv(x, kernel, bias, rank, strides, padding, data_format, dilation_rate, activation, internal_layout,
                 post=None):
    ch_first = data_format == 'channels_first'
    xp = x.contiguous()
    taps, cq = int(math.prod(kernel.shape[:rank])), kernel.shape[-2]
    # folded channel count: a multiple of 8 (fp32 kernels) / of 32 (the 16-bit MFMA path) that holds every
    # (tap, channel) pair -- 32 for the 15-tap TIMIT layer, 64 for e.g. a 7x7 kernel on one channel
    gran = 32 if x.dtype != torch.float32 else 8
    cq2 = (taps * cq + gran - 1) // gran * gran
    call = conv_call(tuple(xp.shape), tuple(kernel.shape), xp.dtype, rank, strides, padding, data_format,
                     dilation_rate, None, False, False)
    d = call.desc
    out_sp = tuple(d.out_spatial[i] for i in range(rank))
    xcol = torch.empty((xp.shape[0],) + out_sp + (4 * cq2,), dtype=xp.dtype, device=xp.device)
    with _on_device(xp.device):
        rc = L.lib().qk_conv_fold_taps(ctypes.byref(d), _ptr(xp), _ptr(xcol), cq2, _stream(xp))
    L.check(rc, 'qk_conv_fold_taps')
    w2 = torch.nn.functional.pad(kernel.reshape(taps * cq, kernel.shape[-1]), (0, 0, 0, cq2 - taps * cq))
    w2 = w2.reshape((1,) * rank + (cq2, kernel.shape[-1]))
    y = quaternion_conv(xcol, w2, bias, 1, 'valid', 'channels_last', 1, activation, False, 'channels_last', False, post)
    return y.movedim(-1, 1) if ch_first else y


def quaternion_dense(x, kernel, bias=None, activation=None):
    """y = act(conj(W) (x) x + b) -- the table dense.py:139-143 builds (transpose of conv's)."""
    _require_device(x, 'quaternion_dense')
    _check_weights(kernel, bias, kernel.shape[-1])
    if x.shape[0] == 0:
        tie = x.sum() * 0 + (kernel.sum() * 0).to(x.dtype) + ((bias.sum() * 0).to(x.dtype) if bias is not None else 0)
        return torch.zeros((0, kernel.shape[-1]), dtype=x.dtype, device=x.device) + tie
    xp = x.contiguous()
    call = dense_call(tuple(xp.shape), tuple(kernel.shape), xp.dtype, activation, bias is not None)
    return _HamiltonFn.apply(xp, kernel.contiguous(), bias, call)


def adam_step(param, grad, m, v, step, lr=0.001, beta1=0.9, beta2=0.999, eps=1e-7, grad_scale=1.0, zero_grad=False,
              decay=None):
    """Fused Keras-Adam update of a flat float32 buffer (qk_adam_step); zero_grad=True also clears `grad`
    once it has been consumed (qk_adam_step_zero_grad), ready for accumulating backward calls.
    decay: per-element coefficients of the l2 kernel regularisers (dp.FlatParams.l2_decay): g += decay * param inside
    the kernel (qk_adam_step_l2) -- the gradient of the term Keras adds to the loss."""
    for t in (param, grad, m, v) + ((decay,) if decay is not None else ()):
        if not t.is_cuda or t.dtype != torch.float32 or not t.is_contiguous():
            raise RuntimeError('adam_step needs contiguous float32 device buffers')
    n = param.numel()
    if isinstance(step, torch.Tensor):
        # the step number lives on the device (qk_adam_step_dev): `step` holds the number of steps applied so far and is
        # incremented by the call -- no argument of the launch depends on the step (a captured graph replays correctly)
        if step.dtype != torch.int32 or not step.is_cuda or step.numel() != 1:
            raise TypeError('adam_step: a device-side step counter must be a one-element int32 device tensor')
        if decay is not None and decay.numel() != n:
            raise ValueError('decay must have one coefficient per parameter element')
        with _on_device(param.device):
            rc = L.lib().qk_adam_step_dev(_ptr(param), _ptr(grad), _ptr(m), _ptr(v), _ptr(decay), n, lr, beta1, beta2, eps,
                                          _ptr(step), grad_scale, int(bool(zero_grad)), _stream(param))
        L.check(rc, 'qk_adam_step_dev')
        torch.autograd.graph.increment_version(param)
        if _PREP_CACHE_ON:
            refresh_prepped_kernels(param)
        return
    with _on_device(param.device):
        if decay is not None:
            if decay.numel() != n:
                raise ValueError('decay must have one coefficient per parameter element')
            rc = L.lib().qk_adam_step_l2(_ptr(param), _ptr(grad), _ptr(m), _ptr(v), _ptr(decay), n, lr, beta1, beta2, eps,
                                         int(step), grad_scale, int(bool(zero_grad)), _stream(param))
        else:
            fn = L.lib().qk_adam_step_zero_grad if zero_grad else L.lib().qk_adam_step
            rc = fn(_ptr(param), _ptr(grad), _ptr(m), _ptr(v), n, lr, beta1, beta2, eps, int(step), grad_scale, _stream(param))
    L.check(rc, 'qk_adam_step')
    # the kernel wrote `param` behind torch's back: move its version counter (every view of a flat buffer shares it), so that
    # cached 16-bit re-layouts of the kernels (_Call._ws) are seen as stale
    torch.autograd.graph.increment_version(param)
    if _PREP_CACHE_ON:
        refresh_prepped_kernels(param)


class _MaxPoolCL(torch.autograd.Function):
    """qk_maxpool2d_fwd / _bwd on a channels_last buffer (N, H, W, C); windows == strides."""

    @staticmethod
    def forward(ctx, x, win, out_hw):
        n, h, w, c = x.shape
        d = L.PoolDesc(n, h, w, c, win[0], win[1], out_hw[0], out_hw[1], _DTYPES[x.dtype])
        y = torch.empty((n, out_hw[0], out_hw[1], c), dtype=x.dtype, device=x.device)
        with _on_device(x.device):
            rc = L.lib().qk_maxpool2d_fwd(ctypes.byref(d), _ptr(x), _ptr(y), _stream(x))
        L.check(rc, 'qk_maxpool2d_fwd')
        ctx.desc = d
        ctx.save_for_backward(x)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, = ctx.saved_tensors
        dy = dy.contiguous()
        dx = torch.empty_like(x)
        with _on_device(x.device):
            rc = L.lib().qk_maxpool2d_bwd(ctypes.byref(ctx.desc), _ptr(x), _ptr(dy), _ptr(dx), _stream(x))
        L.check(rc, 'qk_maxpool2d_bwd')
        return dx, None, None


def maxpool2d_channels_last(x, window, out_hw):
    """Max pooling of a contiguous (N, H, W, C) device tensor with non-overlapping windows and
    high-side-only padding (see include/qk.h); returns (N, out_h, out_w, C)."""
    _require_device(x, 'maxpool2d_channels_last')
    return _MaxPoolCL.apply(x, tuple(window), tuple(out_hw))


def maxpool2d_supported(x, window, strides):
    vec = 4 if x.dtype == torch.float32 else 8
    return (x.is_cuda and x.dtype in _DTYPES and x.dim() == 4 and tuple(window) == tuple(strides)
            and x.shape[-1] % vec == 0 and x.numel() > 0 and x.numel() < 2 ** 31 and x.data_ptr() % 16 == 0)


class _ConvReluPoolFn(torch.autograd.Function):
    """qk_conv_relu_pool_fwd / _bwd: conv (3,5) 'same' + relu + max-pool over the first spatial axis as one launch per
    direction (the TIMIT model's first layer); the input gets no gradient."""

    @staticmethod
    def forward(ctx, x, w, bias, call, pool):
        n, h, wd = call.desc.batch, call.desc.in_spatial[0], call.desc.in_spatial[1]      # (x: (N, H, W, 4) or planes (N, 4, H, W))
        out = torch.empty((n, -(-h // pool), wd, w.shape[-1]), dtype=x.dtype, device=x.device)
        nb = int(L.lib().qk_conv_relu_pool_aux_bytes(ctypes.byref(call.desc), pool))
        keep = any(ctx.needs_input_grad[1:3])
        ctx.params = (w, bias)
        aux = torch.empty(nb, dtype=torch.uint8, device=x.device) if keep else None
        with _on_device(x.device):
            rc = L.lib().qk_conv_relu_pool_fwd(ctypes.byref(call.desc), pool, _ptr(x), _ptr(w), _ptr(bias), _ptr(out), _ptr(aux), _stream(x))
        L.check(rc, 'qk_conv_relu_pool_fwd')
        ctx.call, ctx.pool, ctx.has_bias = call, pool, bias is not None
        ctx.w_shape = tuple(w.shape)
        ctx.save_for_backward(x, aux)
        return out

    @staticmethod
    def backward(ctx, dout):
        x, aux = ctx.saved_tensors
        dout = dout.contiguous()
        direct = _direct_grad(ctx.params[0], ctx.params[1], ctx.needs_input_grad[1], ctx.has_bias and ctx.needs_input_grad[2])
        if direct is not None:
            dw, db = direct
        else:
            dw = torch.empty(ctx.w_shape, dtype=torch.float32, device=x.device)
            db = torch.empty((ctx.w_shape[-1],), dtype=torch.float32, device=x.device) if ctx.has_bias else None
        with _on_device(x.device):
            rc = L.lib().qk_conv_relu_pool_bwd(ctypes.byref(ctx.call.desc), ctx.pool, _ptr(x), _ptr(dout), _ptr(aux), _ptr(dw), _ptr(db),
                                               L.QK_BWD_ACCUMULATE if direct is not None else 0, _stream(x))
        L.check(rc, 'qk_conv_relu_pool_bwd')
        if direct is not None:
            _grad_ready(*ctx.params)
            return None, None, None, None, None
        return None, dw, db, None, None


def _first_layer_call(x, kernel, activation, use_bias, x_layout):
    """Descriptor of the fused first layer; x_layout 'channels_last' = (N, H, W, 4), 'channels_first' = the four component
    planes (N, 4, H, W) read as they lie (include/qk.h: for these entry points desc.layout describes x only)."""
    if x_layout == 'channels_first':
        n, c, h, w = x.shape
        call = conv_call((n, h, w, c), tuple(kernel.shape), x.dtype, 2, 1, 'same', 'channels_last', 1, activation, use_bias, False)
        call.desc.layout = L.QK_CH_FIRST
        return call
    return conv_call(tuple(x.shape), tuple(kernel.shape), x.dtype, 2, 1, 'same', 'channels_last', 1, activation, use_bias, False)


def conv_relu_pool_supported(x, kernel, pool, x_layout='channels_last'):
    """True when relu(QuaternionConv2D(kernel, 'same')(x)) followed by MaxPooling over the first spatial axis (window =
    stride = pool, 'same') can run as the fused first-layer kernels: x a contiguous channels_last (N, H, W, 4) 16-bit
    device tensor that needs no gradient, kernel (3, 5, 1, 4F) with F % 8 == 0, pool == 3, and a height whose
    TensorFlow 'same' pooling pads on the high side only (H % 3 != 1: the kernel's windows start at row 0).  The
    answer is the C side's (qk_conv_relu_pool_aux_bytes is non-zero exactly for the geometries it takes), so the
    two predicates cannot drift apart."""
    ch_axis, h_axis = (1, 2) if x_layout == 'channels_first' else (-1, 1)
    if not (x.is_cuda and x.dtype in (torch.bfloat16, torch.float16) and x.dim() == 4 and x.shape[ch_axis] == 4 and not x.requires_grad
            and kernel.dim() == 4 and tuple(kernel.shape[:3]) == (3, 5, 1) and kernel.shape[-1] % 4 == 0 and x.shape[0] > 0):
        return False
    if tf_pads(x.shape[h_axis], pool, pool, 1, 'same')[0] != 0:
        return False
    try:
        call = _first_layer_call(x, kernel, 'relu', True, x_layout)
    except ValueError:
        return False
    return int(L.lib().qk_conv_relu_pool_aux_bytes(ctypes.byref(call.desc), pool)) > 0


def conv_relu_pool(x, kernel, bias=None, pool=3, x_layout='channels_last'):
    """(N, H, W, 4) -- or, with x_layout='channels_first', the component planes (N, 4, H, W) as the reference's model
    receives them -- -> channels_last (N, ceil(H / pool), W, 4F): the first TIMIT layer and its frequency pooling in one
    kernel (include/qk.h: qk_conv_relu_pool_*).  Check conv_relu_pool_supported first."""
    _require_device(x, 'conv_relu_pool')
    _check_weights(kernel, bias, kernel.shape[-1])
    xc = x.contiguous()
    call = _first_layer_call(xc, kernel, 'relu', bias is not None, x_layout)
    if not L.lib().qk_conv_relu_pool_aux_bytes(ctypes.byref(call.desc), pool):
        raise RuntimeError('conv_relu_pool: geometry outside the fused first-layer kernel')
    return _ConvReluPoolFn.apply(xc, kernel.contiguous(), bias, call, pool)


class _ConvPreluPoolFn(torch.autograd.Function):
    """qk_conv_prelu_pool_fwd / _bwd: linear conv (3,5) 'same' + PReLU (slope per row, or one) + max-pool over the first
    spatial axis as one launch per direction; the input gets no gradient."""

    @staticmethod
    def forward(ctx, x, w, bias, alpha, call, pool, post):
        n, h, wd = call.desc.batch, call.desc.in_spatial[0], call.desc.in_spatial[1]
        ctx.params = (w, bias)
        out = torch.empty((n, -(-h // pool), wd, w.shape[-1]), dtype=x.dtype, device=x.device)
        nb = int(L.lib().qk_conv_relu_pool_aux_bytes(ctypes.byref(call.desc), pool))
        keep = any(ctx.needs_input_grad[1:4])
        aux = torch.empty(nb, dtype=torch.uint8, device=x.device) if keep else None
        pre = torch.empty_like(out) if keep else None
        with _on_device(x.device):
            rc = L.lib().qk_conv_prelu_pool_fwd(ctypes.byref(call.desc), pool, ctypes.byref(post.struct), _ptr(x), _ptr(w), _ptr(bias),
                                                _ptr(out), _ptr(pre), _ptr(aux), _stream(x))
        L.check(rc, 'qk_conv_prelu_pool_fwd')
        ctx.call, ctx.pool, ctx.post, ctx.has_bias = call, pool, post, bias is not None
        ctx.w_shape = tuple(w.shape)
        ctx.save_for_backward(x, aux, pre)
        return out

    @staticmethod
    def backward(ctx, dout):
        x, aux, pre = ctx.saved_tensors
        dout = dout.contiguous()
        post = ctx.post
        direct = _direct_grad(ctx.params[0], ctx.params[1], ctx.needs_input_grad[1], ctx.has_bias and ctx.needs_input_grad[2])
        if direct is not None:
            dw, db = direct
        else:
            dw = torch.empty(ctx.w_shape, dtype=torch.float32, device=x.device)
            db = torch.empty((ctx.w_shape[-1],), dtype=torch.float32, device=x.device) if ctx.has_bias else None
        da = torch.zeros(post.flat.numel(), dtype=torch.float32, device=x.device)
        with _on_device(x.device):
            rc = L.lib().qk_conv_prelu_pool_bwd(ctypes.byref(ctx.call.desc), ctx.pool, ctypes.byref(post.struct), _ptr(x), _ptr(dout),
                                                _ptr(pre), _ptr(aux), _ptr(dw), _ptr(db), _ptr(da),
                                                L.QK_BWD_ACCUMULATE if direct is not None else 0, _stream(x))
        L.check(rc, 'qk_conv_prelu_pool_bwd')
        if direct is not None:
            _grad_ready(*ctx.params)
            return None, None, None, da.reshape(post.alpha.shape), None, None, None
        return None, dw, db, da.reshape(post.alpha.shape), None, None, None


def conv_prelu_pool_supported(x, kernel, alpha, alpha_axis, pool, x_layout='channels_last'):
    """conv_relu_pool_supported for the PReLU form: float32 device slopes, one (alpha_axis -1) or one per position of
    the first spatial axis (alpha_axis 0, at most 64)."""
    n = alpha.numel()
    h = x.shape[2] if x_layout == 'channels_first' else x.shape[1]
    return (conv_relu_pool_supported(x, kernel, pool, x_layout) and alpha.is_cuda and alpha.dtype == torch.float32 and
            ((alpha_axis == -1 and n == 1) or (alpha_axis == 0 and n == h and n <= 64)))


def conv_prelu_pool(x, kernel, bias, alpha, alpha_axis=0, pool=3, x_layout='channels_last'):
    """(N, H, W, 4) [or planes (N, 4, H, W), x_layout='channels_first'] -> (N, ceil(H / pool), W, 4F): linear first layer +
    PReLU + frequency pooling in one kernel (include/qk.h: qk_conv_prelu_pool_*).  Check conv_prelu_pool_supported first."""
    _require_device(x, 'conv_prelu_pool')
    _check_weights(kernel, bias, kernel.shape[-1])
    xc = x.contiguous()
    call = _first_layer_call(xc, kernel, None, bias is not None, x_layout)
    post = PostOp(alpha, alpha_axis, 0.0, 0)
    return _ConvPreluPoolFn.apply(xc, kernel.contiguous(), bias, alpha, call, pool, post)


# Diagnostic tap (tests): when set to a callable it receives the list [x, y_1, ..., y_n] of every chain's activations
# as the forward produced them (tests/test_timit_parity.py compares the 16-bit gradients on the GPU's own relu masks).
chain_tap = None


class _ConvChainFn(torch.autograd.Function):
    """A run of quaternion convolutions applied back to back as ONE autograd node, so that the backward knows the
    structure.  Each layer ends in a fused relu (y_i = relu(W_i (x) y_{i-1} + b_i)), is linear, or carries a POST-OP
    (y_i = dropout(prelu(pre_i)), pre_i written beside y_i by the same launch):
      * relu: layer i+1's backward-data returns its input gradient already multiplied by (y_i > 0) (epilogue of the
        kernel, QK_BWD_MASK_DX) and layer i runs its backward without the mask and without reading y_i again
        (QK_BWD_DY_PREMASKED);
      * post-op: layer i+1's backward-data epilogue turns d y_i into d pre_i (dropout mask regenerated from the seed,
        PReLU derivative from pre_i) and accumulates d alpha_i (qk_conv_bwd_post); only the LAST layer's post-op needs
        a pass of its own (qk_postop_bwd).
    Gradients are identical to the layer-by-layer form."""

    @staticmethod
    def forward(ctx, x, calls, posts, *params):
        n = len(calls)
        ws, bs, alphas = params[:n], params[n:2 * n], params[2 * n:]
        acts, pres = [x], []
        for call, w, b, post in zip(calls, ws, bs, posts):
            if post is None:
                acts.append(call.fwd(acts[-1], w, b, wparam=w))
                pres.append(None)
            else:
                pre, y = call.fwd_post(acts[-1], w, b, post, wparam=w)
                acts.append(y)
                pres.append(pre)
        if chain_tap is not None:
            chain_tap(list(acts))
        ctx.calls, ctx.posts = calls, posts
        ctx.param_refs = (ws, bs)
        ctx.has_bias = [b is not None for b in bs]
        ctx.n_pre = [p is not None for p in pres]
        ctx.save_for_backward(*acts, *ws, *[p for p in pres if p is not None])
        return acts[-1]

    @staticmethod
    def backward(ctx, dy):
        calls, posts = ctx.calls, ctx.posts
        n = len(calls)
        saved = ctx.saved_tensors
        acts, ws = saved[:n + 1], saved[n + 1:2 * n + 1]
        it = iter(saved[2 * n + 1:])
        pres = [next(it) if has else None for has in ctx.n_pre]
        g = dy.contiguous()
        dws, dbs, das = [None] * n, [None] * n, [None] * n
        for i, post in enumerate(posts):
            if post is not None and post.alpha is not None:
                das[i] = torch.zeros(post.flat.numel(), dtype=torch.float32, device=g.device)
        if posts[n - 1] is not None:                      # the last post-op has no consumer inside the chain
            last = posts[n - 1]                           # (relu form: the output y is its own mask)
            g = postop_bwd(pres[n - 1] if last.alpha is not None else acts[n], g, last, das[n - 1])
        for i in range(n - 1, -1, -1):
            pw, pb = ctx.param_refs[0][i], ctx.param_refs[1][i]
            direct = _direct_grad(pw, pb, ctx.needs_input_grad[3 + i], ctx.has_bias[i] and ctx.needs_input_grad[3 + n + i])
            if i > 0 and posts[i - 1] is not None:
                g, dws[i], dbs[i] = calls[i].bwd_post(acts[i], g, ws[i], ctx.has_bias[i], posts[i - 1], pres[i - 1], das[i - 1],
                                                      direct=direct, wparam=pw)
                if direct is not None:
                    _grad_ready(pw, pb)
                continue
            if i == 0 and not ctx.needs_input_grad[0]:
                # the chain's input needs no gradient (a first layer): backward-weight only.  (A relu layer whose dy
                # arrives masked is masked once more by this call -- idempotent.)
                if direct is not None:
                    calls[0].bwd_weight(acts[0], g, acts[1], ctx.has_bias[0], out=direct, accumulate=True)
                    _grad_ready(pw, pb)
                else:
                    dws[0], dbs[0] = calls[0].bwd_weight(acts[0], g, acts[1], ctx.has_bias[0])
                g = None
                continue
            flags = 0
            if i > 0 and calls[i - 1].relu:
                flags |= L.QK_BWD_MASK_DX
            if i < n - 1 and calls[i].relu:
                flags |= L.QK_BWD_DY_PREMASKED
            if direct is not None:
                dx = torch.empty(calls[i].x_shape, dtype=g.dtype, device=g.device)
                calls[i].bwd(acts[i], g, acts[i + 1], ws[i], ctx.has_bias[i], out=(dx,) + direct, flags=flags | L.QK_BWD_ACCUMULATE,
                             wparam=pw)
                _grad_ready(pw, pb)
                g = dx
            else:
                g, dws[i], dbs[i] = calls[i].bwd(acts[i], g, acts[i + 1], ws[i], ctx.has_bias[i], flags=flags, wparam=pw)
        das = [None if d is None else d.reshape(p.alpha.shape) for d, p in zip(das, posts)]      # (relu form: no slopes)
        dws = [None if d is None else d.reshape(w.shape) for d, w in zip(dws, ctx.param_refs[0])]   # (dense links: 2-D parameters)
        return (g, None, None) + tuple(dws) + tuple(dbs) + tuple(das)


def quaternion_conv_chain(x, layers):
    """Apply consecutive quaternion convolutions as one autograd node (see _ConvChainFn).
    `layers`: sequence of (kernel, bias, kwargs) with the keyword arguments of quaternion_conv
    (strides, padding, dilation_rate, activation ('relu' / 'linear' / None), conj) and optionally
    post=dict(alpha=<float32 tensor>, alpha_axis=-1|0|1|2, rate=<dropout rate>, seed=<int>): PReLU (+ dropout) behind
    a LINEAR layer (alpha=None: relu + dropout, the reference's aact='none' setting -- one output tensor per layer);
    x and every layer are channels_last here -- channels_first callers pass the channels-last view
    and move the axis back."""
    _require_device(x, 'quaternion_conv_chain')
    xp = x.contiguous()
    calls, ws, bs, posts, alphas = [], [], [], [], []
    shape = tuple(xp.shape)
    for kernel, bias, kw in layers:
        order = 0
        if kernel.dim() == 2 and kw.get('dense_kernel_size') is not None:
            # a QuaternionDense weight over the FLATTENED (channel, *kernel_size) axes of the feature map (the TIMIT model's first
            # TimeDistributed dense layer: row cq * F + f of the weight is tap f, channel cq of an (F, 1) 'valid' convolution,
            # interspeech_model.py:140-149): the parameter is read in place as a CHANNEL-MAJOR kernel (qk_conv_desc_t.kernel_order)
            # -- no permuted copy per step, cached 16-bit re-layout, kernel gradient straight into the parameter's own layout
            ksz = tuple(int(v) for v in kw['dense_kernel_size'])
            rank = len(ksz)
            taps = int(math.prod(ksz))
            if kernel.shape[0] % taps:
                raise ValueError('dense kernel rows %d are not a multiple of the %d taps' % (kernel.shape[0], taps))
            kw = dict(kw, conj=True, strides=1, padding='valid', dilation_rate=1)
            w_shape = ksz + (kernel.shape[0] // taps, kernel.shape[1])
            order = L.QK_KERNEL_CHANNEL_MAJOR
        elif kernel.dim() == 2:
            # a QuaternionDense weight (in_q, 4 * units): the layer applied to every position of the channels-last tensor is
            # the 1 x ... x 1 conj-convolution with that weight as its single tap (dense.py:139-143 builds the transposed
            # table).  The PARAMETER itself is passed on (same memory as the (1, ..., in_q, 4 units) kernel): its cached
            # 16-bit re-layout and the direct gradient writes keep working.
            rank = len(shape) - 2
            kw = dict(kw, conj=True, strides=1, padding='valid', dilation_rate=1)
            w_shape = (1,) * rank + tuple(kernel.shape)
        else:
            rank = kernel.dim() - 2
            w_shape = tuple(kernel.shape)
        _check_weights(kernel, bias, kernel.shape[-1])
        po = kw.get('post')
        if po is not None and kw.get('activation') not in (None, 'linear'):
            raise ValueError('a layer with a post-op must be linear (the post-op is its activation)')
        call = conv_call(shape, w_shape, xp.dtype, rank, kw.get('strides', 1), kw.get('padding', 'valid'),
                         'channels_last', kw.get('dilation_rate', 1), kw.get('activation'), bias is not None,
                         bool(kw.get('conj', False)), order)
        calls.append(call)
        ws.append(kernel.contiguous())
        bs.append(bias)
        posts.append(None if po is None else PostOp(po['alpha'], po.get('alpha_axis', -1), po.get('rate', 0.0), po.get('seed', 0), po.get('seed_dev')))
        alphas.append(None if po is None else po['alpha'])
        shape = tuple(call.y_shape)
    return _ConvChainFn.apply(xp, tuple(calls), tuple(posts), *ws, *bs, *alphas)


def softmax_rows_fwd(logits, bias, out_dtype):
    """softmax(logits + bias) over the last axis of an fp32 (rows, cols <= 64) device matrix, written in `out_dtype`
    (include/qk.h: qk_softmax_rows_fwd)."""
    _require_device(logits, 'softmax_rows_fwd')
    if logits.dtype != torch.float32 or logits.dim() != 2 or not logits.is_contiguous() or logits.shape[1] > 64:
        raise ValueError('softmax_rows_fwd: contiguous fp32 (rows, cols <= 64) logits')
    y = torch.empty(logits.shape, dtype=out_dtype, device=logits.device)
    with _on_device(logits.device):
        rc = L.lib().qk_softmax_rows_fwd(_DTYPES[out_dtype], logits.shape[0], logits.shape[1], _ptr(logits), _ptr(bias), _ptr(y), _stream(logits))
    L.check(rc, 'qk_softmax_rows_fwd')
    return y


def softmax_rows_bwd(y, dy, dbias=None):
    """d logits = y * (dy - <dy, y>) (returned, y's dtype); the bias gradient (column sums) is ADDED to the fp32 buffer `dbias`."""
    _require_device(y, 'softmax_rows_bwd')
    if dy.dtype != y.dtype or dy.shape != y.shape or dy.device != y.device or y.dim() != 2:
        raise ValueError('softmax_rows_bwd: dy must match y (2-D, same shape, dtype and device); got %s %s vs %s %s'
                         % (tuple(dy.shape), dy.dtype, tuple(y.shape), y.dtype))
    if dbias is not None and (dbias.dtype != torch.float32 or dbias.numel() != y.shape[1] or dbias.device != y.device or not dbias.is_contiguous()):
        raise ValueError('softmax_rows_bwd: dbias must be a contiguous fp32 device tensor with one entry per column')
    y, dy = y.contiguous(), dy.contiguous()           # raw pointers go to the kernel
    dl = torch.empty_like(y)
    with _on_device(y.device):
        rc = L.lib().qk_softmax_rows_bwd(_DTYPES[y.dtype], y.shape[0], y.shape[1], _ptr(y), _ptr(dy), _ptr(dl), _ptr(dbias), _stream(y))
    L.check(rc, 'qk_softmax_rows_bwd')
    return dl


def dense_softmax_supported(x, units):
    """True when qk_dense_softmax_fwd / _bwd take softmax(x @ kernel + bias) for this 16-bit (rows, in_dim) device matrix."""
    return (x.is_cuda and x.dim() == 2 and x.dtype in (torch.bfloat16, torch.float16) and x.is_contiguous()
            and bool(L.lib().qk_dense_softmax_supported(_DTYPES[x.dtype], x.shape[0], x.shape[1], int(units))))


def dense_softmax_fwd(x, kernel, bias):
    """y = softmax(x @ kernel + bias) in ONE launch (include/qk.h: qk_dense_softmax_fwd): x (rows, in_dim) 16-bit, kernel (in_dim, units)
    and bias (units) the fp32 master weights."""
    _require_device(x, 'dense_softmax_fwd')
    if kernel.dtype != torch.float32 or not kernel.is_contiguous() or kernel.dim() != 2 or kernel.shape[0] != x.shape[1] or kernel.device != x.device:
        raise ValueError('dense_softmax_fwd: kernel must be a contiguous fp32 (in_dim, units) device tensor')
    if bias is not None and (bias.dtype != torch.float32 or bias.numel() != kernel.shape[1] or not bias.is_contiguous() or bias.device != x.device):
        raise ValueError('dense_softmax_fwd: bias must be a contiguous fp32 device tensor with one entry per unit')
    x = x.contiguous()
    y = torch.empty((x.shape[0], kernel.shape[1]), dtype=x.dtype, device=x.device)
    with _on_device(x.device):
        rc = L.lib().qk_dense_softmax_fwd(_DTYPES[x.dtype], x.shape[0], x.shape[1], kernel.shape[1], _ptr(x), _ptr(kernel), _ptr(bias), _ptr(y), _stream(x))
    L.check(rc, 'qk_dense_softmax_fwd')
    return y


def dense_softmax_bwd(x, kernel, y, dy, dkernel, dbias, dy_scale_dev=None, dy_scale=1.0):
    """dx (returned) of y = softmax(x @ kernel + bias); the kernel / bias gradients are ADDED to the fp32 buffers `dkernel` /
    `dbias` (None: not computed) -- qk_dense_softmax_bwd, one launch (+ the slab reduction).  dy enters multiplied by
    `dy_scale_dev` (a one-element fp32 DEVICE tensor or None) x `dy_scale` (a Python float)."""
    if dy_scale_dev is not None and (dy_scale_dev.dtype != torch.float32 or dy_scale_dev.numel() != 1 or dy_scale_dev.device != x.device):
        raise ValueError('dense_softmax_bwd: dy_scale_dev must be a one-element fp32 tensor on the device of x')
    _require_device(x, 'dense_softmax_bwd')
    if dy.dtype != y.dtype or dy.shape != y.shape or y.dtype != x.dtype or y.device != x.device or dy.device != x.device:
        raise ValueError('dense_softmax_bwd: y / dy must match each other and x in dtype and device')
    for name, t, n in (('dkernel', dkernel, kernel.numel()), ('dbias', dbias, kernel.shape[1])):
        if t is not None and (t.dtype != torch.float32 or t.numel() != n or not t.is_contiguous() or t.device != x.device):
            raise ValueError('dense_softmax_bwd: %s must be a contiguous fp32 device tensor of %d elements' % (name, n))
    x, y, dy = x.contiguous(), y.contiguous(), dy.contiguous()
    dx = torch.empty_like(x)
    with _on_device(x.device):
        ws, n = None, 0
        if dkernel is not None or dbias is not None:
            n = int(L.lib().qk_dense_softmax_bwd_workspace_bytes(_DTYPES[x.dtype], x.shape[0], x.shape[1], kernel.shape[1]))
            ws = torch.empty(n, dtype=torch.uint8, device=x.device)           # (the caching allocator hands the same block back every step)
        rc = L.lib().qk_dense_softmax_bwd(_DTYPES[x.dtype], x.shape[0], x.shape[1], kernel.shape[1], _ptr(x), _ptr(kernel), _ptr(y), _ptr(dy),
                                          _ptr(dx), _ptr(dkernel), _ptr(dbias), _ptr(dy_scale_dev), float(dy_scale), _ptr(ws), n, _stream(x))
    L.check(rc, 'qk_dense_softmax_bwd')
    return dx


class _WeightedSumFn(torch.autograd.Function):
    """sum(a * w) for a 16-bit / fp32 device tensor `a` and a CONSTANT fp32 weight tensor `w` as one launch (qk_weighted_sum);
    the backward hands back w in a's dtype (cast once, kept on the weight tensor)."""

    @staticmethod
    def forward(ctx, a, w):
        out = torch.zeros((), dtype=torch.float32, device=a.device)
        with _on_device(a.device):
            rc = L.lib().qk_weighted_sum(_DTYPES[a.dtype], a.numel(), _ptr(a), _ptr(w), _ptr(out), _stream(a))
        L.check(rc, 'qk_weighted_sum')
        cache = w.__dict__.setdefault('_qk_cast', {})
        _CAST_PARAMS[id(w)] = w
        hit = cache.get(a.dtype)
        ver = (w._version, w.data_ptr())
        if hit is None or hit[0] != ver:
            hit = cache[a.dtype] = (ver, w.to(a.dtype))
        ctx.wt = hit[1]
        ctx.a_shape = a.shape
        return out

    @staticmethod
    def backward(ctx, g):
        return (ctx.wt * g.to(ctx.wt.dtype)).reshape(ctx.a_shape), None          # (w may have a's element count in another shape)


def weighted_sum(a, w):
    """The linear functional sum(a * w) of a model output (the bench's stand-in loss), one launch forward, one backward."""
    _require_device(a, 'weighted_sum')
    if w.dtype != torch.float32 or w.numel() != a.numel() or w.requires_grad:
        raise ValueError('weighted_sum: w must be a constant fp32 tensor with as many elements as a')
    return _WeightedSumFn.apply(a.contiguous(), w.contiguous())


class _CtcFn(torch.autograd.Function):
    """qk_ctc_batch_cost: cost and d cost / d y_pred from one launch; the backward only scales the stored gradient."""

    @staticmethod
    def forward(ctx, y_pred, labels, input_length, label_length, loss_scale=1.0):
        b, t, c = y_pred.shape
        lab = labels.to(device=y_pred.device, dtype=torch.int32).contiguous()
        il = input_length.reshape(-1).to(device=y_pred.device, dtype=torch.int32).contiguous()
        ll = label_length.reshape(-1).to(device=y_pred.device, dtype=torch.int32).contiguous()
        lmax = lab.shape[1] if lab.dim() == 2 else 0
        cost = torch.empty(b, dtype=torch.float32, device=y_pred.device)
        want = ctx.needs_input_grad[0]
        dpred = torch.empty_like(y_pred) if want else None
        n = int(L.lib().qk_ctc_workspace_bytes(b, t, lmax))
        ws = torch.empty(n, dtype=torch.uint8, device=y_pred.device)
        with _on_device(y_pred.device):
            rc = L.lib().qk_ctc_batch_cost(_DTYPES[y_pred.dtype], b, t, c, _ptr(y_pred), _ptr(lab), lmax, _ptr(il), _ptr(ll), _ptr(cost),
                                           _ptr(dpred), _ptr(ws), n, _stream(y_pred))
        L.check(rc, 'qk_ctc_batch_cost')
        ctx.save_for_backward(dpred)
        ctx.loss_scale = float(loss_scale)
        return cost.reshape(b, 1)

    @staticmethod
    def backward(ctx, dcost):
        dpred, = ctx.saved_tensors
        if ctx.loss_scale != 1.0:
            # the product is formed in fp32 and rounded once: d cost / d y can be ~1 / y, dcost ~1 / batch
            return (dpred.float() * (dcost.reshape(-1, 1, 1).float() * ctx.loss_scale)).to(dpred.dtype), None, None, None, None
        return dpred * dcost.reshape(-1, 1, 1).to(dpred.dtype), None, None, None, None


def ctc_cost_and_grad(y_pred, labels, input_length, label_length):
    """(cost (B,), d cost / d y_pred) of qk_ctc_batch_cost as plain tensors (no autograd node): for callers that chain the gradient
    themselves (layers._DenseSoftmaxCtcMeanFn)."""
    _require_device(y_pred, 'ctc_cost_and_grad')
    y_pred = y_pred.contiguous()
    b, t, c = y_pred.shape
    lab = labels.to(device=y_pred.device, dtype=torch.int32).contiguous()
    il = input_length.reshape(-1).to(device=y_pred.device, dtype=torch.int32).contiguous()
    ll = label_length.reshape(-1).to(device=y_pred.device, dtype=torch.int32).contiguous()
    lmax = lab.shape[1] if lab.dim() == 2 else 0
    cost = torch.empty(b, dtype=torch.float32, device=y_pred.device)
    dpred = torch.empty_like(y_pred)
    n = int(L.lib().qk_ctc_workspace_bytes(b, t, lmax))
    ws = torch.empty(n, dtype=torch.uint8, device=y_pred.device)
    with _on_device(y_pred.device):
        rc = L.lib().qk_ctc_batch_cost(_DTYPES[y_pred.dtype], b, t, c, _ptr(y_pred), _ptr(lab), lmax, _ptr(il), _ptr(ll), _ptr(cost),
                                       _ptr(dpred), _ptr(ws), n, _stream(y_pred))
    L.check(rc, 'qk_ctc_batch_cost')
    return cost, dpred


def ctc_supported(y_pred, labels):
    """qk_ctc_batch_cost takes a contiguous (B, T, C) device tensor with C <= 256 and at most 127 labels per sample."""
    return (y_pred.is_cuda and y_pred.dtype in _DTYPES and y_pred.dim() == 3 and y_pred.shape[-1] <= 256 and y_pred.shape[0] > 0
            and labels.dim() == 2 and labels.shape[1] <= 127 and (y_pred.shape[1] + 8 * labels.shape[1] + 4 + 2 * y_pred.shape[-1] + 4) * 4 <= 64 * 1024)


def ctc_batch_cost(y_pred, labels, input_length, label_length, loss_scale=1.0):
    """K.ctc_batch_cost(labels, y_pred, input_length, label_length) (interspeech_model.py:37-39): per-sample CTC cost (B, 1) of
    the softmax outputs y_pred (B, T, C), blank = C - 1, Keras / TensorFlow semantics (include/qk.h: qk_ctc_batch_cost).

    loss_scale: the GRADIENT this node sends back is multiplied by it (the cost it returns is not) -- static loss scaling for
    float16 activations: under the CTC cost the gradients of the TIMIT body layers sit at 2^-18.5 (profiles/r04_loss_ab.txt), below
    float16's normal range (2^-14); a power of two (2^12 recommended) moves them into it exactly, and the optimiser undoes it in
    fp32: `adam_step(grad_scale=1 / (world * loss_scale))`.  bfloat16 / float32 need none."""
    _require_device(y_pred, 'ctc_batch_cost')
    if not (loss_scale > 0 and math.isfinite(loss_scale)):
        raise ValueError('loss_scale must be a positive finite number')
    return _CtcFn.apply(y_pred.contiguous(), labels, input_length, label_length, float(loss_scale))
