"""Data-parallel training plumbing: one process per GPU, RCCL gradient all-reduce over xGMI.

The reference has no distributed path at all (an unused `multi_gpu_model`
import, models/interspeech_model.py:25); BASELINE.json's north star asks for pure data
parallelism on one 8xMI355X node.  The hot path shards by batch rows (independent samples),
every rank holds a full replica of the small compact weights, and the only exchange per
step is ONE sum all-reduce of the flat fp32 compact-gradient buffer:

  * all parameters / gradients of the quaternion layers live in two flat fp32 buffers
    (`FlatParams`), so the exchange is a single collective (xGMI is point-to-point, 7 links
    x ~153 GB/s per GPU: a ring all-reduce of S bytes is bound by 2*(N-1)/N * S / link-BW,
    and for these <= 150 MB buffers latency, not bandwidth, is what matters -> one big
    message, not per-layer messages);
  * the 1/world_size averaging is folded into the fused Adam kernel (`grad_scale`);
  * `torch.distributed` backend "nccl" IS RCCL on ROCm; "gloo" is used by the CPU tests.
"""
import os
import socket
import subprocess
import sys
import time

import torch
import torch.distributed as dist

# Diagnostic: with QK_DP_FORCE_COLLECTIVES=1 a single process still creates the process group and issues
# the collectives (a one-rank RCCL communicator).  That is how the 1-GPU test box exercises the exact
# init / all-reduce / barrier calls the 8-GPU run makes.
_FORCE = bool(os.environ.get('QK_DP_FORCE_COLLECTIVES'))


def free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


def spawn_ranks(nproc, argv, env=None, timeout=None, quiet_ranks=True):
    """One process per GPU on this node: start `nproc` copies of the command `argv` with the environment
    torch.distributed.run would give them (RANK, LOCAL_RANK, WORLD_SIZE, LOCAL_WORLD_SIZE, MASTER_ADDR = 127.0.0.1,
    a free MASTER_PORT) and wait for all of them.  This is what `python bench.py --gpus N` does when it is NOT already
    running under a launcher.  Rank 0 inherits stdout (it prints the result line); the other ranks' stdout is dropped
    when `quiet_ranks`; stderr always passes through.  If a rank fails, the others are terminated (by their own
    PIDs) and its exit code is returned; 0 when every rank exited cleanly."""
    if nproc < 1:
        raise ValueError('nproc must be >= 1')
    base = dict(os.environ if env is None else env)
    base.update(WORLD_SIZE=str(nproc), LOCAL_WORLD_SIZE=str(nproc), MASTER_ADDR='127.0.0.1', MASTER_PORT=str(free_port()))
    base.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')       # dmabuf IPC: RCCL's intra-node transport needs it here
    procs = []
    for r in range(nproc):
        e = dict(base, RANK=str(r), LOCAL_RANK=str(r))
        out = subprocess.DEVNULL if (r > 0 and quiet_ranks) else None
        procs.append(subprocess.Popen(list(argv), env=e, stdout=out))
    t0, rc = time.time(), 0
    live = list(procs)
    while live:
        for pr in list(live):
            code = pr.poll()
            if code is None:
                continue
            live.remove(pr)
            if code != 0 and rc == 0:
                rc = code
        if rc != 0 or (timeout is not None and time.time() - t0 > timeout):
            for pr in live:                    # a rank died (or time is up): the others would hang in a collective
                pr.terminate()
            for pr in live:
                try:
                    pr.wait(timeout=10)
                except subprocess.TimeoutExpired:
                    pr.kill()
            if rc == 0:
                rc = 124
            break
        if live:
            time.sleep(0.05)
    return rc


def init_from_env(backend=None):
    """Initialise torch.distributed from RANK / WORLD_SIZE / LOCAL_RANK / MASTER_* (torchrun).
    Returns (rank, world_size, local_rank).  World size 1 needs no process group."""
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    # Diagnostics for a ONE-GPU box: QK_DP_SHARE_DEVICE=1 puts every rank on cuda:0 and QK_DP_BACKEND=gloo carries the
    # collectives (RCCL refuses two ranks on one device) -- the engine's data-parallel step (direct gradient writes, buckets
    # launched from the backward, fused Adam) then runs with REAL peer processes; tests/test_dp_gloo.py uses it.  Not a
    # performance mode.
    if os.environ.get('QK_DP_SHARE_DEVICE'):
        local = 0
    if (world > 1 or _FORCE) and not dist.is_initialized():
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29500')
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        if backend is None:
            backend = 'nccl' if torch.cuda.is_available() else 'gloo'
            override = os.environ.get('QK_DP_BACKEND')
            if override and override != backend:
                # the override exists for ONE purpose: several ranks on one device (RCCL refuses that).  A production
                # run -- one rank per GPU -- must not be able to end up on gloo because a variable leaked into its
                # environment: host-staged collectives would be silently ~100x slower.
                if not os.environ.get('QK_DP_SHARE_DEVICE'):
                    raise RuntimeError('QK_DP_BACKEND=%s is a diagnostic for ranks sharing one GPU and is honoured only together '
                                       'with QK_DP_SHARE_DEVICE=1; unset it (one rank per GPU always uses RCCL: backend "nccl")' % override)
                backend = override
        if backend == 'nccl':
            torch.cuda.set_device(local)
            dist.init_process_group(backend, rank=rank, world_size=world,
                                    device_id=torch.device('cuda', local))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    return rank, world, local


class FlatParams(object):
    """Re-homes a list of parameters into ONE flat fp32 buffer (+ a flat gradient buffer).

    After construction every parameter's `.data` is a view into `self.param` and `.grad` a view
    into `self.grad`, so an optimizer step / all-reduce touches two tensors, not 2*len(params)."""

    def __init__(self, params, direct='auto'):
        """direct: which parameters the engine's backward kernels may write their gradient into WITHOUT going through
        autograd's AccumulateGrad (functional._direct_grad).  That is only sound for a parameter whose gradient has no
        second source in the graph: 'auto' (default) excludes every parameter that carries a Keras regulariser
        (`p._qk_regularized`, set by keras_like.Layer.add_weight -- `model.regularization_loss()` differentiates
        through autograd and would add to the same buffer later); True takes all (the caller folds the regulariser
        into the optimiser step instead: `l2_decay()` + functional.adam_step(decay=)); False none."""
        self.params = [p for p in params]
        if not self.params:
            raise ValueError('no parameters')
        if direct not in ('auto', True, False):
            raise ValueError("direct must be 'auto', True or False")
        dev = self.params[0].device
        sizes = [p.numel() for p in self.params]
        # 64-element (256 B) alignment keeps every view usable for 16-byte vector access
        self.offsets, off = [], 0
        for n in sizes:
            self.offsets.append(off)
            off += (n + 63) // 64 * 64
        self.numel = off
        self.param = torch.zeros(off, dtype=torch.float32, device=dev)
        self.grad = torch.zeros(off, dtype=torch.float32, device=dev)
        with torch.no_grad():
            for p, o in zip(self.params, self.offsets):
                n = p.numel()
                self.param[o:o + n].copy_(p.detach().reshape(-1).float())
                p.data = self.param[o:o + n].view(p.shape)
                p.grad = self.grad[o:o + n].view(p.shape)
                p._qk_flat_base = self.param     # functional._Call._ws: writes through the flat buffer invalidate cached re-layouts
                # the engine's backward nodes add their kernel / bias gradients straight into these views (no
                # temporary, no memset, no AccumulateGrad add): functional._direct_grad
                p._qk_direct_grad = bool(direct is True or (direct == 'auto' and not getattr(p, '_qk_regularized', None)))

    def grad_view(self, i):
        p, o = self.params[i], self.offsets[i]
        return self.grad[o:o + p.numel()].view(p.shape)

    def zero_grad(self):
        self.grad.zero_()

    def l2_decay(self):
        """Per-element coefficient c with  d(sum of the parameters' l2 regularisers)/dw = c * w : 2 * l2 on the
        elements of a parameter created with `kernel_regularizer=l2(...)` (interspeech_model.py:63,68,173), 0 elsewhere.
        functional.adam_step(decay=...) adds c * w to the (averaged) gradient inside the fused Adam kernel, which is
        what Keras' loss term contributes -- without a second gradient source in the autograd graph.  Returns None
        when no parameter is regularised.  l1 terms cannot be folded this way and raise."""
        dec = torch.zeros_like(self.param)
        any_reg = False
        for p, o in zip(self.params, self.offsets):
            reg = getattr(p, '_qk_regularized', None)
            if reg is None:
                continue
            if getattr(reg, 'l1', 0.0) or not hasattr(reg, 'l2'):
                raise ValueError('only l2 regularisers fold into the optimiser step, got %r' % (reg,))
            if reg.l2:
                dec[o:o + p.numel()] = 2.0 * reg.l2
                any_reg = True
        return dec if any_reg else None


def broadcast_params(flat, src=0, group=None):
    """Identical replicas: rank `src`'s weights win (the reference init is host-side and seeded,
    so ranks usually agree already; this makes it unconditional)."""
    if dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.broadcast(flat.param, src=src, group=group)
        # c10d collectives write the buffer WITHOUT moving its version counter; the cached 16-bit kernel re-layouts
        # (functional._Call._ws) compare versions -- without this a non-src rank that had already run a forward would keep
        # multiplying by its pre-broadcast kernels until the next optimiser step.  Any other raw write into flat.param
        # (parameter averaging, a checkpoint load through a collective) must do the same.
        torch.autograd.graph.increment_version(flat.param)


def allreduce_sum_(tensor, group=None, async_op=False):
    """Sum all-reduce of the flat gradient buffer (RCCL on GPUs).  Returns the work handle when
    async_op; no-op for world size 1."""
    if not dist.is_initialized() or (dist.get_world_size(group) == 1 and not _FORCE):
        return None
    return dist.all_reduce(tensor, op=dist.ReduceOp.SUM, group=group, async_op=async_op)


class BucketedAllReduce(object):
    """Gradient all-reduce in buckets, launched WHILE the backward pass is still running.

    The flat gradient buffer of `FlatParams` is cut into contiguous buckets of at least `bucket_bytes`
    (parameters in creation order; the backward produces their gradients roughly in reverse).  A
    post-accumulate hook on every parameter counts its bucket down; the moment a bucket is complete its
    slice of the flat buffer goes out as one asynchronous sum all-reduce (RCCL on the process group's own
    stream, overlapping the remaining backward kernels).  `finish()` waits for every bucket -- call it before
    the optimiser step -- and after EVERY backward pass: the per-parameter counters assume one gradient event per parameter
    between two finish() calls (accumulating several micro-batches before finish() is refused, not mis-counted).
    For the 6.7 MB TIMIT model this is 3-7 messages; for the 145 MB config-5 stack one
    message per 15.7 MB layer, each hidden behind the next layer's 100+ ms of backward.
    Single-process runs (no process group) do nothing unless QK_DP_FORCE_COLLECTIVES is set."""

    def __init__(self, flat, bucket_bytes=1 << 20, group=None):
        self.flat, self.group = flat, group
        self.active = dist.is_initialized() and (dist.get_world_size(group) > 1 or _FORCE)
        self.buckets = []                 # [lo, hi) element ranges of the flat buffer
        self.bucket_of = {}               # id(param) -> bucket index
        self.sizes = []                   # parameters per bucket
        lo, cur = 0, []
        for p, off in zip(flat.params, flat.offsets):
            cur.append(p)
            hi = off + (p.numel() + 63) // 64 * 64
            if (hi - lo) * 4 >= bucket_bytes:
                self._close(lo, hi, cur)
                lo, cur = hi, []
        if cur:
            self._close(lo, flat.numel, cur)
        self.pending = list(self.sizes)
        self.fired = set()                # id(param) of the parameters that reported their gradient this step
        self.works = []
        self.launch_order = []
        self.handles = []
        self.enabled = True               # False: hooks count but nothing is launched (bench: exposed-collective time)
        self.direct_done = set()          # ... of those, the ones whose gradient the backward kernels wrote themselves
        self._defined = {}                # id(param) -> did autograd deliver a DEFINED gradient in this pass
        if self.active:
            for p in flat.params:
                if p.requires_grad:
                    self.handles.append(p.register_hook(lambda g, p=p: self._saw_grad(p, g)))
                    self.handles.append(p.register_post_accumulate_grad_hook(self._auto))
                    p._qk_grad_ready = self._direct      # called by the engine when it wrote the gradient itself

    def _close(self, lo, hi, params):
        b = len(self.buckets)
        self.buckets.append((lo, hi))
        n = 0
        for p in params:
            self.bucket_of[id(p)] = b
            n += bool(p.requires_grad)
        self.sizes.append(n)

    # One event per parameter and backward pass must mean "its gradient is complete".  Two sources of events:
    #   _direct  the engine's backward kernels ADDED the gradient into the flat buffer themselves and say so
    #            (functional._grad_ready) -- complete only if nothing else in the graph produces a gradient for the
    #            same parameter;
    #   _auto    autograd's post-accumulate hook: all contributions were summed and accumulated.  It ALSO fires, with
    #            an undefined gradient, for a parameter whose backward node returned None (the direct writers do):
    #            that event is dropped.  A DEFINED gradient for a parameter already reported by _direct is a second
    #            source (a regulariser differentiated through autograd, a tied weight); the bucket may already be on
    #            the wire with a partial sum -- refuse loudly instead of reducing garbage.
    def _saw_grad(self, p, g):
        self._defined[id(p)] = g is not None
        return None

    def _refuse(self, p, what):
        b = self.bucket_of[id(p)]
        raise RuntimeError(
            'BucketedAllReduce: %s for a parameter of shape %s in bucket %d (%s).  It is written directly by the backward '
            'kernels AND receives an autograd gradient (regulariser term, tied weight): build FlatParams(direct="auto" / '
            'False) for such parameters, or fold the regulariser into the optimiser step (FlatParams.l2_decay + '
            'adam_step(decay=)).' % (what, tuple(p.shape), b,
                                      'its bucket was already launched' if b in self.launch_order else 'bucket not launched yet'))

    def _count(self, p):
        b = self.bucket_of[id(p)]
        self.fired.add(id(p))
        self.pending[b] -= 1
        if self.pending[b] == 0:
            self._launch(b)

    def _direct(self, p):
        if id(p) in self.fired:
            self._refuse(p, 'the gradient was reported twice in one backward pass')
        self.direct_done.add(id(p))
        self._count(p)

    def _auto(self, p):
        defined = self._defined.pop(id(p), True)
        if id(p) in self.direct_done:
            if defined:
                self._refuse(p, 'a second gradient arrived through autograd')
            return                                  # the undefined-gradient echo of a direct write
        if id(p) in self.fired:
            self._refuse(p, 'the gradient was reported twice in one backward pass')
        if defined:
            self._count(p)
        # (undefined and not direct: no gradient this pass; finish() sends the bucket)

    _hook = _auto                                   # (name kept for callers that drive the counter by hand)

    def _launch(self, b):
        lo, hi = self.buckets[b]
        self.launch_order.append(b)
        if not self.enabled:
            return
        self.works.append(dist.all_reduce(self.flat.grad[lo:hi], op=dist.ReduceOp.SUM, group=self.group, async_op=True))

    def finish(self):
        """Launch whatever never fired (parameters without a gradient this step), wait for all buckets."""
        if not self.active:
            return
        for b, left in enumerate(self.pending):
            if left > 0 or (self.sizes[b] == 0 and b not in self.launch_order):
                self._launch(b)
        for w in self.works:
            w.wait()
        self.works, self.launch_order = [], []
        self.pending = list(self.sizes)
        self.fired, self.direct_done, self._defined = set(), set(), {}

    def bucket_bytes(self):
        return [4 * (hi - lo) for lo, hi in self.buckets]

    def remove(self):
        for h in self.handles:
            h.remove()
        self.handles = []
        for p in self.flat.params:
            if getattr(p, '_qk_grad_ready', None) == self._direct:
                del p._qk_grad_ready


def world_size(group=None):
    return dist.get_world_size(group) if dist.is_initialized() else 1


def shard_rows(n_rows, rank, world):
    """Contiguous batch shard of `rank` (weak scaling keeps per-rank rows fixed instead)."""
    per = (n_rows + world - 1) // world
    lo = min(n_rows, rank * per)
    return lo, min(n_rows, lo + per)
