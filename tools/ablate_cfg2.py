#!/usr/bin/env python
"""Per-phase ablation of the fp32-MFMA kernels on BASELINE configs[1] (one QuaternionConv1D layer, fp32, (64, 200, 160), 64 filters):
qk_set_debug_flags ablation bits (timing only, wrong results) -- hgemm: 4 no K loop, 8 no epilogue, 12 prologue only; wgrad: 1 no fold /
atomics, 2 no HBM atomics."""
import os, sys, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.realpath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from qcnn_amd import _lib
from ab_layers import timeit

dev = torch.device('cuda:0')
job = bench.LayerTrainStep(dict(bench.WORKLOADS['cfg2_qconv1d_timit_b64_fp32'], activation='relu'), dev, 0, 1)
for name, fn, abl in (('fwd', job.k_fwd, (0, 16, 4, 8, 12)), ('bwd_data', job.k_bwd_data, (0, 16, 4, 8, 12, 32)), ('bwd_weight', job.k_bwd_weight, (0, 1, 2))):
    for a in abl:
        with _lib.debug_flags(0, ablate=a):
            fn(); torch.cuda.synchronize()
            t = timeit(fn, 20, 5)
        print('cfg2 %-10s ablate %2d  med %7.1f us  min %7.1f us' % (name, a, statistics.median(t), min(t)))
