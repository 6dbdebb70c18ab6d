#!/usr/bin/env python
"""Where the time of k_hconv16_small goes (16 -> 16 forward at B = 256): ablation bits -- WRONG results, timing only:
  0 full | 4 no MFMA stages | 8 stores dropped | 16 band loads dropped (out-of-range DMA lanes) | 32 no epilogue | sums combine"""
import os, sys, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.realpath(__file__)))
sys.path.insert(0, ROOT)
import torch
import qcnn_amd
from qcnn_amd import functional as F, _lib
from ab_layers import timeit, SHAPES
dev = torch.device('cuda:0'); dt = torch.bfloat16
g = torch.Generator(device=dev).manual_seed(0)
for n in (sys.argv[1:] or ['c16']):
    s = SHAPES[n]
    x = torch.randn(s['x'], device=dev, generator=g).to(dt)
    w = torch.randn(s['w'], device=dev, generator=g) / 30
    b = torch.zeros(s['w'][-1], device=dev)
    call = F.conv_call(tuple(s['x']), tuple(s['w']), dt, 2, 1, s['pad'], 'channels_last', 1, None, True, False)
    call.static_buffers = True
    y = call.fwd(x, w, b)
    for a in (0, 4, 8, 16, 32, 4 + 16, 4 + 32, 16 + 32, 4 + 16 + 32):
        with _lib.debug_flags(0, ablate=a):
            call.fwd(x, w, b, out=y); torch.cuda.synchronize()
            t = timeit(lambda: call.fwd(x, w, b, out=y), 5, 4)
        print('%-8s fwd ablate %2d  med %8.1f us  min %8.1f us  (%s)' % (n, a, statistics.median(t), min(t), _lib.last_path()))
