#!/usr/bin/env python
"""Where the time of the 16-bit kernels goes: fwd / bwd-data / bwd-weight of the TIMIT layer shapes with parts of
the kernel switched off (qk_set_debug_flags ablation bits -- results are WRONG, timing only):
  hgemm: 0 full | 4 no K loop (prologue + epilogue) | 8 no epilogue | 12 prologue only
  wgrad: 0 full | 1 no fold/atomics | 2 no HBM atomics | 4 no per-step row decode | 8 no per-step barrier | 12 both"""
import os, sys, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.realpath(__file__)))
sys.path.insert(0, ROOT)
import torch
import qcnn_amd
from qcnn_amd import functional as F, _lib
from ab_layers import SHAPES, timeit

dev = torch.device('cuda:0')
dt = torch.bfloat16
g = torch.Generator(device=dev).manual_seed(0)
for n in (sys.argv[1:] or ['c64', 'c32', 'head']):
    s = SHAPES[n]
    x = torch.randn(s['x'], device=dev, generator=g).to(dt)
    w = torch.randn(s['w'], device=dev, generator=g) / 30
    b = torch.zeros(s['w'][-1], device=dev)
    call = F.conv_call(tuple(s['x']), tuple(s['w']), dt, 2, 1, s['pad'], 'channels_last', 1, 'linear', True, s['conj'])
    call.static_buffers = True
    y = call.fwd(x, w, b)
    dy = torch.randn(y.shape, device=dev, generator=g).to(dt)
    dx = torch.empty_like(x); dw = torch.empty_like(w); db = torch.empty_like(b)
    fns = {'fwd': (lambda: call.fwd(x, w, b, out=y), (0, 4, 8, 12)),
           'bwd_data': (lambda: call.bwd_data(dy, y, w, out=dx), (0, 4, 8, 12)),
           'bwd_weight': (lambda: call.bwd_weight(x, dy, y, True, out=(dw, db)), (0, 1, 2, 4, 8, 12))}
    for k, (fn, abl) in fns.items():
        for a in abl:
            with _lib.debug_flags(0, ablate=a):
                fn(); torch.cuda.synchronize()
                t = timeit(fn, 5, 4)
            print('%-8s %-10s ablate %2d  med %8.1f us  min %8.1f us' % (n, k, a, statistics.median(t), min(t)))
