#!/usr/bin/env python
"""Where the time of the streaming point-form kernel goes (the TIMIT head's backward-data, chain form: producer's relu + dropout
mask in the epilogue): qk_set_debug_flags ablation bits -- WRONG results, timing only:
  0 full | 4 no MFMAs | 8 stores dropped (out-of-range offsets) | 16 mask loads dropped | 32 no A-fragment prefetch | sums combine"""
import os, sys, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.realpath(__file__)))
sys.path.insert(0, ROOT)
import torch
import qcnn_amd
from qcnn_amd import functional as F, _lib
from ab_layers import timeit

dev = torch.device('cuda:0')
dt = torch.bfloat16
g = torch.Generator(device=dev).manual_seed(0)
xs, ws = (256, 14, 200, 256), (14, 1, 64, 256)
x = torch.relu(torch.randn(xs, device=dev, generator=g)).to(dt)          # the producer's output: its sign pattern is the mask
w = torch.randn(ws, device=dev, generator=g) / 30
b = torch.zeros(ws[-1], device=dev)
call = F.conv_call(xs, ws, dt, 2, 1, 'valid', 'channels_last', 1, None, True, True)
call.static_buffers = True
y = call.fwd(x, w, b)
dy = torch.randn(y.shape, device=dev, generator=g).to(dt)
post_x = F.PostOp(None, -1, 0.3, 1)
dw, db = torch.zeros(ws, device=dev), torch.zeros(ws[-1], device=dev)
forms = {'masked (chain: relu + dropout of the producer)': lambda: call.bwd_post(x, dy, w, True, post_x, None, None, direct=(dw, db)),
         'linear (no mask)': lambda: call.bwd_data(dy, None, w)}
for name, fn in forms.items():
    for a in (0, 4, 8, 16, 32, 8 + 16, 4 + 8 + 16, 4 + 8 + 16 + 32):
        with _lib.debug_flags(0, ablate=a):
            fn(); torch.cuda.synchronize()
            t = timeit(fn, 5, 4)
        print('%-48s ablate %2d  med %8.1f us  min %8.1f us   (%s)' % (name, a, statistics.median(t), min(t), _lib.last_path()))
