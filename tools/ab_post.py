#!/usr/bin/env python
"""Cost of the fused PReLU / dropout epilogues: fwd vs fwd_post, backward-data (chain mask) vs bwd_post, same data."""
import os, sys, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.realpath(__file__)))
sys.path.insert(0, ROOT)
import torch
import qcnn_amd
from qcnn_amd import functional as F, _lib as L
from ab_layers import SHAPES, timeit
dev = torch.device('cuda:0'); dt = torch.bfloat16
g = torch.Generator(device=dev).manual_seed(0)
for n in (sys.argv[1:] or ['c64', 'c32']):
    s = SHAPES[n]
    x = torch.relu(torch.randn(s['x'], device=dev, generator=g)).to(dt)
    w = torch.randn(s['w'], device=dev, generator=g) / 30
    b = torch.zeros(s['w'][-1], device=dev)
    lin = F.conv_call(tuple(s['x']), tuple(s['w']), dt, 2, 1, s['pad'], 'channels_last', 1, 'linear', True, s['conj'])
    rel = F.conv_call(tuple(s['x']), tuple(s['w']), dt, 2, 1, s['pad'], 'channels_last', 1, 'relu', True, s['conj'])
    lin.static_buffers = rel.static_buffers = True
    alpha = torch.full((s['x'][1],), 0.1, device=dev)
    for rate in (0.0, 0.3):
        post = F.PostOp(alpha, 0, rate, 1234)
        y = lin.fwd(x, w, b)
        dy = torch.randn(y.shape, device=dev, generator=g).to(dt)
        dal = torch.zeros(alpha.numel(), device=dev)
        fns = {'fwd_relu': lambda: rel.fwd(x, w, b, out=y), 'fwd_linear': lambda: lin.fwd(x, w, b, out=y),
               'fwd_post': lambda: lin.fwd_post(x, w, b, post),
               'bwd_chain_relu': lambda: rel.bwd(x, dy, y, w, True, flags=L.QK_BWD_MASK_DX | L.QK_BWD_DY_PREMASKED),
               'bwd_linear': lambda: lin.bwd(x, dy, None, w, True),
               'bwd_post': lambda: lin.bwd_post(x, dy, w, True, post, x, dal)}
        for k, fn in fns.items():
            fn(); torch.cuda.synchronize()
            t = timeit(fn, 5, 4)
            print('%-6s rate %.1f %-16s med %8.1f us' % (n, rate, k, statistics.median(t)))
