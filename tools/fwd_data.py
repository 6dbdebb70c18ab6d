"""Does the forward band kernel's time depend on the operand values?  (same launch, different x)"""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import qcnn_amd
from qcnn_amd import functional as F
from ab_layers import timeit
dev = torch.device('cuda:0')
dt = torch.bfloat16
g = torch.Generator(device=dev).manual_seed(0)
xs, ws = (256, 14, 200, 256), (3, 5, 64, 256)
w = torch.randn(ws, device=dev, generator=g) / 30
b = torch.zeros(256, device=dev)
call = F.conv_call(xs, ws, dt, 2, 1, 'same', 'channels_last', 1, 'relu', True, False)
call.static_buffers = True
r = torch.randn(xs, device=dev, generator=g)
data = {'dense normal': r, 'relu(normal): half zeros, non-negative': torch.relu(r), 'normal * mask: half zeros, both signs': r * (torch.randn(xs, device=dev, generator=g) > 0),
        'all zeros': torch.zeros(xs, device=dev), '|normal|: dense, non-negative': r.abs(), '0.01 * relu(normal)': 0.01 * torch.relu(r)}
y = None
for rep in range(2):
    for name, x in data.items():
        x = x.to(dt)
        y = call.fwd(x, w, b) if y is None else y
        fn = lambda: call.fwd(x, w, b, out=y)
        fn(); torch.cuda.synchronize()
        t = timeit(fn, 10, 4)
        print('%-42s med %7.1f us  min %7.1f us' % (name, statistics.median(t), min(t)))
