"""Probe (not product code): stand-alone time of the fused first-layer kernels at the benchmarked size (B = 256, planes in)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, qcnn_amd
Fq = qcnn_amd.functional
dev = 'cuda:0'
torch.manual_seed(0)
x = torch.randn(256, 4, 41, 200, device=dev).bfloat16()
w = (torch.randn(3, 5, 1, 128, device=dev) / 60 ** 0.5).requires_grad_()
b = (0.1 * torch.randn(128, device=dev)).requires_grad_()
out = Fq.conv_relu_pool(x, w, b, 3, 'channels_first')
dp = torch.randn_like(out)
def t(fn, n=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
    ev[0].record()
    for i in range(n): fn(); ev[i + 1].record()
    torch.cuda.synchronize()
    d = sorted(ev[i].elapsed_time(ev[i + 1]) for i in range(n))
    return d[n // 2] * 1e3, d[0] * 1e3
with torch.no_grad():
    print('fwd  median %.1f us  min %.1f us' % t(lambda: Fq.conv_relu_pool(x, w, b, 3, 'channels_first')))
def fb():
    o = Fq.conv_relu_pool(x, w, b, 3, 'channels_first'); o.backward(dp)
print('fwd+bwd median %.1f us  min %.1f us' % t(fb))
