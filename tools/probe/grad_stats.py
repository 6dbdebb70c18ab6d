#!/usr/bin/env python
"""Why is the step slower under the CTC cost than under the linear stand-in loss?  Statistics of the gradient tensors the
backward kernels of the default bench workload consume under both losses (zero fraction, magnitude, exponent spread)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import bench
from qcnn_amd import functional as Fq


def stats(name, t):
    f = t.detach().float()
    nz = f != 0
    a = f.abs()[nz]
    ex = torch.log2(a) if a.numel() else torch.zeros(1, device=f.device)
    print('    %-34s shape %-22s zeros %.3f  mean|v| %.3e  max %.3e  log2|v|: mean %.1f std %.1f' % (
        name, tuple(t.shape), 1.0 - float(nz.float().mean()), float(a.mean()) if a.numel() else 0.0, float(a.max()) if a.numel() else 0.0,
        float(ex.mean()), float(ex.std())))


dev = torch.device('cuda:0')
for loss in ('sum', 'ctc'):
    job = bench.ModelTrainStep(dict(bench.WORKLOADS[bench.DEFAULT_WORKLOAD], activation='relu'), dev, 0, 1, loss=loss)
    for _ in range(3):
        job.step()
    torch.cuda.synchronize()
    print('loss = %s' % loss)
    orig = Fq._ConvChainFn.backward
    def spy(ctx, dy, _orig=orig):
        stats('d(chain output)', dy)
        out = _orig(ctx, dy)
        stats('d(chain input)', out[0])
        acts = ctx.saved_tensors[:len(ctx.calls) + 1]
        stats('activation into layer 6 (64->64)', acts[6])
        return out
    Fq._ConvChainFn.backward = staticmethod(spy)
    try:
        job.step()
        torch.cuda.synchronize()
    finally:
        Fq._ConvChainFn.backward = orig
    # (kernel times are NOT taken here: the statistics passes above run inside the backward and disturb them -- ab_loss.py times both losses)
    del job
    torch.cuda.empty_cache()
