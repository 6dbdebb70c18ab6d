#!/usr/bin/env python
"""Per-kernel timing of the TIMIT layer shapes (fwd / bwd-data / bwd-weight, bf16, B=256) for A/B runs:
    QK_LIB=<other .so> python tools/ab_layers.py [shape ...]
Prints one line per (shape, kernel): mean microseconds over `reps` back-to-back launches, best/median of rounds."""
import os, sys, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.realpath(__file__)))
sys.path.insert(0, ROOT)
import torch
import qcnn_amd
from qcnn_amd import functional as F

SHAPES = {
    'c64': dict(x=(256, 14, 200, 256), w=(3, 5, 64, 256), pad='same', conj=False),
    'c32': dict(x=(256, 14, 200, 128), w=(3, 5, 32, 128), pad='same', conj=False),
    'c16': dict(x=(256, 14, 200, 64), w=(3, 5, 16, 64), pad='same', conj=False),
    'c16to32': dict(x=(256, 14, 200, 64), w=(3, 5, 16, 128), pad='same', conj=False),
    'c32to64': dict(x=(256, 14, 200, 128), w=(3, 5, 32, 256), pad='same', conj=False),
    'head': dict(x=(256, 14, 200, 256), w=(14, 1, 64, 256), pad='valid', conj=True),
    'first': dict(x=(256, 41, 200, 128), w=(1, 1, 32, 128), pad='valid', conj=False),     # folded first layer (1x1, cq2=32)
    'dense64': dict(x=(51200, 1, 1, 256), w=(1, 1, 64, 256), pad='valid', conj=True),
}


def timeit(fn, reps, rounds):
    out = []
    for _ in range(rounds):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record(); e1.synchronize()
        out.append(1e3 * e0.elapsed_time(e1) / reps)
    return out


def main():
    names = sys.argv[1:] or list(SHAPES)
    dev = torch.device('cuda:0')
    dt = torch.bfloat16
    g = torch.Generator(device=dev).manual_seed(0)
    tag = os.environ.get('AB_TAG') or os.path.basename(os.environ.get('QK_LIB', 'libqk_hip.so'))
    for n in names:
        s = SHAPES[n]
        x = torch.randn(s['x'], device=dev, generator=g).to(dt)
        w = torch.randn(s['w'], device=dev, generator=g) / 30
        b = torch.zeros(s['w'][-1], device=dev)
        for act in ('relu', 'linear'):
            call = F.conv_call(tuple(s['x']), tuple(s['w']), dt, 2, 1, s['pad'], 'channels_last', 1, act, True, s['conj'])
            call.static_buffers = True
            y = call.fwd(x, w, b)
            dy = torch.randn(y.shape, device=dev, generator=g).to(dt)
            dx = torch.empty_like(x)
            dw = torch.empty_like(w); db = torch.empty_like(b)
            fns = {'fwd': lambda: call.fwd(x, w, b, out=y),
                   'bwd_data': lambda: call.bwd_data(dy, y, w, out=dx),
                   'bwd_weight': lambda: call.bwd_weight(x, dy, y, True, out=(dw, db))}
            flops = 2.0 * y.numel() // s['w'][-1] * s['w'][-1] * s['w'][0] * s['w'][1] * 4 * s['w'][2]
            for k, fn in fns.items():
                fn(); torch.cuda.synchronize()
                t = timeit(fn, 5, 5)
                print('%-14s %-8s %-7s %-10s med %8.1f us  min %8.1f us  %7.1f TF' % (tag, n, act, k, statistics.median(t), min(t), flops / min(t) / 1e6))
            if n in ('first', 'dense64'):
                break


if __name__ == '__main__':
    main()
