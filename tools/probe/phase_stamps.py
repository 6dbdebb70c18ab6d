#!/usr/bin/env python
"""Where a tile's time goes in the 16-bit band kernels: per-workgroup shader-clock stamps (start, prologue done, K loop done, end) from
a probe build of the library (tools/probe/build_variant.sh stamps -DQK_PHASE_STAMPS; QK_LIB points at it):

    bash tools/probe/build_variant.sh stamps -DQK_PHASE_STAMPS && QK_LIB=$PWD/tools/probe/libqk_stamps.so python tools/probe/phase_stamps.py [c64 c32 c32to64]

Prints, per shape and direction, the mean cycles of the three phases of a workgroup, the ideal K-loop time (its MFMAs x 32 cycles x the 2 waves
that share a SIMD) and the kernel's span in cycles."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tools'))
import torch
import qcnn_amd
from qcnn_amd import functional as F, _lib
from ab_layers import SHAPES

dev = torch.device('cuda:0')
dt = torch.bfloat16
g = torch.Generator(device=dev).manual_seed(0)
buf = torch.zeros(8 * 65536, dtype=torch.int64, device=dev)
for n in (sys.argv[1:] or ['c64', 'c32', 'c32to64']):
    s = SHAPES[n]
    keep = (torch.rand(s['x'], device=dev, generator=g) >= 0.3)
    x = (torch.relu(torch.randn(s['x'], device=dev, generator=g)) * keep).to(dt)           # relu + dropout: what the step feeds the kernels
    w = torch.randn(s['w'], device=dev, generator=g) / 30
    nobias = bool(os.environ.get('STAMPS_NO_BIAS'))
    b = None if nobias else torch.zeros(s['w'][-1], device=dev)
    call = F.conv_call(tuple(s['x']), tuple(s['w']), dt, 2, 1, s['pad'], 'channels_last', 1, 'linear', not nobias, s['conj'])
    call.static_buffers = True
    y = call.fwd(x, w, b)
    dy = (torch.randn(y.shape, device=dev, generator=g) * (torch.rand(y.shape, device=dev, generator=g) >= 0.65)).to(dt)
    dx = torch.empty_like(x)
    for name, fn in (('fwd', lambda: call.fwd(x, w, b, out=y)), ('bwd_data', lambda: call.bwd_data(dy, y, w, out=dx))):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        buf.zero_()
        _lib.lib().qk_set_debug_buffer(buf.data_ptr(), buf.numel() * 8)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); e1.synchronize()
        _lib.lib().qk_set_debug_buffer(None, 0)
        t = buf.view(-1, 8).cpu()
        t = t[t[:, 3] > 0]
        pro, loop, epi = (t[:, 1] - t[:, 0]).float(), (t[:, 2] - t[:, 1]).float(), (t[:, 3] - t[:, 2]).float()
        if int(t[:, 5].max()) > 0:        # finer prologue stamps: loads issued, band in LDS
            print('         prologue: start -> loads issued %.0f -> band stored %.0f -> barrier passed %.0f cycles' % (
                float((t[:, 5] - t[:, 0]).float().mean()), float((t[:, 6] - t[:, 5]).float().mean()), float((t[:, 1] - t[:, 6]).float().mean())))
        span = int(t[:, 3].max() - t[:, 0].min())
        taps, cq, f4 = s['w'][0] * s['w'][1], s['w'][2], s['w'][3]
        mfma_per_wave = taps * (cq // 32) * 2 * 16            # per tile: sub-steps x 2 ks x 16 MFMAs (a wave tile is 32 rows x 128 columns)
        # residency per CU: how long 2 / 1 / 0 workgroups were resident, and were in their K loop (sweep over the stamps of each CU)
        hw = t[:, 4]
        cu = ((hw >> 32) & 0xf) * 4096 + ((hw >> 13) & 7) * 512 + ((hw >> 12) & 1) * 256 + ((hw >> 8) & 0xf)      # xcc | se | sh | cu
        res, inloop, tot = [0.0, 0.0, 0.0], [0.0, 0.0, 0.0], 0.0
        for c in cu.unique().tolist():
            q = t[cu == c]
            ev = sorted([(int(a), 0, +1) for a in q[:, 0]] + [(int(a), 0, -1) for a in q[:, 3]] + [(int(a), 1, +1) for a in q[:, 1]] + [(int(a), 1, -1) for a in q[:, 2]])
            nn = [0, 0]
            last = ev[0][0]
            for when, kind, d in ev:
                res[min(nn[0], 2)] += when - last
                inloop[min(nn[1], 2)] += when - last
                last = when
                nn[kind] += d
            tot += ev[-1][0] - ev[0][0]
        print('         CUs %d  resident workgroups 2/1/0: %.1f / %.1f / %.1f %%   in K loop 2/1/0: %.1f / %.1f / %.1f %%' % (
            len(cu.unique()), 100 * res[2] / tot, 100 * res[1] / tot, 100 * res[0] / tot, 100 * inloop[2] / tot, 100 * inloop[1] / tot, 100 * inloop[0] / tot))
        print('%-8s %-9s workgroups %5d  us %7.1f  span %8d cyc (%.2f GHz)   prologue %6.0f   K loop %6.0f (ideal 2 waves/SIMD: %6d = %.0f %%)   epilogue %6.0f   sum %6.0f' % (
            n, name, len(t), 1e3 * e0.elapsed_time(e1), span, span / (1e3 * e0.elapsed_time(e1)) / 1e3, float(pro.mean()), float(loop.mean()),
            mfma_per_wave * 64, 100.0 * mfma_per_wave * 64 / float(loop.mean()), float(epi.mean()), float((pro + loop + epi).mean())))
