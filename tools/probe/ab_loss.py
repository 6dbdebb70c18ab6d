#!/usr/bin/env python
"""Same process, same box, alternating: the default bench step under the linear stand-in loss, under the CTC cost, and under the
CTC cost scaled by 2^12 (same gradient DISTRIBUTION, different magnitude) -- step time, the 64 -> 64 kernels' in-step times and
the socket power / shader clock while each runs (bench.GpuTelemetry).  Round 4: why is the CTC step 1.5 ms slower?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, bench
dev = torch.device('cuda:0')
mk = lambda l: bench.ModelTrainStep(dict(bench.WORKLOADS[bench.DEFAULT_WORKLOAD], activation='relu'), dev, 0, 1, loss=l)
jobs = {'sum': mk('sum'), 'ctc': mk('ctc'), 'ctc x 4096': mk('ctc')}
orig = jobs['ctc x 4096'].model.ctc_loss
jobs['ctc x 4096'].model.ctc_loss = lambda *a: orig(*a) * 4096.0
for j in jobs.values():
    for _ in range(3): j.step()
torch.cuda.synchronize()
for rep in range(2):
    for l, j in jobs.items():
        for _ in range(10): j.step()
        tele = bench.GpuTelemetry(dev); tele.start()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(60): j.step()
        torch.cuda.synchronize(); t1 = time.perf_counter()
        tl = tele.summary(t0, t1) or {}
        r = bench.in_step_kernel_times(j, dev, 2500.0)
        top = {(c['op'], c['n'], c['k']): c['ms'] for c in r['calls']}
        print('%d %-11s ms/step %.3f  64->64 fwd %.3f bwd_data %.3f bwd_weight %.3f   %s W  %s MHz' % (
            rep, l, 1e3 * (t1 - t0) / 60, top[('fwd', 256, 3840)], top[('bwd_data', 256, 3840)], top[('bwd_weight', 256, 3840)],
            '%.0f' % tl['mean_socket_w'] if tl.get('mean_socket_w') else 'n/a', '%.0f' % tl['mean_sclk_mhz'] if tl.get('mean_sclk_mhz') else 'n/a'))
