#!/usr/bin/env python
"""Driver of tools/probe/operand_probe.hip: the band kernels' MFMA stream with the kernel fragment as the FIRST operand (today) or as the
SECOND, on (kernel values, activation values) = (dense, relu + dropout) -- the training step's case -- and the symmetric cases; socket power and
shader clock sampled (tools/power_trace.py's sampler).  python tools/probe/operand_probe.py [--seconds 3]"""
import argparse, ctypes, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tools'))
import torch
from power_trace import Sampler

ap = argparse.ArgumentParser()
ap.add_argument('--seconds', type=float, default=3.0)
args = ap.parse_args()
lib = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'liboperand_probe.so'))
lib.op_launch.restype = ctypes.c_double
lib.op_launch.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint, ctypes.c_void_p, ctypes.c_void_p]
dev = torch.device('cuda:0')
units = 1 << 20
g = torch.Generator(device=dev).manual_seed(0)
dense = (torch.randn(units * 8, device=dev, generator=g) / 8).to(torch.bfloat16)
relu = (torch.relu(torch.randn(units * 8, device=dev, generator=g)) * (torch.rand(units * 8, device=dev, generator=g) >= 0.3)).to(torch.bfloat16)
zero = torch.zeros(units * 8, device=dev, dtype=torch.bfloat16)
out = torch.zeros(512 * 256, device=dev)
smp = Sampler()
smp.start()
time.sleep(1.0)
print('# sensor: %s (%s); idle sample (W, MHz): %r' % (smp.src, smp.card, smp.read()))
print('# %-34s %-44s %9s %8s %8s' % ('operand order', 'kernel values / activation values', 'TFLOP/s', 'W', 'MHz'))
stream = torch.cuda.current_stream().cuda_stream
cases = [('dense / relu+dropout (the step)', dense, relu), ('dense / dense', dense, dense), ('relu+dropout / relu+dropout', relu, relu),
         ('relu+dropout / dense (roles exchanged)', relu, dense), ('zeros / zeros', zero, zero)]
for rep in range(2):
    for cname, w, x in cases:
        for order, oname in ((0, 'mfma(kernel, activation)  [today]'), (1, 'mfma(activation, kernel)')):
            fn = lambda: lib.op_launch(order, 2000, w.data_ptr(), x.data_ptr(), units, out.data_ptr(), stream)
            for _ in range(5):
                n = fn()
            torch.cuda.synchronize()
            t0 = time.time()
            ev = []
            while time.time() - t0 < args.seconds:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(20):
                    fn()
                e1.record()
                e1.synchronize()
                ev.append(e0.elapsed_time(e1) / 20)
            t1 = time.time()
            rows = [(p, s) for (t, p, s) in smp.rows if t0 + 0.6 <= t <= t1]
            pw = [p for p, _ in rows if p is not None]
            sc = [s for _, s in rows if s is not None]
            us = 1e3 * sum(ev) / len(ev)
            print('  %-34s %-44s %9.0f %8.0f %8.0f' % (oname, cname, n * 32768.0 / (us * 1e-6) / 1e12, sum(pw) / max(len(pw), 1), sum(sc) / max(len(sc), 1)))
            sys.stdout.flush()
            time.sleep(0.5)
smp.stop_flag = True
