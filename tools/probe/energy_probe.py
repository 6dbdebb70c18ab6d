#!/usr/bin/env python
"""Driver of tools/probe/energy_probe.hip: each inner-loop variant launched back to back for a few seconds on zero and on
random bf16 operands while the socket power and the shader clock are sampled (tools/power_trace.py's sampler).

    hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC tools/probe/energy_probe.hip -o tools/probe/libenergy_probe.so
    python tools/probe/energy_probe.py [--seconds 3] > gpurun_out/r4_energy_probe.txt

Columns: TFLOP/s (bf16 MFMA work only), W, MHz, and the derived energy figures
    pJ/FLOP  = W / (FLOP/s)                       (everything the socket draws, per useful FLOP)
    W/GHz    = (W - W_idle) / f                   (dynamic energy per shader cycle)
    busy     = MFMA issue cycles / shader cycles  = (MFMAs per SIMD * 32) / (f * t)
"""
import argparse, ctypes, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tools'))
import torch
from power_trace import Sampler


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--seconds', type=float, default=3.0)
    ap.add_argument('--iters', type=int, default=1000)
    args = ap.parse_args()
    lib = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'libenergy_probe.so'))
    lib.probe_launch.restype = ctypes.c_double
    lib.probe_launch.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_uint, ctypes.c_void_p, ctypes.c_void_p]
    dev = torch.device('cuda:0')
    units = 1 << 22                                                     # 64 MB of 16-byte units
    g = torch.Generator(device=dev).manual_seed(0)
    rnd = torch.randn(units * 8, device=dev, generator=g).to(torch.bfloat16)
    relu = (torch.relu(torch.randn(units * 8, device=dev, generator=g)) * (torch.rand(units * 8, device=dev, generator=g) >= 0.3)).to(torch.bfloat16)
    zero = torch.zeros(units * 8, device=dev, dtype=torch.bfloat16)
    out = torch.zeros(256 * 2 * 256, device=dev)
    smp = Sampler()
    smp.start()
    time.sleep(1.0)
    idle = smp.read()
    p_idle = idle[0] or 0.0
    print('# sensor: %s (%s); idle sample (W, MHz): %r' % (smp.src, smp.card, idle))
    print('# %-58s %-22s %9s %9s %8s %8s %8s %8s %6s' % ('variant', 'operands', 'us/launch', 'TFLOP/s', 'W', 'MHz', 'pJ/FLOP', 'W/GHz', 'busy'))
    variants = [
        ('idle: resident waves in s_sleep', 1, 9, 2),
        ('A  MFMA only, 32-row tiles, 2 waves/SIMD (2 WG x 4 waves)', 1, 0, 2),
        ('A1 MFMA only, 32-row tiles, 1 wave/SIMD', 1, 0, 1),
        ('B  + LDS fragment reads (0.50 b128/MFMA), 2 waves/SIMD', 1, 1, 2),
        ('C  + staging (global loads + LDS stores), 2 waves/SIMD', 1, 2, 2),
        ('D  + staging by LDS-DMA (global_load_lds_dwordx4), 2 waves/SIMD', 1, 3, 2),
        ('A2 MFMA only, 64-row tiles, 1 wave/SIMD', 2, 0, 1),
        ('B2 + LDS fragment reads (0.375 b128/MFMA), 1 wave/SIMD', 2, 1, 1),
        ('C2 + staging, 64-row tiles, 1 wave/SIMD', 2, 2, 1),
        ('D2 + staging by LDS-DMA, 64-row tiles, 1 wave/SIMD', 2, 3, 1),
    ]
    stream = torch.cuda.current_stream().cuda_stream
    for name, rb, kind, wgs in variants:
        for dname, src in (('zeros', zero), ('relu+dropout (65 % zeros)', relu), ('dense normal', rnd)):
            if kind == 9 and dname != 'zeros':
                continue
            it = args.iters if kind != 9 else 2000
            fn = lambda: lib.probe_launch(rb, kind, wgs, it, src.data_ptr(), units, out.data_ptr(), stream)
            for _ in range(10):
                n_mfma = fn()
            torch.cuda.synchronize()
            t0 = time.time()
            ev = []
            while time.time() - t0 < args.seconds:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(50):
                    fn()
                e1.record()
                e1.synchronize()
                ev.append(e0.elapsed_time(e1) / 50)
            t1 = time.time()
            rows = [(p, s) for (t, p, s) in smp.rows if t0 + 0.6 <= t <= t1]
            pw = [p for p, _ in rows if p is not None]
            sc = [s for _, s in rows if s is not None]
            us = 1e3 * sum(ev) / len(ev)
            w = sum(pw) / len(pw) if pw else float('nan')
            mhz = sum(sc) / len(sc) if sc else float('nan')
            flops = n_mfma * 32768.0
            tf = flops / (us * 1e-6) / 1e12
            busy = (n_mfma / 1024.0 * 32.0) / (mhz * 1e6 * us * 1e-6) if n_mfma else 0.0
            print('  %-58s %-22s %9.1f %9.0f %8.0f %8.0f %8.3f %8.0f %6.2f' % (name, dname[:22], us, tf, w, mhz, (w / (flops / (us * 1e-6)) * 1e12) if flops else 0.0,
                                                                           (w - p_idle) / (mhz / 1e3), busy))
            sys.stdout.flush()
            time.sleep(0.8)
    smp.stop_flag = True


if __name__ == '__main__':
    main()
