#!/usr/bin/env python
"""Power / shader-clock trace of ONE hot kernel on different operand VALUES ("these kernels are
power-bound").  The 64 -> 64 forward band kernel (716 800 x 256 x 3840, bf16) is launched back to back for a few
seconds per data set -- all zeros, relu(normal) (half zeros), relu + dropout 0.3 (65 % zeros: what the headline step
feeds it), dense normal -- while a sampler thread reads the GPU's socket power and shader clock (amdgpu hwmon / sysfs;
`rocm-smi --json` as the fall-back).  Same binary, same launch geometry: only the values differ.

    python tools/power_trace.py [--seconds 4] [--kernel fwd|bwd_data|bwd_weight] > gpurun_out/power/trace.txt
"""
import argparse, glob, json, os, subprocess, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import qcnn_amd
from qcnn_amd import functional as F


def find_hwmon():
    """sysfs directory (and its hwmon) of the GPU torch calls cuda:0 -- matched by PCI address: the host shows every
    card of the node, the container only computes on one."""
    want = None
    try:
        pr = torch.cuda.get_device_properties(0)
        want = '%04x:%02x:%02x' % (pr.pci_domain_id, pr.pci_bus_id, pr.pci_device_id)
    except Exception:
        pass
    cands = []
    for card in sorted(glob.glob('/sys/class/drm/card[0-9]*/device')):
        if not os.path.exists(os.path.join(card, 'pp_dpm_sclk')):
            continue
        hw = glob.glob(os.path.join(card, 'hwmon', 'hwmon*'))
        addr = os.path.basename(os.path.realpath(card))
        cands.append((card, hw[0] if hw else None, addr))
    for card, hw, addr in cands:
        if want and addr.lower().startswith(want):
            return card, hw
    return (cands[0][0], cands[0][1]) if len(cands) == 1 else (None, None)


class Sampler(threading.Thread):
    def __init__(self, period=0.05):
        super().__init__(daemon=True)
        self.period, self.rows, self.stop_flag = period, [], False
        self.card, self.hw = find_hwmon()
        self.src = 'sysfs' if self.card else 'rocm-smi'

    def read(self):
        if self.card:
            p = sclk = None
            for name in ('power1_average', 'power1_input'):
                f = os.path.join(self.hw or '', name)
                if self.hw and os.path.exists(f):
                    try:
                        p = int(open(f).read()) / 1e6
                        break
                    except Exception:
                        pass
            f = os.path.join(self.hw or '', 'freq1_input')
            if self.hw and os.path.exists(f):
                try:
                    sclk = int(open(f).read()) / 1e6
                except Exception:
                    pass
            if sclk is None:
                try:
                    for ln in open(os.path.join(self.card, 'pp_dpm_sclk')):
                        if '*' in ln:
                            sclk = float(ln.split(':')[1].strip().split('M')[0])
                except Exception:
                    pass
            return p, sclk
        try:
            out = subprocess.run(['rocm-smi', '--showpower', '--showclocks', '--json'], capture_output=True, text=True, timeout=5).stdout
            d = json.loads(out)
            c = d[sorted(d)[0]]
            p = next((float(v) for k, v in c.items() if 'ower' in k and 'W' in k), None)
            s = next((float(str(v).strip('()').lower().replace('mhz', '')) for k, v in c.items() if 'sclk' in k.lower() and 'level' not in k.lower()), None)
            return p, s
        except Exception:
            return None, None

    def run(self):
        while not self.stop_flag:
            p, s = self.read()
            self.rows.append((time.time(), p, s))
            time.sleep(self.period)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--seconds', type=float, default=4.0)
    ap.add_argument('--kernel', default='fwd', choices=['fwd', 'bwd_data', 'bwd_weight'])
    ap.add_argument('--cq', type=int, default=64)
    ap.add_argument('--fq', type=int, default=64)
    args = ap.parse_args()
    dev = torch.device('cuda:0')
    dt = torch.bfloat16
    g = torch.Generator(device=dev).manual_seed(0)
    xs, ws = (256, 14, 200, 4 * args.cq), (3, 5, args.cq, 4 * args.fq)
    ys = (256, 14, 200, 4 * args.fq)
    w = torch.randn(ws, device=dev, generator=g) / 30
    b = torch.zeros(4 * args.fq, device=dev)
    call = F.conv_call(xs, ws, dt, 2, 1, 'same', 'channels_last', 1, None, True, False)
    call.static_buffers = True
    in_shape = xs if args.kernel == 'fwd' else ys          # the tensor whose VALUES are varied: x (forward) or dy (backward)
    r = torch.randn(in_shape, device=dev, generator=g)
    keep = (torch.rand(in_shape, device=dev, generator=g) >= 0.3).float() / 0.7
    xfix = torch.relu(torch.randn(xs, device=dev, generator=g)).to(dt)
    data = [('all zeros', torch.zeros(in_shape, device=dev)), ('relu(normal): 50 % zeros', torch.relu(r)),
            ('relu + dropout 0.3: 65 % zeros', torch.relu(r) * keep), ('dense normal', r)]
    y = torch.empty(call.y_shape, dtype=dt, device=dev)
    dx = torch.empty(xs, dtype=dt, device=dev)
    dw, db = torch.zeros(ws, device=dev), torch.zeros(4 * args.fq, device=dev)
    flops = 2.0 * 716800 * (4 * args.fq) * (15 * 4 * args.cq)
    smp = Sampler()
    cap = None
    try:
        cap = int(open(os.path.join(smp.hw, 'power1_cap')).read()) / 1e6
    except Exception:
        pass
    smp.start()
    print('# kernel: %s of the %d -> %d body layer (716800 x %d x %d, bf16);' % (args.kernel, args.cq, args.fq, 4 * args.fq, 60 * args.cq) + ' sensor source: %s (%s); idle sample (W, MHz): %r; power cap: %s W' % (smp.src, smp.card, smp.read(), cap))
    print('# %-32s %9s %9s %10s %10s %8s' % ('operand values', 'us/launch', 'TFLOP/s', 'power W', 'sclk MHz', 'samples'))
    for name, x in data:
        x = x.to(dt)
        if args.kernel == 'fwd':
            fn = lambda: call.fwd(x, w, b, out=y)
        elif args.kernel == 'bwd_data':
            fn = lambda: call.bwd_data(x, None, w, out=dx)
        else:
            fn = lambda: call.bwd_weight(xfix, x, None, True, out=(dw, db), accumulate=True)
        for _ in range(20):
            fn()
        torch.cuda.synchronize()
        t0 = time.time()
        n, ev = 0, []
        while time.time() - t0 < args.seconds:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(50):
                fn()
            e1.record()
            e1.synchronize()
            ev.append(e0.elapsed_time(e1) / 50)
            n += 50
        t1 = time.time()
        rows = [(p, s) for (t, p, s) in smp.rows if t0 + 0.5 <= t <= t1]
        pw = [p for p, _ in rows if p is not None]
        sc = [s for _, s in rows if s is not None]
        us = 1e3 * sum(ev) / len(ev)
        print('  %-32s %9.1f %9.0f %10s %10s %8d' % (name, us, flops / (us * 1e-6) / 1e12, '%.0f' % (sum(pw) / len(pw)) if pw else 'n/a',
                                                  '%.0f' % (sum(sc) / len(sc)) if sc else 'n/a', len(rows)))
        time.sleep(1.0)
    smp.stop_flag = True


if __name__ == '__main__':
    main()
