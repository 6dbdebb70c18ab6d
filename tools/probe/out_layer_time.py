"""The TIMIT output layer, Dense(62, softmax) on 51 200 x 256 rows (B = 256): the hand-written kernels (qk_dense_softmax_fwd / _bwd, round 6)
against the round-4/5 composition (library GEMMs + qk_softmax_rows_*; QK_DBG_NO_FUSED_SOFTMAX selects... the torch path, so the composition is
called directly here), alternating, HIP events.  Usage: python tools/probe/out_layer_time.py [rows]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import qcnn_amd                                                              # noqa: E402
from qcnn_amd import functional as Fq                                        # noqa: E402
from qcnn_amd.layers import _TallDenseFn                                     # noqa: E402


def timeit(fn, reps=20, rounds=5):
    best, tot = 1e9, 0.0
    for _ in range(rounds):
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(reps):
            fn()
        b.record()
        torch.cuda.synchronize()
        ms = a.elapsed_time(b) / reps
        best, tot = min(best, ms), tot + ms
    return tot / rounds, best


def main():
    rows = int(sys.argv[1]) if len(sys.argv) > 1 else 51200
    dev = torch.device('cuda:0')
    g = torch.Generator(device=dev).manual_seed(0)
    for dt in (torch.bfloat16, torch.float16):
        x = torch.randn(rows, 256, device=dev, generator=g).to(dt)
        w = (torch.randn(256, 62, device=dev, generator=g) * 0.2)
        b = torch.randn(62, device=dev, generator=g) * 0.1
        dy = (torch.randn(rows, 62, device=dev, generator=g) * 0.01).to(dt)
        dw, db = torch.zeros_like(w), torch.zeros_like(b)
        y = Fq.dense_softmax_fwd(x, w, b)
        w16 = w.to(dt)

        def old_fwd():
            w16_ = w.to(dt)
            return Fq.softmax_rows_fwd(torch.mm(x, w16_, out_dtype=torch.float32), b, dt)

        def old_bwd():
            dl = Fq.softmax_rows_bwd(y, dy, db)
            dx = dl @ w16.t()
            s = _TallDenseFn.SPLITS
            dwp = torch.bmm(x.view(s, -1, 256).transpose(1, 2), dl.view(s, -1, 62), out_dtype=torch.float32).sum(0)
            return dx, dwp
        y_old = old_fwd()
        print('%s rows %d: max |y_new - y_old| = %.3g' % (dt, rows, float((y.float() - y_old.float()).abs().max())))
        for name, fn in (('fwd  new', lambda: Fq.dense_softmax_fwd(x, w, b)), ('fwd  old', old_fwd),
                         ('bwd  new', lambda: Fq.dense_softmax_bwd(x, w, y, dy, dw, db)), ('bwd  old', old_bwd),
                         ('bwd  new, no dW / db', lambda: Fq.dense_softmax_bwd(x, w, y, dy, None, None))):
            ms, mn = timeit(fn)
            print('  %-22s %8.1f us (min %7.1f)' % (name, 1e3 * ms, 1e3 * mn))
        nb_f = rows * 256 * 2 + rows * 62 * 2
        nb_b = 2 * rows * 256 * 2 + 2 * rows * 62 * 2
        print('  algorithmic bytes: fwd %.1f MB, bwd %.1f MB' % (nb_f / 1e6, nb_b / 1e6))


if __name__ == '__main__':
    main()
