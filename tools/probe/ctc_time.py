"""qk_ctc_batch_cost at the benchmarked size (B = 256, T = 200, C = 62, 20-50 labels): the concurrent-sweep kernel (k_ctc_fast; round 6: linear-domain
recursion with lagged rescaling) and the two-sweep form, HIP events.  python tools/probe/ctc_time.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import qcnn_amd                                                              # noqa: E402
from qcnn_amd import _lib, functional as Fq                                  # noqa: E402

dev = torch.device('cuda:0')
g = torch.Generator().manual_seed(0)
B, T, C, L = 256, 200, 62, 50
pred = torch.softmax(torch.randn(B, T, C, generator=g) * 2, -1).to(dev).to(torch.bfloat16)
labels = torch.randint(0, C - 1, (B, L), generator=g).to(dev, torch.int32)
ll = torch.randint(20, L + 1, (B, 1), generator=g).to(dev, torch.int32)
il = torch.full((B, 1), int(os.environ.get('CTC_IL', T)), dtype=torch.int32, device=dev)     # CTC_IL=<frames that count>: how the time scales with the sweep length
ll = torch.minimum(ll, il // 2 - 1).clamp(min=1)


def t(fn, reps=20, rounds=5):
    out = []
    for _ in range(rounds):
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(reps):
            fn()
        b.record()
        torch.cuda.synchronize()
        out.append(1e3 * a.elapsed_time(b) / reps)
    return sum(out) / len(out), min(out)


c0, g0 = Fq.ctc_cost_and_grad(pred, labels, il, ll)
with _lib.debug_flags(_lib.QK_DBG_CTC_TWO_SWEEPS):
    c1, g1 = Fq.ctc_cost_and_grad(pred, labels, il, ll)
print('cost: max |fast - two_sweeps| / max = %.3g   gradient: %.3g' % (float((c0 - c1).abs().max() / c1.abs().max()),
                                                                        float((g0.float() - g1.float()).abs().max() / g1.float().abs().max())))
print('concurrent sweeps  %7.1f us (min %7.1f)' % t(lambda: Fq.ctc_cost_and_grad(pred, labels, il, ll)))
with _lib.debug_flags(_lib.QK_DBG_CTC_TWO_SWEEPS):
    print('two sweeps         %7.1f us (min %7.1f)' % t(lambda: Fq.ctc_cost_and_grad(pred, labels, il, ll)))
