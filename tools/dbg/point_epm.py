import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, numpy as np
import qcnn_amd
from qcnn_amd import _lib
F = qcnn_amd.functional
dev = torch.device('cuda:0')
torch.manual_seed(0)
for dtype in (torch.bfloat16,):
    for (cq, fq) in ((64, 32), (64, 64), (128, 64)):
        x = torch.randn(2, 6, 40, 4 * cq, device=dev).to(dtype).requires_grad_(True)
        w0 = (torch.randn(3, 5, cq, 4 * cq, device=dev) / 60).requires_grad_(True)
        w1 = (torch.randn(6, 1, cq, 4 * fq, device=dev) / 40).requires_grad_(True)
        def run():
            for t in (x, w0, w1):
                t.grad = None
            y = F.quaternion_conv_chain(x, [(w0, None, dict(padding='same', activation='relu')), (w1, None, dict(padding='valid', activation='relu', conj=True))])
            y.backward(torch.ones_like(y))
            torch.cuda.synchronize()
            return [t.grad.float().clone() for t in (x, w0, w1)]
        a = run()
        with _lib.debug_flags(_lib.QK_DBG_NO_POINT16):
            b = run()
        for n, u, v in zip(('dx', 'dw0', 'dw1'), a, b):
            bad = ~torch.isfinite(u)
            print(cq, fq, n, 'nan', int(bad.sum()), 'maxdiff', float((u - v)[~bad].abs().max()), 'ref max', float(v.abs().max()))
            if bad.any():
                idx = bad.nonzero()
                print('  first bad', idx[:5].tolist(), 'last-axis set', sorted(set(idx[:, -1].tolist()))[:20])
print('--- 3-layer chain with biases')
rng = np.random.RandomState(23)
dtype = torch.bfloat16
specs = [((3, 5, 32, 128), dict(padding='same', activation='relu')),
         ((3, 5, 32, 256), dict(padding='same', activation='relu')),
         ((6, 1, 64, 128), dict(padding='valid', activation='relu', conj=True))]
x = torch.tensor(rng.randn(2, 6, 40, 128).astype(np.float32), device=dev).to(dtype).requires_grad_(True)
wt = [torch.tensor((rng.randn(*s) / np.sqrt(np.prod(s[:-1]) * 4)).astype(np.float32), device=dev).requires_grad_(True) for s, _ in specs]
bt = [torch.tensor((0.1 * rng.randn(s[-1])).astype(np.float32), device=dev).requires_grad_(True) for s, _ in specs]
def run3():
    for t in [x] + wt + bt:
        t.grad = None
    y = F.quaternion_conv_chain(x, [(w, b, kw) for w, b, (_, kw) in zip(wt, bt, specs)])
    torch.manual_seed(1)
    y.backward(torch.randn(y.shape, device=dev).to(dtype))
    torch.cuda.synchronize()
    return [t.grad.float().clone() for t in [x] + wt + bt]
for rep in range(3):
    a = run3()
    with _lib.debug_flags(_lib.QK_DBG_NO_POINT16):
        b = run3()
    for n, u, v in zip(('dx', 'dw0', 'dw1', 'dw2', 'db0', 'db1', 'db2'), a, b):
        bad = ~torch.isfinite(u)
        print(rep, n, 'nan', int(bad.sum()), 'nan(ref)', int((~torch.isfinite(v)).sum()), 'maxdiff', float((u - v)[~bad].abs().max()), 'ref max', float(v.abs().max()))
