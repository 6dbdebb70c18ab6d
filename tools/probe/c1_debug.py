"""Probe (not product code): where the fused first-layer kernels differ from the oracle (element coordinates)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import qcnn_amd
from oracle import oracle
sys.path.insert(0, 'tests')
from test_gpu_parity import _np_pool_h_same
Fq = qcnn_amd.functional
dev = 'cuda:0'
for shape, F, planes in [((3, 41, 50, 4), 32, False), ((3, 41, 50, 4), 32, True), ((2, 9, 230, 4), 64, False), ((1, 8, 19, 4), 32, False)]:
    dtype = torch.bfloat16
    rng = np.random.RandomState(53)
    rnd = lambda a: torch.tensor(a).to(dtype).double().numpy()
    x = rnd(rng.randn(*shape)); w = rnd(rng.randn(3, 5, 1, 4 * F) / np.sqrt(60.0))
    b = (0.1 * rng.randn(4 * F)).astype(np.float32).astype(np.float64)
    kw = dict(padding='same', activation='relu')
    y = oracle.forward(x, w, b, 2, **kw)
    pooled, arg = _np_pool_h_same(y)
    dp = rnd(rng.randn(*pooled.shape))
    dy = np.zeros_like(y)
    n, ho, wd, c = pooled.shape
    ii = np.meshgrid(np.arange(n), np.arange(ho), np.arange(wd), np.arange(c), indexing='ij')
    np.add.at(dy, (ii[0], arg, ii[2], ii[3]), dp)
    _, dw, db = oracle.backward(x, w, b, dy, 2, y=y, **kw)
    xt = torch.tensor(x, device=dev).to(dtype)
    wt = torch.tensor(w, device=dev, dtype=torch.float32, requires_grad=True)
    bt = torch.tensor(b, device=dev, dtype=torch.float32, requires_grad=True)
    lay = 'channels_last'
    if planes: xt, lay = xt.permute(0, 3, 1, 2).contiguous(), 'channels_first'
    out = Fq.conv_relu_pool(xt, wt, bt, 3, lay)
    out.backward(torch.tensor(dp, device=dev).to(dtype))
    o = out.detach().float().cpu().numpy()
    bad = np.abs(o - pooled) > 0.02 * (1 + np.abs(pooled))
    print(shape, F, planes, 'bad outputs', bad.sum(), 'of', bad.size)
    if bad.any():
        idx = np.argwhere(bad)
        for ax, nm in enumerate(['n', 'ho', 'w', 'c']):
            u, cnt = np.unique(idx[:, ax], return_counts=True)
            print('  ', nm, dict(zip(u.tolist()[:40], cnt.tolist()[:40])))
        for i in idx[:6]: print('   ', i, o[tuple(i)], pooled[tuple(i)])
    g = wt.grad.cpu().numpy(); e = np.abs(g - dw)
    print('  dw rel', np.abs(g - dw).max() / np.abs(dw).max(), 'db rel', np.abs(bt.grad.cpu().numpy() - db).max() / np.abs(db).max())
    if e.max() > 1e-4 * np.abs(dw).max():
        idx = np.argwhere(e > 1e-4 * np.abs(dw).max())
        for ax, nm in enumerate(['kh', 'kw', 'one', 'col']):
            u, cnt = np.unique(idx[:, ax], return_counts=True)
            print('  dw', nm, dict(zip(u.tolist()[:40], cnt.tolist()[:40])))
