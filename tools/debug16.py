import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import qcnn_amd
from oracle import oracle
F = qcnn_amd.functional
dev = torch.device('cuda:0')
def rel(a, b): return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))
def run(rank, xs, ws, kw, dtype):
    rng = np.random.RandomState(11)
    x = torch.tensor(rng.randn(*xs).astype(np.float32)).to(dtype).float().numpy()
    w = (rng.randn(*ws) / np.sqrt(np.prod(ws[:-1]) * 4)).astype(np.float32)
    wr = torch.tensor(w).to(dtype).float().numpy()
    b = (0.1 * rng.randn(ws[-1])).astype(np.float32)
    y = oracle.forward(x, wr, b, rank, **kw)
    dy = torch.tensor(rng.randn(*y.shape).astype(np.float32)).to(dtype).float().numpy()
    dx, dw, db = oracle.backward(x, wr, b, dy, rank, y=y, **kw)
    xt = torch.tensor(x, device=dev).to(dtype).requires_grad_(True)
    wt = torch.tensor(w, device=dev).requires_grad_(True)
    bt = torch.tensor(b, device=dev).requires_grad_(True)
    if rank == 0: yt = F.quaternion_dense(xt, wt, bt, activation=kw['activation'])
    else: yt = F.quaternion_conv(xt, wt, bt, **kw)
    yt.backward(torch.tensor(dy, device=dev).to(dtype))
    torch.cuda.synchronize()
    print(rank, xs, ws, kw, 'y %.2e dx %.2e dw %.2e db %.2e' % (rel(yt.detach().float().cpu().numpy(), y), rel(xt.grad.float().cpu().numpy(), dx), rel(wt.grad.cpu().numpy(), dw), rel(bt.grad.cpu().numpy(), db)))
for act in ('relu', None):
    run(2, (2, 14, 40, 128), (3, 5, 32, 128), dict(padding='same', activation=act), torch.bfloat16)
    run(1, (2, 40, 128), (1, 32, 128), dict(padding='same', activation=act), torch.bfloat16)
    run(1, (2, 40, 128), (3, 32, 128), dict(padding='same', activation=act), torch.bfloat16)
    run(0, (300, 128), (32, 128), dict(activation=act), torch.bfloat16)
    run(1, (2, 40, 256), (1, 64, 256), dict(padding='same', activation=act), torch.bfloat16)
    run(1, (2, 40, 128), (1, 32, 256), dict(padding='same', activation=act), torch.bfloat16)
