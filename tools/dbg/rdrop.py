import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), 'tests'))
import numpy as np, torch
import qcnn_amd
from oracle import oracle
from test_gpu_parity import _np_drop_factor, _rel_err
F = qcnn_amd.functional
dev = torch.device('cuda:0')
dtype = torch.bfloat16
def run(specs, xshape, rate, tag):
    rng = np.random.RandomState(49)
    rnd = lambda a: torch.tensor(a).to(dtype).double().numpy()
    x = rnd(rng.randn(*xshape).astype(np.float32).astype(np.float64))
    ws = [rnd((rng.randn(*s) / np.sqrt(np.prod(s[:-1]) * 4)).astype(np.float32).astype(np.float64)) for s, _ in specs]
    bs = [(0.1 * rng.randn(s[-1])).astype(np.float32).astype(np.float64) for s, _ in specs]
    n = len(specs)
    seeds = [111 * (i + 1) for i in range(n)]
    acts, keeps = [x], []
    for w, b, (_, kw), sd in zip(ws, bs, specs, seeds):
        pre = oracle.forward(acts[-1], w, b, 2, **kw)
        keep = _np_drop_factor(pre.shape, sd, rate)
        keeps.append(keep)
        acts.append(rnd(np.maximum(pre, 0) * keep))
    dy = rnd(rng.randn(*acts[-1].shape).astype(np.float32).astype(np.float64))
    xt = torch.tensor(x, device=dev).to(dtype).requires_grad_(True)
    wt = [torch.nn.Parameter(torch.tensor(w, device=dev, dtype=torch.float32)) for w in ws]
    bt = [torch.nn.Parameter(torch.tensor(b, device=dev, dtype=torch.float32)) for b in bs]
    layers = [(wt[i], bt[i], dict(specs[i][1], post=dict(alpha=None, rate=rate, seed=seeds[i]))) for i in range(n)]
    taps = []
    F.chain_tap = taps.append
    y = F.quaternion_conv_chain(xt, layers)
    F.chain_tap = None
    y.backward(torch.tensor(dy, device=dev).to(dtype))
    print(tag, 'y', _rel_err(y.detach().float().cpu().numpy(), acts[-1]), 'paths', qcnn_amd._lib.last_path())
    for i in range(n):
        print('   act', i + 1, _rel_err(taps[0][i + 1].detach().float().cpu().numpy(), acts[i + 1]))
    g = dy
    for i in reversed(range(n)):
        dpre = g * keeps[i] * (acts[i + 1] > 0)
        g, dw, db = oracle.backward(acts[i], ws[i], bs[i], dpre, 2, **specs[i][1])
        print('   layer', i, 'dw', _rel_err(wt[i].grad.cpu().numpy(), dw), 'db', _rel_err(bt[i].grad.cpu().numpy(), db))
    print('   dx', _rel_err(xt.grad.float().cpu().numpy(), g))
S = dict(padding='same', activation=None)
full = [((3, 5, 32, 128), S), ((3, 5, 32, 256), S), ((6, 1, 64, 128), dict(padding='valid', activation=None, conj=True))]
for rate in (0.0, 0.25):
    run(full, (2, 6, 40, 128), rate, 'full rate %.2f' % rate)
    run(full[:2], (2, 6, 40, 128), rate, 'first two rate %.2f' % rate)
    run(full[1:], (2, 6, 40, 128), rate, 'last two rate %.2f' % rate)
    run([((3, 5, 64, 256), S), ((3, 5, 64, 256), S)], (2, 6, 40, 256), rate, '64-64 x2 rate %.2f' % rate)
print('---- pieces')
rng = np.random.RandomState(1)
rnd = lambda a: torch.tensor(a).to(dtype).double().numpy()
for shape in ((2, 1, 40, 128), (2, 6, 40, 256)):
    for rate in (0.0, 0.25):
        pre = rnd(rng.randn(*shape)); dyn = rnd(rng.randn(*shape))
        keep = _np_drop_factor(shape, 333, rate)
        post = F.PostOp(None, -1, rate, 333)
        pt = torch.tensor(pre, device=dev).to(dtype)
        y = F.postop_fwd(pt, post)
        yn = rnd(np.maximum(pre, 0) * keep)
        dp = F.postop_bwd(y, torch.tensor(dyn, device=dev).to(dtype), post, None)
        want = dyn * keep * (yn > 0)
        print(shape, rate, 'fwd', _rel_err(y.float().cpu().numpy(), yn), 'bwd', _rel_err(dp.float().cpu().numpy(), want),
              'sum gpu', float(dp.float().sum()), 'sum want', want.sum())
print('---- manual last layer')
rng = np.random.RandomState(49)
S = dict(padding='same', activation=None)
for rate in (0.0, 0.25):
    xs, wsh = (2, 6, 40, 128), (3, 5, 32, 256)
    x = rnd(rng.randn(*xs)); w = rnd(rng.randn(*wsh) / np.sqrt(np.prod(wsh[:-1]) * 4)); b = (0.1 * rng.randn(wsh[-1])).astype(np.float32).astype(np.float64)
    pre = oracle.forward(x, w, b, 2, **S)
    keep = _np_drop_factor(pre.shape, 222, rate)
    yn = rnd(np.maximum(pre, 0) * keep)
    dyn = rnd(rng.randn(*yn.shape))
    xt = torch.tensor(x, device=dev).to(dtype); wt = torch.tensor(w, device=dev, dtype=torch.float32); bt = torch.tensor(b, device=dev, dtype=torch.float32)
    call = F.conv_call(xs, wsh, dtype, 2, 1, 'same', 'channels_last', 1, None, True, False)
    post = F.PostOp(None, -1, rate, 222)
    _, y = call.fwd_post(xt, wt, bt, post)
    print(rate, 'fwd', _rel_err(y.float().cpu().numpy(), yn), qcnn_amd._lib.last_path())
    dpre = F.postop_bwd(y, torch.tensor(dyn, device=dev).to(dtype), post, None)
    dpre_n = dyn * keep * (yn > 0)
    print('   dpre', _rel_err(dpre.float().cpu().numpy(), dpre_n))
    dxn, dwn, dbn = oracle.backward(x, w, b, dpre_n, 2, **S)
    dw, db = call.bwd_weight(xt, dpre, None, True)
    print('   wgrad on gpu dpre: dw', _rel_err(dw.cpu().numpy(), dwn), 'db', _rel_err(db.cpu().numpy(), dbn), qcnn_amd._lib.last_path())
    dw2, db2 = call.bwd_weight(xt, torch.tensor(dpre_n, device=dev).to(dtype), None, True)
    print('   wgrad on np dpre : dw', _rel_err(dw2.cpu().numpy(), dwn), 'db', _rel_err(db2.cpu().numpy(), dbn))
    print('   db sums: gpu', float(db.sum()), 'np', dbn.sum(), 'sum dpre gpu', float(dpre.double().sum()), 'np', dpre_n.sum(), 'rnd(np)', rnd(dpre_n).sum())
    dbn_r = rnd(dpre_n).reshape(-1, dpre_n.shape[-1]).sum(0)
    print('   db vs sum of ROUNDED np dpre', _rel_err(db.cpu().numpy(), dbn_r))
print('---- instrumented chain')
log = {}
orig_pb, orig_bp = F.postop_bwd, F._Call.bwd_post
def pb(pre, dy, post, dalpha):
    out = orig_pb(pre, dy, post, dalpha)
    log['pb'] = (pre.detach().clone(), dy.detach().clone(), out.detach().clone(), post.rate, post.seed)
    return out
def bp(self, x, dy, w, has_bias, post_x, x_pre, dalpha_x, direct=None):
    r = orig_bp(self, x, dy, w, has_bias, post_x, x_pre, dalpha_x, direct)
    log.setdefault('bp', []).append((x.detach().clone(), dy.detach().clone(), r[0].detach().clone(), None if r[1] is None else r[1].clone(), post_x.rate, post_x.seed))
    return r
F.postop_bwd, F._Call.bwd_post = pb, bp
rng = np.random.RandomState(49)
specs = full[:2]
rate = 0.25
x = rnd(rng.randn(2, 6, 40, 128).astype(np.float32).astype(np.float64))
ws = [rnd((rng.randn(*s) / np.sqrt(np.prod(s[:-1]) * 4)).astype(np.float32).astype(np.float64)) for s, _ in specs]
bs = [(0.1 * rng.randn(s[-1])).astype(np.float32).astype(np.float64) for s, _ in specs]
seeds = [111, 222]
acts, keeps = [x], []
for w, b, (_, kw), sd in zip(ws, bs, specs, seeds):
    pre = oracle.forward(acts[-1], w, b, 2, **kw)
    keep = _np_drop_factor(pre.shape, sd, rate); keeps.append(keep)
    acts.append(rnd(np.maximum(pre, 0) * keep))
dy = rnd(rng.randn(*acts[-1].shape).astype(np.float32).astype(np.float64))
xt = torch.tensor(x, device=dev).to(dtype).requires_grad_(True)
wt = [torch.nn.Parameter(torch.tensor(w, device=dev, dtype=torch.float32)) for w in ws]
bt = [torch.nn.Parameter(torch.tensor(b, device=dev, dtype=torch.float32)) for b in bs]
y = F.quaternion_conv_chain(xt, [(wt[i], bt[i], dict(specs[i][1], post=dict(alpha=None, rate=rate, seed=seeds[i]))) for i in range(2)])
y.backward(torch.tensor(dy, device=dev).to(dtype))
pre_t, dy_t, out_t, r_, s_ = log['pb']
print('postop_bwd called with rate', r_, 'seed', s_, ' y vs oracle', _rel_err(pre_t.float().cpu().numpy(), acts[2]), ' dy vs given', _rel_err(dy_t.float().cpu().numpy(), dy))
want = dy * keeps[1] * (acts[2] > 0)
print('   out vs numpy', _rel_err(out_t.float().cpu().numpy(), want))
xb, dyb, dxb, dwb, r_, s_ = log['bp'][0]
print('bwd_post: x vs acts[1]', _rel_err(xb.float().cpu().numpy(), acts[1]), 'dy vs want', _rel_err(dyb.float().cpu().numpy(), want), 'post_x rate', r_, 'seed', s_)
g, dw, db = oracle.backward(acts[1], ws[1], bs[1], want, 2, **specs[1][1])
print('   dw returned', _rel_err(dwb.cpu().numpy(), dw), ' param grad', _rel_err(wt[1].grad.cpu().numpy(), dw))
