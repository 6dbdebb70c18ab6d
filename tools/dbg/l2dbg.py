import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from qcnn_amd import dp, functional as F
from qcnn_amd.models import TimitQCNN
dev = torch.device('cuda', 0)
def build():
    np.random.seed(1); torch.manual_seed(1)
    m = TimitQCNN(num_layers=2, start_filter=32, l2=1e-2)
    x = torch.randn(3, 4, 41, 24, device=dev).to(torch.bfloat16)
    with torch.no_grad():
        m(x[:1])
    m.to(dev)
    return m, x
torch.manual_seed(5)
labels = torch.randint(0, 61, (3, 5), device=dev); il = torch.full((3, 1), 24); ll = torch.full((3, 1), 5)
outs = []
for rep in range(2):
    m, x = build()
    print('rep', rep, 'x', float(x.float().sum()), 'w', float(sum(p.detach().double().sum() for p in m.parameters())))
    with torch.no_grad():
        pr = m(x)
        print('  pred sum', float(pr.float().sum()), 'pred^2', float((pr.float()**2).sum()), 'ctc', m.ctc_loss(x, labels, il, ll).flatten().tolist(), 'reg', float(m.regularization_loss()))
        pr2 = m(x)
        print('  same forward twice equal:', bool(torch.equal(pr, pr2)))
    loss = m.training_loss(x, labels, il, ll)
    print('  loss', float(loss))
    loss.backward()
    outs.append([p.grad.detach().float().cpu().numpy().copy() for p in m.parameters()])
names = [n for n, _ in m.named_parameters()]
print('x checksum', float(x.float().sum()), 'w checksum', float(sum(p.detach().double().sum() for p in m.parameters())))
for n, a, b in zip(names, outs[0], outs[1]):
    print('%-28s rep-to-rep err %.3g  max %.3g' % (n, np.abs(a - b).max() / max(np.abs(a).max(), 1e-30), np.abs(a).max()))
m, x = build()
flat = dp.FlatParams([p for p in m.parameters() if p.requires_grad])
m.training_loss(x, labels, il, ll).backward()
for n, p, w in zip(names, flat.params, outs[0]):
    g = p.grad.detach().float().cpu().numpy()
    print('%-28s flat(auto) err %.3g direct=%s' % (n, np.abs(g - w).max() / max(np.abs(w).max(), 1e-30), p._qk_direct_grad))
