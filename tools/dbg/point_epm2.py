import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, numpy as np
import qcnn_amd
from qcnn_amd import _lib
F = qcnn_amd.functional
dev = torch.device('cuda:0')
torch.manual_seed(0)
dtype = torch.float16 if os.environ.get('FP16') else torch.bfloat16
def case(cq, fq, B, T, ab):
    x = torch.randn(B, 6, T, 4 * cq, device=dev).to(dtype).requires_grad_(True)
    w0 = (torch.randn(3, 5, cq, 4 * cq, device=dev) / 60).requires_grad_(True)
    w1 = (torch.randn(6, 1, cq, 4 * fq, device=dev) / 40).requires_grad_(True)
    def run():
        x.grad = None
        y = F.quaternion_conv_chain(x, [(w0, None, dict(padding='same', activation='relu')), (w1, None, dict(padding='valid', activation='relu', conj=True))])
        y.backward(torch.ones_like(y))
        torch.cuda.synchronize()
        return x.grad.float().clone()
    with _lib.debug_flags(_lib.QK_DBG_NO_POINT16):
        ref = run()
    nbad = 0; nmis = 0
    with _lib.debug_flags(ab << 8):
        for rep in range(20):
            a = run()
            bad = ~torch.isfinite(a)
            nbad += int(bad.any())
            nmis += int(((a - ref)[~bad].abs().max() > 1e-6))
    if nbad or nmis:
        bad = (~torch.isfinite(a)) | ((a - ref).abs() > 1e-6)
        idx = bad.nonzero()
        print('  n bad', len(idx), 'first', idx[:3].tolist(), 'last', idx[-3:].tolist(), 'chan blocks', sorted(set((idx[:, -1] // 64).tolist())))
    print('cq %d fq %d B %d T %d ablate %d: runs with nan %d, with mismatch %d of 20' % (cq, fq, B, T, ab, nbad, nmis))
for ab in (0,):
    for (cq, fq, B, T) in ((256, 64, 2, 40), (128, 64, 8, 100), (256, 64, 32, 200), (256, 32, 4, 50)):
        case(cq, fq, B, T, ab)
