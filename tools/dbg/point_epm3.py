import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, numpy as np
import qcnn_amd
from qcnn_amd import _lib
F = qcnn_amd.functional
dev = torch.device('cuda:0')
torch.manual_seed(0)
def case(dtype, cq, fq, B, T, Fr=14):
    call = F.conv_call((B, Fr, T, 4 * cq), (Fr, 1, cq, 4 * fq), dtype, 2, 1, 'valid', 'channels_last', 1, None, True, True)
    w = (torch.randn(Fr, 1, cq, 4 * fq, device=dev) / 40)
    dy = torch.randn(B, 1, T, 4 * fq, device=dev).to(dtype)
    with _lib.debug_flags(_lib.QK_DBG_NO_POINT16):
        ref = call.bwd_data(dy, None, w).float()
    nbad = nmis = 0
    for rep in range(10):
        a = call.bwd_data(dy, None, w).float()
        assert _lib.last_path() == 'mfma16_point'
        bad = ~torch.isfinite(a)
        nbad += int(bad.any())
        d = (a - ref)[~bad].abs().max()
        nmis += int(d > 1e-6)
    print(dtype, 'cq %d fq %d B %d T %d: nan runs %d mismatch runs %d maxdiff %.3g' % (cq, fq, B, T, nbad, nmis, float(d)))
    if nbad or nmis:
        bad = (~torch.isfinite(a)) | ((a - ref).abs() > 1e-6)
        idx = bad.nonzero()
        print('  n bad', len(idx), 'first', idx[:3].tolist(), 'last', idx[-3:].tolist(), 'channels', sorted(set((idx[:, -1] // 64).tolist())))
for dtype in (torch.float16, torch.bfloat16):
    case(dtype, 256, 64, 32, 200)
    case(dtype, 128, 64, 8, 100)
    case(dtype, 64, 64, 256, 200)
