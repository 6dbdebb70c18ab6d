import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, numpy as np
import qcnn_amd
from qcnn_amd import _lib
F = qcnn_amd.functional
dev = torch.device('cuda:0')
torch.manual_seed(0)
dtype = torch.bfloat16
cq, fq, B, T = int(os.environ.get('CQ', 256)), 64, 32, 200
call = F.conv_call((B, 6, T, 4 * cq), (6, 1, cq, 4 * fq), dtype, 2, 1, 'valid', 'channels_last', 1, None, False, True)
w = (torch.randn(6, 1, cq, 4 * fq, device=dev) / 40)
dy = torch.randn(B, 1, T, 4 * fq, device=dev).to(dtype)
x = torch.randn(B, 6, T, 4 * cq, device=dev).to(dtype)
flags = _lib.QK_BWD_MASK_DX | _lib.QK_BWD_DY_PREMASKED
with _lib.debug_flags(_lib.QK_DBG_NO_POINT16):
    ref = call.bwd(x, dy, None, w, False, flags=flags)[0].float()
nomask = call.bwd_data(dy, None, w).float()
print('path', _lib.last_path(), 'host-masked point result vs reference: max diff', float((nomask * (x.float() > 0) - ref).abs().max()))
for rep in range(2):
    a = call.bwd(x, dy, None, w, False, flags=flags)[0].float()
    bad = (~torch.isfinite(a)) | ((a - ref).abs() > 1e-6)
    rows = bad.any(dim=-1)                      # (B, 6, T)
    print('rep', rep, 'bad elems', int(bad.sum()), 'bad rows', int(rows.sum()), 'of', rows.numel())
    if rows.any():
        r = rows.reshape(B, 6, T)
        print('  bad rows per tap', r.sum(dim=(0, 2)).tolist())
        print('  bad rows per sample (first 8)', r.sum(dim=(1, 2)).tolist()[:8])
        flat = rows.permute(1, 0, 2).reshape(6, B * T)      # tap-major gathered-row index
        for tap in range(6):
            idx = flat[tap].nonzero().flatten()
            if len(idx):
                print('  tap', tap, 'bad gathered rows: n', len(idx), 'first', idx[:12].tolist(), 'mod128 set', sorted(set((idx % 128).tolist()))[:40])
        i = rows.nonzero()[0].tolist()
        ch = bad[i[0], i[1], i[2]].nonzero().flatten()
        print('  one bad row', i, 'bad channels n', len(ch), ch[:16].tolist(), 'got', a[i[0], i[1], i[2], ch[:4]].tolist(), 'want', ref[i[0], i[1], i[2], ch[:4]].tolist())
