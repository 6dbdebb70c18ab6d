"""Repeatability / exactness probe for k_wgrad16's side outputs (diagnostic)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import qcnn_amd
F = qcnn_amd.functional
dev = torch.device('cuda:0')
g = torch.Generator(device=dev).manual_seed(3)
for dtype in (torch.bfloat16, torch.float16):
    for xs, ws in (((2, 14, 40, 128), (3, 5, 32, 128)), ((1, 14, 40, 256), (3, 5, 64, 256)), ((1, 9, 33, 128), (3, 3, 32, 256)), ((3, 70, 256), (5, 64, 128))):
        rank = len(xs) - 2
        x = torch.randn(xs, device=dev, generator=g).to(dtype)
        w = torch.randn(ws, device=dev, generator=g) / 20
        b = torch.randn(ws[-1], device=dev, generator=g) / 10
        call = F.conv_call(tuple(xs), tuple(ws), dtype, rank, 1, 'same', 'channels_last', 1, 'relu', True)
        y = call.fwd(x, w, b)
        dy = torch.randn(y.shape, device=dev, generator=g).to(dtype)
        want = torch.where(y > 0, dy, torch.zeros_like(dy))
        from qcnn_amd import _lib
        with _lib.debug_flags(_lib.QK_DBG_NO_MFMA16):
            dw_ref, db_ref = call.bwd_weight(x, dy, y, True)
        bad_dym = bad_dw = 0
        worst = 0.0
        for rep in range(20):
            dym = torch.full(((dy.numel() + 127) // 128 * 128,), 7.0, dtype=dtype, device=dev)
            dw, db = call.bwd_weight(x, dy, y, True, masked_dy_out=dym)
            torch.cuda.synchronize()
            got = dym[:dy.numel()].view_as(dy)
            if not torch.equal(got, want):
                bad_dym += 1
                diff = (got != want)
                if bad_dym == 1:
                    idx = diff.nonzero()
                    i0 = tuple(idx[0].tolist())
                    print('   got', got[i0[:-1]][i0[-1]:i0[-1]+8].float().tolist(), 'want', want[i0[:-1]][i0[-1]:i0[-1]+8].float().tolist(), 'dy', dy[i0[:-1]][i0[-1]:i0[-1]+8].float().tolist())
                    print('   first bad dym idx', idx[0].tolist(), 'n bad', int(diff.sum()), 'rows', sorted(set(idx[:, :-1].flatten().tolist()))[:10] if idx.shape[1] > 1 else '')
            e = float((dw - dw_ref).abs().max() / dw_ref.abs().max())
            worst = max(worst, e)
            if e > 2e-3:
                bad_dw += 1
        print(dtype, xs, ws, 'bad dym runs', bad_dym, 'bad dw runs', bad_dw, 'worst dw err %.2e' % worst)
