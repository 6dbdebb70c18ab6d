import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import qcnn_amd
F = qcnn_amd.functional
dev = torch.device('cuda:0')
g = torch.Generator(device=dev).manual_seed(3)
dtype = torch.bfloat16
cases = [((1, 9, 33, 128), (3, 3, 32, 256)), ((1, 9, 33, 256), (3, 3, 64, 128)), ((2, 14, 40, 128), (3, 5, 32, 128)),
         ((1, 9, 33, 128), (3, 3, 32, 128)), ((1, 9, 33, 256), (3, 3, 64, 256)), ((1, 9, 40, 256), (3, 5, 64, 128)), ((3, 70, 256), (5, 64, 128)), ((3, 70, 128), (3, 32, 128))]
for xs, ws in cases:
    rank = len(xs) - 2
    x = torch.randn(xs, device=dev, generator=g).to(dtype)
    w = torch.randn(ws, device=dev, generator=g) / 20
    b = torch.randn(ws[-1], device=dev, generator=g) / 10
    call = F.conv_call(tuple(xs), tuple(ws), dtype, rank, 1, 'same', 'channels_last', 1, 'linear', True)
    outs = []
    for nb in (False, True):
        from qcnn_amd import _lib
        prev = _lib.lib().qk_set_debug_flags(_lib.QK_DBG_NO_BAND16 if nb else 0)
        y = call.fwd(x, w, b)
        dy = torch.randn(y.shape, device=dev, generator=torch.Generator(device=dev).manual_seed(5)).to(dtype)
        dx = call.bwd_data(dy, None, w)
        torch.cuda.synchronize()
        _lib.lib().qk_set_debug_flags(prev)
        outs.append((y.float(), dx.float()))
    ey = float((outs[0][0] - outs[1][0]).abs().max() / outs[1][0].abs().max())
    ex = float((outs[0][1] - outs[1][1]).abs().max() / outs[1][1].abs().max())
    print(xs, ws, 'y err %.3g dx err %.3g' % (ey, ex))
    if ex > 1e-2:
        d = (outs[0][1] - outs[1][1]).abs()
        idx = (d > 0.05 * outs[1][1].abs().max()).nonzero()
        print('   bad dx count', idx.shape[0], 'first', idx[:6].tolist(), 'rows(o1)', sorted(set(idx[:, -2].tolist()))[:20], 'ch', sorted(set(idx[:, -1].tolist()))[:12])
