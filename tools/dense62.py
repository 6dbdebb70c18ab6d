import torch, time
dev = torch.device('cuda:0')
def t(fn, n=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); e1.synchronize()
    return 1e3 * e0.elapsed_time(e1) / n
for dt in (torch.bfloat16, torch.float32):
    for N in (62, 64):
        x = torch.randn(51200, 256, device=dev, dtype=dt)
        w = torch.randn(N, 256, device=dev, dtype=dt)
        dy = torch.randn(51200, N, device=dev, dtype=dt)
        print(dt, N, 'fwd x@w.T %.1f us' % t(lambda: x @ w.t()), ' dx dy@w %.1f us' % t(lambda: dy @ w), ' dw dy.T@x %.1f us' % t(lambda: dy.t() @ x),
              ' dw (x.T@dy).T %.1f us' % t(lambda: (x.t() @ dy).t()))
print('split-K forms of dW (51200 rows)')
dt = torch.bfloat16
x = torch.randn(51200, 256, device=dev, dtype=dt); dy = torch.randn(51200, 62, device=dev, dtype=dt)
ref = (x.float().t() @ dy.float())
for S in (16, 32, 64, 128, 256):
    f = lambda: torch.bmm(x.view(S, -1, 256).transpose(1, 2), dy.view(S, -1, 62)).float().sum(0)
    err = float((f() - ref).abs().max() / ref.abs().max())
    print(dt, 'S', S, '%.1f us' % t(f), 'err %.2e' % err)
    f2 = lambda: torch.bmm(x.view(S, -1, 256).transpose(1, 2), dy.view(S, -1, 62), out_dtype=torch.float32).sum(0) if hasattr(torch, 'bmm') else None
    try:
        print('   fp32 out: %.1f us err %.2e' % (t(f2), float((f2() - ref).abs().max() / ref.abs().max())))
    except Exception as e:
        print('   fp32 out unsupported:', type(e).__name__)
