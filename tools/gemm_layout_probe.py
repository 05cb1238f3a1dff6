import torch, sys
def timed(fn, reps=40):
    for _ in range(40): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps
for rows in (16000, 8000):
    x = torch.randn(rows, 2048, device='cuda'); w = torch.randn(8192, 2048, device='cuda'); wt = w.t().contiguous()
    out = torch.empty(rows, 8192, device='cuda')
    fl = 2.0 * rows * 2048 * 8192
    a = timed(lambda: torch.mm(x, w.t(), out=out)); b = timed(lambda: torch.mm(x, wt, out=out))
    print('rows', rows, 'xw NT (now) {:.3f} ms {:.1f} TF | NN (pre-transposed W) {:.3f} ms {:.1f} TF'.format(a, fl/a/1e9, b, fl/b/1e9))
    dxw = torch.randn(rows, 8192, device='cuda'); out2 = torch.empty(rows, 2048, device='cuda')
    a = timed(lambda: torch.mm(dxw, w, out=out2)); b = timed(lambda: torch.mm(dxw, wt.t(), out=out2))
    print('rows', rows, 'dy_below NN (now) {:.3f} ms {:.1f} TF | NT {:.3f} ms {:.1f} TF'.format(a, fl/a/1e9, b, fl/b/1e9))
    # first layer
    x0 = torch.randn(rows, 640, device='cuda'); w0 = torch.randn(8192, 640, device='cuda'); w0t = w0.t().contiguous()
    fl0 = 2.0 * rows * 640 * 8192
    a = timed(lambda: torch.mm(x0, w0.t(), out=out)); b = timed(lambda: torch.mm(x0, w0t, out=out))
    print('rows', rows, 'xw0 NT (now) {:.3f} ms {:.1f} TF | NN {:.3f} ms {:.1f} TF'.format(a, fl0/a/1e9, b, fl0/b/1e9))
    # dense4 fwd [rows x 2048] x [2048 x 2048] (kernel stored [in, out]: NN now)
    k = torch.randn(2048, 2048, device='cuda'); o3 = torch.empty(rows, 2048, device='cuda'); fl3 = 2.0*rows*2048*2048
    a = timed(lambda: torch.mm(x, k, out=o3)); b = timed(lambda: torch.mm(x, k.t(), out=o3))
    print('rows', rows, 'dense4 NN (now) {:.3f} ms {:.1f} TF | NT {:.3f} ms {:.1f} TF'.format(a, fl3/a/1e9, b, fl3/b/1e9))
