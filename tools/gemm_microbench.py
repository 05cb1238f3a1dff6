#!/usr/bin/env python
"""Times the library GEMM forms of the input projection xw = x W_ih^T + b (rows x K times K x N)
for the two operand layouts hipBLASLt can be given: W stored [N, K] (transposed view, what the
parameter arena holds) or a [K, N] copy.  python tools/gemm_microbench.py [rows K N]"""
import sys

import torch


def timed(fn, reps=10):
    fn()
    torch.cuda.synchronize()
    start, stop = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    start.record()
    for _ in range(reps):
        fn()
    stop.record()
    torch.cuda.synchronize()
    return start.elapsed_time(stop) / reps


def main():
    rows, k, n = (int(v) for v in sys.argv[1:4]) if len(sys.argv) >= 4 else (16000, 2048, 8192)
    x = torch.randn(rows, k, device='cuda')
    w = torch.randn(n, k, device='cuda') / k ** 0.5
    wt = w.t().contiguous()
    bias = torch.randn(n, device='cuda')
    out = torch.empty(rows, n, device='cuda')
    flops = 2.0 * rows * k * n
    for name, fn in (('addmm(b, x, W[N,K].t())', lambda: torch.addmm(bias, x, w.t(), out=out)),
                     ('addmm(b, x, Wt[K,N])', lambda: torch.addmm(bias, x, wt, out=out)),
                     ('mm(x, W.t())', lambda: torch.mm(x, w.t(), out=out)),
                     ('mm(x, Wt)', lambda: torch.mm(x, wt, out=out)),
                     ('transpose copy W -> Wt', lambda: wt.copy_(w.t()))):
        ms = timed(fn)
        print('{:<28s} {:8.3f} ms  {:7.1f} TFLOP/s'.format(name, ms, flops / ms / 1e9))


if __name__ == '__main__':
    main()
