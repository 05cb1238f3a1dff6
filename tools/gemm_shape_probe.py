#!/usr/bin/env python
"""Which kernels hipBLASLt's heuristic picks for the weight-gradient GEMMs (A^T B with a deep K =
T' x B rows) at the padded lengths a bucketed C5 sequence produces, and what splitting K by hand
(addmm_ over row blocks) or computing the transpose instead would cost.

    python tools/gemm_shape_probe.py [batch]
"""
import sys

import torch


def timed(fn, reps=10):
    fn(); fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


def main():
    batch = int(sys.argv[1]) if len(sys.argv) > 1 else 16
    shapes = {'dense4 dW = rnn_flat^T dz [2048 x R] x [R x 2048]': (2048, 2048),
              'dW_ih  = dxw_d^T x   [4096 x R] x [R x 2048]': (4096, 2048),
              'dW_hh  = dxw_d^T h   [4096 x R] x [R x 1024]': (4096, 1024)}
    for t_out in (181, 329, 500, 617, 801, 850):
        rows = t_out * batch
        print('T\' = {} (rows {})'.format(t_out, rows))
        for name, (m, n) in shapes.items():
            a = torch.randn(rows, m, device='cuda')
            b = torch.randn(rows, n, device='cuda')
            out = torch.empty(m, n, device='cuda')
            out_t = torch.empty(n, m, device='cuda')
            flops = 2.0 * rows * m * n
            plain = timed(lambda: torch.mm(a.t(), b, out=out))
            trans = timed(lambda: torch.mm(b.t(), a, out=out_t))

            def split(parts):
                step = -(-rows // parts)
                torch.mm(a[:step].t(), b[:step], out=out)
                for lo in range(step, rows, step):
                    out.addmm_(a[lo:lo + step].t(), b[lo:lo + step])
            s2, s4 = timed(lambda: split(2)), timed(lambda: split(4))
            print('  {:<52s} plain {:.3f} ms ({:5.1f} TF) | transposed {:.3f} | split-K x2 {:.3f} '
                  'x4 {:.3f}'.format(name, plain, flops / plain / 1e9, trans, s2, s4))


if __name__ == '__main__':
    main()
