#!/usr/bin/env python
"""The data-gradient GEMM of the FIRST recurrent layer, dy_below = dxw [R x 8192] x W_ih
[8192 x 640] (640 = 20 frequencies x 32 channels of the conv stack), over the row counts of a
bucketed C5 sequence: hipBLASLt's heuristic picks kernels between 23 and 100 TFLOP/s for it.
Alternatives: rows rounded up to a multiple of 256, the transposed product, N padded to 768,
and the same for the forward projection xw = x [R x 640] x W_ih^T [640 x 8192].

    python tools/gemm_n640_probe.py
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from ctc_asr_amd import hip  # noqa: E402


def timed(fn, reps=6):
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
    hip.load()
    seq = bench.c5_bucket_sequence(16, 24)
    t_outs = sorted({(hip.features_num_frames(int(s.max())) + 1) // 2 for s in seq} |
                    {35, 329, 500, 850})
    batches = (16, 32)
    big = 32 * 850 + 256
    dxw = torch.randn(big, 8192, device='cuda')
    w = torch.randn(8192, 640, device='cuda')
    w768 = torch.zeros(8192, 768, device='cuda')
    w768[:, :640] = w
    out = torch.empty(big, 640, device='cuda')
    out768 = torch.empty(big, 768, device='cuda')
    out_t = torch.empty(640, big, device='cuda')
    x = torch.randn(big, 640, device='cuda')
    xw = torch.empty(big, 8192, device='cuda')
    for batch in batches:
        print('batch', batch)
        for t_out in (t_outs if batch == 16 else (35, 181, 329, 500, 617, 801, 850)):
            rows = batch * t_out
            r256 = -(-rows // 256) * 256
            flops = 2.0 * rows * 8192 * 640
            tf = lambda ms: flops / ms / 1e9
            a = timed(lambda: torch.mm(dxw[:rows], w, out=out[:rows]))
            b = timed(lambda: torch.mm(dxw[:r256], w, out=out[:r256]))
            c = timed(lambda: torch.mm(w.t(), dxw[:rows].t(), out=out_t[:, :rows]) if False else
                      torch.mm(w.t(), dxw[:rows].t()))
            d = timed(lambda: torch.mm(dxw[:rows], w768, out=out768[:rows]))
            f = timed(lambda: torch.mm(x[:rows], w.t(), out=xw[:rows]))
            g = timed(lambda: torch.mm(x[:r256], w.t(), out=xw[:r256]))
            print('  T\' {:4d} rows {:5d}: bwd plain {:.3f} ms {:5.1f} TF | rows->256 {:5.1f} TF | '
                  'transposed {:5.1f} TF | N->768 {:5.1f} TF || fwd plain {:.3f} ms {:5.1f} TF | '
                  'rows->256 {:5.1f} TF'.format(t_out, rows, a, tf(a), tf(b), tf(c), tf(d), f,
                                                 tf(f), tf(g)))


if __name__ == '__main__':
    main()
