#!/usr/bin/env python
"""hipBLASLt's heuristic over the row counts a bucketed C5 sequence produces (rows = T' x 16):
the activation-side GEMMs of a BiLSTM-1024 layer (rows are the M dimension) at the exact row
count and at the row count rounded up to a multiple of 128 / 256.  TFLOP/s per shape.

    python tools/gemm_rows_probe.py
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
    t_outs = sorted({(hip.features_num_frames(int(s.max())) + 1) // 2 for s in seq} | {35, 329, 850})
    kinds = {   # name: (K, N, transposed weight?)
        'xw = y W_ih^T      [R x 2048] x [2048 x 8192]': (2048, 8192, True),
        'dy_below = dxw W_ih [R x 8192] x [8192 x 2048]': (8192, 2048, False),
        'dense4 fwd         [R x 2048] x [2048 x 2048]': (2048, 2048, False),
    }
    big = 16 * 850 + 256
    for name, (k, n, trans) in kinds.items():
        x = torch.randn(big, k, device='cuda')
        w = torch.randn((n, k) if trans else (k, n), device='cuda')
        out = torch.empty(big, n, device='cuda')
        wm = w.t() if trans else w
        print(name)
        for t_out in t_outs:
            rows = 16 * t_out
            res = []
            for r in (rows, -(-rows // 128) * 128, -(-rows // 256) * 256):
                ms = timed(lambda r=r: torch.mm(x[:r], wm, out=out[:r]))
                res.append((r, ms, 2.0 * rows * k * n / ms / 1e9))
            print('  T\' {:4d}: '.format(t_out) + ' | '.join(
                'rows {:5d} {:.3f} ms {:5.1f} TF'.format(r, ms, tf) for r, ms, tf in res))


if __name__ == '__main__':
    main()
