#!/usr/bin/env python
"""Times the epilogue kernels alone at the C3 shapes: bias_act_bwd (mask + bias gradient),
colsum_accumulate, bias_act_fwd, dropout, Adam.  GB/s = bytes the kernel must move / time."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ctc_asr_amd import hip  # noqa: E402


def timed(fn, reps=20):
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
    hip.load()
    for name, rows, cols in (('conv0 out (B=32)', 32 * 500 * 40, 32), ('conv1 out', 32 * 500 * 20, 32),
                             ('dense4 (B=32)', 16000, 2048), ('xw (B=32)', 16000, 8192),
                             ('conv0 out (B=16)', 16 * 500 * 40, 32)):
        y = torch.rand(rows, cols, device='cuda') * 2 - 0.5
        dy = torch.randn(rows, cols, device='cuda')
        dz = torch.empty_like(dy)
        db = torch.zeros(cols, device='cuda')
        mb = rows * cols * 4 / 1e6
        ms = timed(lambda: hip.bias_act_bwd(y, dy, 20.0, 0.0, db, dz=dz))
        print('{:<18s} [{} x {}] bias_act_bwd {:8.1f} us  {:7.0f} GB/s'.format(
            name, rows, cols, ms * 1e3, 3 * mb / ms))
        ms = timed(lambda: hip.colsum_accumulate(dy, db))
        print('{:<18s} [{} x {}] colsum       {:8.1f} us  {:7.0f} GB/s'.format(
            name, rows, cols, ms * 1e3, mb / ms))
        ms = timed(lambda: hip.bias_act_fwd(y, db, 20.0, 0.1, 1234))
        print('{:<18s} [{} x {}] bias_act_fwd {:8.1f} us  {:7.0f} GB/s'.format(
            name, rows, cols, ms * 1e3, 2 * mb / ms))


if __name__ == '__main__':
    main()
