#!/usr/bin/env python
"""Does a tuned hipBLASLt / rocBLAS solution beat the heuristic one for the C3 input projection
(16000 x 2048 x 8192, fp32) when the GEMM runs back to back?  torch's TunableOp with 400 ms per
candidate reports 3.50 ms for its pick; the same pick, and the default, sustain 4.0 ms (134
TFLOP/s): the rate is set by the power limit, not by the kernel.  python tools/gemm_tune_probe.py"""
import os, sys, time
import torch
import torch.cuda.tunable as tunable

def timed(fn, reps=30):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / reps

rows, k, n = 16000, 2048, 8192
x = torch.randn(rows, k, device='cuda'); w = torch.randn(n, k, device='cuda') / k ** 0.5
out = torch.empty(rows, n, device='cuda')
f = lambda: torch.mm(x, w.t(), out=out)
base = timed(f)
print('default sustained {:.3f} ms {:.1f} TF'.format(base, 2.0 * rows * k * n / base / 1e9))
tunable.enable(True); tunable.tuning_enable(True)
tunable.set_max_tuning_duration(400); tunable.set_max_tuning_iterations(2000)
tunable.set_filename('/tmp/tune_long.csv')
t0 = time.time(); f(); torch.cuda.synchronize()
print('tuning took {:.1f} s'.format(time.time() - t0))
tunable.tuning_enable(False)
tuned = timed(f)
print('tuned sustained {:.3f} ms {:.1f} TF'.format(tuned, 2.0 * rows * k * n / tuned / 1e9))
print(tunable.get_results())
