#!/usr/bin/env python
"""Forward recurrence on the fp16 matrix pipe (CTCASR_RNN_F16) next to the fp32-MFMA kernel:
time per step and the error of y / the cell state against a float64 recurrence on the same
inputs (batched torch ops on the GPU).

    python tools/rnn_fwd_f16_probe.py [T B H [cell]]        default 500 32 1024 lstm
"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ctc_asr_amd import hip  # noqa: E402


def reference(cell, xw, w_hh, b_hh):
    """float64 recurrence, every row over all T steps: y [T, B, 2H]."""
    T, B, _, GH = xw.shape
    H = w_hh.shape[2]
    x64, w64 = xw.double(), w_hh.double()
    y = torch.zeros(T, B, 2 * H, dtype=torch.float64, device=xw.device)
    for d in (0, 1):
        h = torch.zeros(B, H, dtype=torch.float64, device=xw.device)
        c = torch.zeros_like(h)
        for s in range(T):
            t = s if d == 0 else T - 1 - s
            rec = h @ w64[d].t()
            x = x64[t, :, d]
            if cell == 'lstm':
                i, f, g, o = (x + rec).split(H, dim=1)
                c = torch.sigmoid(f) * c + torch.sigmoid(i) * torch.tanh(g)
                h = torch.sigmoid(o) * torch.tanh(c)
            else:
                xr, xz, xn = x.split(H, dim=1)
                rr, rz, rn = rec.split(H, dim=1)
                r, z = torch.sigmoid(xr + rr), torch.sigmoid(xz + rz)
                n = torch.tanh(xn + r * (rn + b_hh[d, 2 * H:].double()))
                h = (1 - z) * n + z * h
            y[t, :, d * H:(d + 1) * H] = h
    return y


def main():
    T, B, H = (int(v) for v in sys.argv[1:4]) if len(sys.argv) >= 4 else (500, 32, 1024)
    cell = sys.argv[4] if len(sys.argv) >= 5 else 'lstm'
    w_mult = float(os.environ.get('W_MULT', '1'))        # livelier recurrent weights
    G = hip.CELL_GATES[cell]
    hip.load(os.environ.get('CTCASR_LIB'))
    g = torch.Generator(device='cuda').manual_seed(0)
    xw = torch.randn(T, B, 2, G * H, device='cuda', generator=g) * 0.5
    w = torch.randn(2, G * H, H, device='cuda', generator=g) / np.sqrt(H) * w_mult
    b_hh = torch.randn(2, G * H, device='cuda', generator=g) * 0.3 if cell == 'gru' else None
    ref = reference(cell, xw, w, b_hh)
    print('T {} B {} H {} {}: |y| rms {:.3f}'.format(T, B, H, cell, float(ref.pow(2).mean().sqrt())))
    for name, flags in (('fp32 MFMA', hip.RNN_DEFAULT), ('fp16 x 3', hip.RNN_F16)):
        y, reserve, ws = hip.rnn_fwd(cell, xw, w, flags=flags, b_hh_n=b_hh)
        hip.rnn_poll_error(cell, ws, T, B, H)
        start, stop = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 5
        start.record()
        for _ in range(reps):
            hip.rnn_fwd(cell, xw, w, y=y, reserve=reserve, workspace=ws, flags=flags, b_hh_n=b_hh)
        stop.record()
        torch.cuda.synchronize()
        hip.rnn_poll_error(cell, ws, T, B, H)
        ms = start.elapsed_time(stop) / reps
        err = (y.double() - ref).abs()
        # error by time step of the recurrence (the last steps carry the accumulated drift)
        tail = torch.cat([err[-10:, :, :H], err[:10, :, H:]]).max()
        print('{:<10s} {:.3f} ms per call, {:.2f} us per time step; y vs float64: max {:.2e} rms '
              '{:.2e}, max over the last 10 steps of both directions {:.2e}'.format(
                  name, ms, ms * 1e3 / T, float(err.max()), float(err.pow(2).mean().sqrt()),
                  float(tail)))


if __name__ == '__main__':
    main()
