"""Weight-gradient GEMMs of one finished step range (C3: 250 steps x 32 rows) - today four library
calls (W_ih and W_hh per direction) - against two batched calls over the directions (W_hh's 64
output tiles per call become 128).  fp16 pieces [rows, 3, cols] read as [3 rows, cols] (TN).
python tools/wgrad_bmm_probe.py [rows] [beside]   (beside: a half-chip backward recurrence runs)"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ctc_asr_amd import hip        # noqa: E402

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 8000
beside = len(sys.argv) > 2
DEV, GH, H, IN = 'cuda', 4096, 1024, 2048
F32 = torch.float32
d16 = torch.randn(2, rows, 3, GH, device=DEV).half()
x16 = torch.randn(16000, 3, IN, device=DEV).half()
y16 = torch.randn(16000, 3, 2 * H, device=DEV).half()
a0, a1 = 8000, 0            # first rows of the two directions' ranges (dir 1 mirrored)

if beside:
    T, B = 500, 32
    g = torch.Generator(device=DEV).manual_seed(0)
    xw = torch.randn(T, B, 2, GH, device=DEV, generator=g) * 0.5
    w = torch.randn(2, GH, H, device=DEV, generator=g) / np.sqrt(H)
    dy = torch.randn(T, B, 2 * H, device=DEV, generator=g)
    wt = hip.transpose_batched(w)
    y, reserve, ws = hip.rnn_fwd('lstm', xw, w, flags=hip.RNN_F16)
    dxw = torch.empty(T, B, 2, GH, device=DEV)
    rec = torch.cuda.Stream()


def separate(which):
    outs = []
    for d, a in enumerate((a0, a1)):
        lhs = d16[d].reshape(rows * 3, GH).t()
        if which == 'ih':
            rhs = x16.view(-1, IN)[3 * a:3 * (a + rows)]
        else:
            rhs = y16.view(-1, 2 * H)[3 * a:3 * (a + rows), d * H:(d + 1) * H]
        outs.append(torch.mm(lhs, rhs, out_dtype=F32))
    return outs


def batched(which):
    lhs = d16.view(2, rows * 3, GH).transpose(1, 2)
    if which == 'ih':
        flat = x16.view(-1, IN)
        rhs = torch.as_strided(flat, (2, 3 * rows, IN), (3 * (a0 - a1) * IN, IN, 1),
                               3 * a1 * IN)          # batch 0 = dir 1 (lower offset), 1 = dir 0
        return torch.bmm(lhs.flip(0), rhs, out_dtype=F32)
    flat = y16.view(-1, 2 * H)
    off0, off1 = 3 * a0 * 2 * H, 3 * a1 * 2 * H + H
    rhs = torch.as_strided(flat, (2, 3 * rows, H), (off0 - off1, 2 * H, 1), off1)
    return torch.bmm(lhs.flip(0), rhs, out_dtype=F32)


def timed(fn, reps=10):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    if beside:
        with torch.cuda.stream(rec):
            hip.rnn_bwd('lstm', dy, y, wt, reserve, dxw=dxw, workspace=ws, flags=hip.RNN_F16)
        torch.cuda._sleep(200000)
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        out = fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / reps, out


for which in ('ih', 'hh'):
    ms_s, o_s = timed(lambda: separate(which))
    ms_b, o_b = timed(lambda: batched(which))
    err = max(float((o_b[1] - o_s[0]).abs().max()), float((o_b[0] - o_s[1]).abs().max()))
    print('dW_{} both directions: two calls {:.3f} ms, one batched call {:.3f} ms  (max diff {:.2e}){}'
          .format(which, ms_s, ms_b, err, '  beside a backward recurrence' if beside else ''))
