"""Prices VERDICT r04 item 4 (a half-chip forward recurrence at B = 32 with the next layer's input
projection beside it): how long does the C3 projection - the fp16 x 3 library GEMM [16000 x 6144] x
[6144 x 8192] and its per-direction halves over a quarter of the steps - take on the 128 CUs a
half-chip persistent launch leaves?  A half-chip BACKWARD recurrence launch stands in for the
resident kernel (no half-chip forward kernel exists at B = 32).  python tools/proj_half_chip_probe.py"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ctc_asr_amd import hip        # noqa: E402

F32 = torch.float32
T, B, H = 500, 32, 1024
g = torch.Generator(device='cuda').manual_seed(0)
xw = torch.randn(T, B, 2, 4 * H, device='cuda', generator=g) * 0.5
w_hh = torch.randn(2, 4 * H, H, device='cuda', generator=g) / 32
dy = torch.randn(T, B, 2 * H, device='cuda', generator=g)
wt = hip.transpose_batched(w_hh)
y, reserve, ws = hip.rnn_fwd('lstm', xw, w_hh, flags=hip.RNN_F16)
dxw = hip.rnn_bwd('lstm', dy, y, wt, reserve, workspace=ws, flags=hip.RNN_F16)
rows = T * B
x16 = torch.randn(rows, 3, 2 * H, device='cuda').half()
w16 = torch.randn(8 * H, 3, 2 * H, device='cuda').half()
out = torch.empty(rows, 8 * H, device='cuda')
quarter = rows // 4


def whole():
    torch.mm(x16.view(rows, -1), w16.view(8 * H, -1).t(), out_dtype=F32, out=out)


def halves_of_a_quarter():
    # one direction's half of K (three piece blocks of H columns) for a quarter of the steps
    for d in (0, 1):
        dst = out[d * quarter:(d + 1) * quarter]
        for blk in range(3):
            a = x16[d * quarter:(d + 1) * quarter, blk, d * H:(d + 1) * H]
            b = w16[:, blk, d * H:(d + 1) * H]
            if blk == 0:
                torch.mm(a, b.t(), out_dtype=F32, out=dst)
            else:
                torch.addmm(dst, a, b.t(), out_dtype=F32, out=dst)


def timed(fn, beside):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    ready = torch.cuda.Event()
    ready.record()
    if beside:
        hip.rnn_bwd('lstm', dy, y, wt, reserve, dxw=dxw, workspace=ws, flags=hip.RNN_F16, ticket=7)
    with torch.cuda.stream(side):
        side.wait_event(ready)
        if beside:
            hip.rnn_resident_gate('lstm', ws, T, B, H, 7, 300)
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(side)
        reps = 2 if beside else 5
        for _ in range(reps):
            fn()
        e.record(side)
    torch.cuda.synchronize()
    return s.elapsed_time(e) / reps


for name, fn in (('whole projection (one call)', whole),
                 ('a quarter of the steps, per-direction half-K calls (6 calls)', halves_of_a_quarter)):
    print('{}: alone {:.3f} ms, beside a half-chip recurrence launch {:.3f} ms'.format(
        name, timed(fn, False), timed(fn, True)))
hip.rnn_poll_error('lstm', ws, T, B, H)
