#!/usr/bin/env python
"""Round 6: the staggered backward recurrence (B = 32) beside the REAL side-stream kernels of a C3
layer, one kind at a time: which of them costs it its ~1.2 us per time step?"""
import os, sys
import numpy as np, torch
here = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(here)))
from ctc_asr_amd import hip
hip.load()
DEV, GH, H, IN, ROWS, B, T = 'cuda', 4096, 1024, 2048, 8000, 32, 500
g = torch.Generator(device=DEV).manual_seed(0)
dxw2 = torch.randn(16000, 2 * GH, device=DEV, generator=g) * 1e-4
x = torch.rand(16000, IN, device=DEV, generator=g) * 2 - 1
yv = torch.rand(16000, 2 * H, device=DEV, generator=g) * 2 - 1
g_ih, g_hh = torch.zeros(GH, IN, device=DEV), torch.zeros(GH, H, device=DEV)
colmax = dxw2[8000:, :GH].abs().amax(dim=0).view(torch.int32)
stages = ROWS // 32
bufs = [torch.empty(hip.load().ctcasr_wgrad16_packed_bytes(stages, n), dtype=torch.uint8, device=DEV)
        for n in (GH, IN, H)]
scale, inv = hip.colscale_from_max(colmax)
def pack_d(): hip.wgrad16_pack(dxw2[8000:, :GH], ROWS, 0, stages, 1.0, col_scale=scale, out=bufs[0])
def pack_xy():
    hip.wgrad16_pack(x[8000:], ROWS, 0, stages, 32768.0, out=bufs[1])
    hip.wgrad16_pack(yv[:, :H], 16000, 8000 - B, stages, 32768.0, out=bufs[2])
pack_d(); pack_xy()
def gemm(parts):
    return lambda: hip.wgrad16_gemm(bufs[0], GH, stages, inv, bufs[1], 0, 32768.0, g_ih, y_packed=bufs[2],
                                    y_scale=32768.0, dw_y=g_hh, parts=parts)
xw = torch.randn(T, B, 2, GH, device=DEV, generator=g) * 0.5
w = torch.randn(2, GH, H, device=DEV, generator=g) / np.sqrt(H)
dy = torch.randn(T, B, 2 * H, device=DEV, generator=g)
wt = hip.transpose_batched(w)
flags = hip.RNN_F16 | hip.RNN_XCD_SPLIT | hip.RNN_STAGGER
yy, reserve, ws = hip.rnn_fwd('lstm', xw, w)
dd = hip.rnn_bwd('lstm', dy, yy, wt, reserve, workspace=ws, flags=flags)
side = torch.cuda.Stream()
ticket = [0]
def run(fn, count):
    times, side_ms = [], []
    for _ in range(4):
        torch.cuda.synchronize()
        ticket[0] += 1
        ready = torch.cuda.Event(); ready.record()
        r0, r1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        r0.record()
        hip.rnn_bwd('lstm', dy, yy, wt, reserve, dxw=dd, workspace=ws, flags=flags, ticket=ticket[0])
        r1.record()
        if fn is not None:
            with torch.cuda.stream(side):
                side.wait_event(ready)
                hip.rnn_resident_gate('lstm', ws, T, B, H, ticket[0], 300)
                s0, s1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                s0.record(side)
                for _ in range(count):
                    fn()
                s1.record(side)
        torch.cuda.synchronize()
        times.append(r0.elapsed_time(r1) * 1e3 / T)
        if fn is not None:
            side_ms.append(s0.elapsed_time(s1))
    return min(times), (min(side_ms) if side_ms else 0.0)
print('alone: {:.2f} us per time step'.format(run(None, 0)[0]))
for name, fn, count in (('packs of dxw (x 14)', pack_d, 14), ('packs of x and y (x 20)', pack_xy, 20),
                        ('weight-gradient kernel, 1 part (x 5)', gemm(1), 5),
                        ('weight-gradient kernel, 2 parts (x 5)', gemm(2), 5),
                        ('weight-gradient kernel, 4 parts (x 5)', gemm(4), 5)):
    us, ms = run(fn, count)
    print('{:40s}: {:.2f} us per time step; the side work took {:.2f} ms of the launch\'s {:.2f}'.format(
        name, us, ms, us * T / 1e3), flush=True)
hip.rnn_poll_error('lstm', ws, T, B, H)
