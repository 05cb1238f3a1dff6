"""Times the own weight-gradient kernel (csrc/wgrad16.hip: packs + one launch for W_ih and W_hh of
a direction and step range) against the library form it replaces (column split + two TN GEMMs +
rescales), alone and beside a half-chip backward recurrence.  python tools/wgrad16_probe.py [beside]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ctc_asr_amd import hip, split_gemm        # noqa: E402

beside = len(sys.argv) > 1
DEV, GH, H, IN, ROWS, B = 'cuda', 4096, 1024, 2048, 8000, 32
g = torch.Generator(device=DEV).manual_seed(0)
dxw = torch.randn(16000, 2 * GH, device=DEV, generator=g) * 1e-4
x = torch.rand(16000, IN, device=DEV, generator=g) * 2 - 1
y = torch.rand(16000, 2 * H, device=DEV, generator=g) * 2 - 1
x16 = split_gemm.split16(x, 32768.0, split_gemm.H_A)
y16 = split_gemm.split16(y, 32768.0, split_gemm.H_A)
g_ih, g_hh = torch.zeros(GH, IN, device=DEV), torch.zeros(GH, H, device=DEV)
colmax = dxw[8000:, :GH].abs().amax(dim=0).view(torch.int32)
stages = ROWS // 32
bufs = [torch.empty(hip.load().ctcasr_wgrad16_packed_bytes(stages, n), dtype=torch.uint8, device=DEV)
        for n in (GH, IN, H)]

if beside:
    T = 500
    xw = torch.randn(T, B, 2, GH, device=DEV, generator=g) * 0.5
    w = torch.randn(2, GH, H, device=DEV, generator=g) / np.sqrt(H)
    dy = torch.randn(T, B, 2 * H, device=DEV, generator=g)
    wt = hip.transpose_batched(w)
    yy, reserve, ws = hip.rnn_fwd('lstm', xw, w, flags=hip.RNN_F16)
    dd = torch.empty(T, B, 2, GH, device=DEV)
    rec = torch.cuda.Stream()


def library():
    d16, inv = split_gemm.wgrad16_operand(dxw[8000:, :GH], colmax)
    split_gemm.wgrad16(g_ih, d16, inv, x16, 32768.0, 8000)
    split_gemm.wgrad16(g_hh, d16, inv, y16, 32768.0, 8000 - B, x_cols=slice(0, H))


PARTS = int(os.environ.get('PARTS', '4'))


def own():
    scale, inv = hip.colscale_from_max(colmax)
    d_pk = hip.wgrad16_pack(dxw[8000:, :GH], ROWS, 0, stages, 1.0, col_scale=scale, out=bufs[0])
    x_pk = hip.wgrad16_pack(x[8000:], ROWS, 0, stages, 32768.0, out=bufs[1])
    y_pk = hip.wgrad16_pack(y[:, :H], 16000, 8000 - B, stages, 32768.0, out=bufs[2])
    hip.wgrad16_gemm(d_pk, GH, stages, inv, x_pk, 0, 32768.0, g_ih, y_packed=y_pk, y_scale=32768.0,
                     dw_y=g_hh, parts=PARTS)


def timed(fn, reps=8):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    if beside:
        with torch.cuda.stream(rec):
            for _ in range(2):
                hip.rnn_bwd('lstm', dy, yy, wt, reserve, dxw=dd, workspace=ws, flags=hip.RNN_F16)
        torch.cuda._sleep(400000)
        reps = 4
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / reps


for name, fn in (('library form (column split, 2 TN GEMMs, rescales)', library),
                 ('own kernel (3 packs + 1 launch)', own)):
    g_ih.zero_(); g_hh.zero_()
    ms = timed(fn)
    g_ih.zero_(); g_hh.zero_()
    fn()
    ref = dxw[8000:, :GH].double().t() @ x[8000:].double()
    err = float((g_ih.double() - ref).norm() / ref.norm())
    print('{}: {:.3f} ms{}  (dW_ih rms error {:.2e})'.format(
        name + (' parts {}'.format(PARTS) if fn is own else ''), ms, ' beside a half-chip backward recurrence' if beside else '', err))

if not beside:
    scale, inv = hip.colscale_from_max(colmax)
    d_pk = hip.wgrad16_pack(dxw[8000:, :GH], ROWS, 0, stages, 1.0, col_scale=scale, out=bufs[0])
    x_pk = hip.wgrad16_pack(x[8000:], ROWS, 0, stages, 32768.0, out=bufs[1])
    y_pk = hip.wgrad16_pack(y[:, :H], 16000, 8000 - B, stages, 32768.0, out=bufs[2])
    parts = {
        'pack d (column scales)': lambda: hip.wgrad16_pack(dxw[8000:, :GH], ROWS, 0, stages, 1.0, col_scale=scale, out=bufs[0]),
        'pack x': lambda: hip.wgrad16_pack(x[8000:], ROWS, 0, stages, 32768.0, out=bufs[1]),
        'pack y': lambda: hip.wgrad16_pack(y[:, :H], 16000, 8000 - B, stages, 32768.0, out=bufs[2]),
        'kernel W_ih + W_hh': lambda: hip.wgrad16_gemm(d_pk, GH, stages, inv, x_pk, 0, 32768.0, g_ih, y_packed=y_pk,
                                                        y_scale=32768.0, dw_y=g_hh),
        'kernel W_ih + W_hh, 2 parts': lambda: hip.wgrad16_gemm(d_pk, GH, stages, inv, x_pk, 0, 32768.0, g_ih, y_packed=y_pk,
                                                        y_scale=32768.0, dw_y=g_hh, parts=2),
        'kernel W_ih + W_hh, 4 parts': lambda: hip.wgrad16_gemm(d_pk, GH, stages, inv, x_pk, 0, 32768.0, g_ih, y_packed=y_pk,
                                                        y_scale=32768.0, dw_y=g_hh, parts=4),
        'kernel W_ih + W_hh, 8 parts': lambda: hip.wgrad16_gemm(d_pk, GH, stages, inv, x_pk, 0, 32768.0, g_ih, y_packed=y_pk,
                                                        y_scale=32768.0, dw_y=g_hh, parts=8),
        'kernel W_ih + W_hh, 16 parts': lambda: hip.wgrad16_gemm(d_pk, GH, stages, inv, x_pk, 0, 32768.0, g_ih, y_packed=y_pk,
                                                        y_scale=32768.0, dw_y=g_hh, parts=16),
        'kernel W_ih only': lambda: hip.wgrad16_gemm(d_pk, GH, stages, inv, x_pk, 0, 32768.0, g_ih),
        'library: column split of d': lambda: split_gemm.wgrad16_operand(dxw[8000:, :GH], colmax),
    }
    d16, inv16 = split_gemm.wgrad16_operand(dxw[8000:, :GH], colmax)
    parts['library: W_ih GEMM + rescale'] = lambda: split_gemm.wgrad16(g_ih, d16, inv16, x16, 32768.0, 8000)
    parts['library: W_hh GEMM + rescale'] = lambda: split_gemm.wgrad16(g_hh, d16, inv16, y16, 32768.0, 8000 - B,
                                                                      x_cols=slice(0, H))
    for name, fn in parts.items():
        print('  {}: {:.3f} ms'.format(name, timed(fn, 16)))
