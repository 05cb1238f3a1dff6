#!/usr/bin/env python
"""The own split-GEMM kernel (ctcasr_gemm_split_nt: fp32 tiles split in registers, six bf16 MFMAs
per fragment pair) against the library path (split kernel + K-concatenated bf16 GEMM) and the
fp32 GEMM: time back to back and error against fp64, C3 layer shapes.
    python tools/split_gemm_kernel_probe.py"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ctc_asr_amd import hip, split_gemm as sg

hip.load()


def timed(fn, reps=30, warm=8):
    for _ in range(warm):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


def err(got, ref):
    d = got.double() - ref
    return float(d.pow(2).mean().sqrt() / ref.pow(2).mean().sqrt()), float(d.abs().max())


g = torch.Generator(device='cuda').manual_seed(0)
# correctness on an awkward shape first (partial tiles both ways, asymmetric operands)
a = torch.randn(300, 48, device='cuda', generator=g)
b = torch.randn(270, 48, device='cuda', generator=g)
ref = a.double() @ b.double().t()
got = hip.gemm_split_nt(a, b)
print('small [300 x 48] x [270 x 48]^T  rms rel / max abs', err(got, ref), flush=True)
acc = torch.ones(300, 270, device='cuda')
hip.gemm_split_nt(a, b, out=acc, accumulate=True)
print('accumulate', err(acc - 1.0, ref), flush=True)
report = {}
for name, m, n, k in (('forward projection', 16000, 8192, 2048),
                      ('data gradient', 16000, 2048, 8192),
                      ('layer 0 forward', 16000, 8192, 640),
                      ('C2 forward', 8000, 8192, 2048)):
    x = torch.randn(m, k, device='cuda', generator=g).clamp_(0, 20)
    w = torch.randn(n, k, device='cuda', generator=g) / k ** 0.5
    out = torch.empty(m, n, device='cuda')
    sub = slice(0, 512)
    ref = x[sub].double() @ w.double().t()
    xs, ws = sg.split(x, sg.A_ORDER), sg.split(w, sg.B_ORDER)

    def library():
        sg.split(x, sg.A_ORDER, out=xs)
        sg.mm_nt(xs, ws, out=out)
    flops = 2.0 * m * n * k
    row = {}
    t = timed(lambda: hip.gemm_split_nt(x, w, out=out))
    row['own kernel'] = {'ms': round(t, 3), 'raw_bf16_tflops': round(6 * flops / t / 1e9),
                         'err': err(out[sub], ref)}
    t = timed(library)
    row['split kernel + library bf16 GEMM'] = {'ms': round(t, 3), 'err': err(out[sub], ref)}
    t = timed(lambda: torch.mm(x, w.t(), out=out))
    row['library fp32 GEMM'] = {'ms': round(t, 3), 'err': err(out[sub], ref)}
    report['{} [{} x {}] x [{} x {}]^T'.format(name, m, k, n, k)] = row
    print(json.dumps({name: row}), flush=True)

# weight-gradient form: one third of the C3 steps, one direction
from ctc_asr_amd import split_gemm as sg2
rows = 5344
d = torch.randn(16000, 8192, device='cuda', generator=g) * 1e-3
x = torch.randn(16000, 2048, device='cuda', generator=g).clamp_(0, 20)
ds, xs = sg2.split(d, sg2.B_ORDER), sg2.split(x, sg2.A_ORDER)
for name, ncols in (('W_ih gradient [5344 x 4096]^T x [5344 x 2048]', 2048),
                    ('W_hh gradient [5344 x 4096]^T x [5344 x 1024]', 1024)):
    dw = torch.zeros(4096, ncols, device='cuda')
    ref = d[:rows, :4096].double().t()[:256] @ x[:rows, :ncols].double()
    row = {}
    t = timed(lambda: hip.gemm_split_tn(d[:rows, :4096], x[:rows, :ncols], dw, accumulate=False))
    row['own kernel (fp32 operands)'] = {'ms': round(t, 3), 'err': err(dw[:256], ref)}
    t = timed(lambda: sg2.mm_tn_rows(dw, ds, xs, 0, rows, a_cols=slice(0, 4096),
                                     b_cols=slice(0, ncols), accumulate=False))
    row['library bf16 GEMM on six-block pieces'] = {'ms': round(t, 3), 'err': err(dw[:256], ref)}
    t = timed(lambda: torch.mm(d[:rows, :4096].t(), x[:rows, :ncols], out=dw))
    row['library fp32 GEMM'] = {'ms': round(t, 3), 'err': err(dw[:256], ref)}
    print(json.dumps({name: row}), flush=True)
