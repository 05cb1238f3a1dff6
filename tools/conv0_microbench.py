#!/usr/bin/env python
"""First DS2 convolution (1 -> 32 channels, 11 x 41, stride (2, 2)): the fp32-MFMA kernels of
conv.hip against the fp16-pipe kernels of conv16.hip, forward and kernel gradient.
    python tools/conv0_microbench.py [batch frames]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ctc_asr_amd import hip  # noqa: E402


def timed(fn, reps=20):
    for _ in range(3):
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
    batch, frames = (int(v) for v in sys.argv[1:3]) if len(sys.argv) >= 3 else (32, 999)
    hip.load(os.environ.get('CTCASR_LIB'))
    gen = torch.Generator(device='cuda').manual_seed(0)
    x = torch.randn(batch, frames, 80, device='cuda', generator=gen)
    w = torch.randn(32, 1, 11, 41, device='cuda', generator=gen) * 0.1
    bias = torch.randn(32, device='cuda', generator=gen)
    t_out = (frames + 1) // 2
    dz = torch.randn(batch, t_out, 40, 32, device='cuda', generator=gen)
    y = torch.empty(batch, t_out, 40, 32, device='cuda')
    dw = torch.empty(32, 1, 11, 41, device='cuda')
    flops = 2.0 * batch * t_out * 40 * 32 * 11 * 41
    packed = hip.conv0_pack_weights16(w)
    rows = [('conv0_fwd', lambda: hip.conv0_fwd(x, w, bias, y, relu_cutoff=20.0)),
            ('conv0_pack_weights16', lambda: hip.conv0_pack_weights16(w, packed)),
            ('conv0_fwd16', lambda: hip.conv0_fwd16(x, packed, bias, y, relu_cutoff=20.0)),
            ('conv0_wrw', lambda: hip.conv0_wrw(dz, x, dw)),
            ('conv0_wrw16 (all passes)', lambda: hip.conv0_wrw16(dz, x, dw))]
    for name, fn in rows:
        ms = timed(fn)
        print('{:28s} {:.3f} ms  {:.1f} TFLOP/s fp32-equivalent'.format(name, ms, flops / ms / 1e9))


main()
