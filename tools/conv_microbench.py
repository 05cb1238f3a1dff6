#!/usr/bin/env python
"""Times ctcasr_conv_s12_bwd_data at the C2 shape (B=16, T'=500) against MIOpen's backward-data
of the same layer (autotuned, explicitly padded input, incl. the interior copy it needs)."""
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
    batch, frames = (int(v) for v in sys.argv[1:3]) if len(sys.argv) >= 3 else (16, 500)
    hip.load(os.environ.get('CTCASR_LIB'))
    torch.backends.cudnn.benchmark = True
    gen = torch.Generator(device='cuda').manual_seed(0)
    dz = torch.randn(batch, frames, 20, 32, device='cuda', generator=gen)
    weight = torch.randn(32, 32, 11, 21, device='cuda', generator=gen) * 0.05
    packed = hip.conv_s12_pack_weights(weight)
    out = torch.empty(batch, frames, 40, 32, device='cuda')
    x = torch.randn(batch, frames, 40, 32, device='cuda', generator=gen)
    y = torch.empty(batch, frames, 20, 32, device='cuda')
    bias = torch.randn(32, device='cuda', generator=gen)
    flops = 2.0 * batch * frames * 20 * 32 * 32 * 11 * 21
    ms = timed(lambda: hip.conv_s12_pack_weights(weight, packed))
    print('pack weights: {:.3f} ms'.format(ms))
    ms = timed(lambda: hip.conv_s12_fwd(x, packed, 32, bias, y))
    print('conv_s12_fwd: {:.3f} ms  {:.1f} TFLOP/s'.format(ms, flops / ms / 1e9))
    ms = timed(lambda: hip.conv_s12_bwd_data(dz, packed, out))
    print('conv_s12_bwd_data: {:.3f} ms  {:.1f} TFLOP/s'.format(ms, flops / ms / 1e9))
    dw = torch.empty(32, 32, 11, 21, device='cuda')
    ms = timed(lambda: hip.conv_s12_wrw(dz, x, dw))
    print('conv_s12_wrw: {:.3f} ms  {:.1f} TFLOP/s'.format(ms, flops / ms / 1e9))
    xb = x.clamp(0.0, 20.0)
    ms = timed(lambda: hip.conv_s12_wrw16(dz, xb, 2.0 ** 11, dw))
    print('conv_s12_wrw16: {:.3f} ms  {:.1f} TFLOP/s fp32-equivalent'.format(
        ms, flops / ms / 1e9))
    if os.environ.get('CONV_MICROBENCH_OWN_ONLY') == '1':
        return
    xp = torch.zeros(batch, 32, frames + 10, 59, device='cuda') \
        .contiguous(memory_format=torch.channels_last)
    w_cl = weight.contiguous(memory_format=torch.channels_last)
    dz_nchw = dz.permute(0, 3, 1, 2)

    def library():
        dxp, _, _ = torch.ops.aten.convolution_backward(
            dz_nchw, xp, w_cl, [32], [1, 2], [0, 0], [1, 1], False, [0, 0], 1,
            [True, False, False])
        return dxp[:, :, 5:5 + frames, 9:49].permute(0, 2, 3, 1).contiguous()
    ms = timed(library)
    print('MIOpen bwd-data + interior copy: {:.3f} ms  {:.1f} TFLOP/s'.format(ms, flops / ms / 1e9))
    print('bwd-data max |diff| {:.2e}'.format(float((library() - out).abs().max())))
    xpad = torch.zeros(batch, 32, frames + 10, 59, device='cuda') \
        .contiguous(memory_format=torch.channels_last)
    xpad[:, :, 5:5 + frames, 9:49] = x.permute(0, 3, 1, 2)

    def library_fwd():
        padded = torch.nn.functional.pad(x.permute(0, 3, 1, 2), (9, 10, 5, 5)) \
            .contiguous(memory_format=torch.channels_last)
        return torch.ops.aten.convolution(padded, w_cl, bias, [1, 2], [0, 0], [1, 1], False,
                                          [0, 0], 1)
    ms = timed(library_fwd)
    print('MIOpen fwd incl. padding copy: {:.3f} ms  {:.1f} TFLOP/s'.format(ms, flops / ms / 1e9))
    print('fwd max |diff| {:.2e}'.format(
        float((library_fwd().permute(0, 2, 3, 1) - y).abs().max())))
    def library_wrw():
        _, dwl, _ = torch.ops.aten.convolution_backward(
            dz_nchw, xpad, w_cl, [32], [1, 2], [0, 0], [1, 1], False, [0, 0], 1,
            [False, True, False])
        return dwl
    ms = timed(library_wrw)
    print('MIOpen wrw (padded input given): {:.3f} ms  {:.1f} TFLOP/s'.format(ms, flops / ms / 1e9))
    print('wrw max |diff| {:.2e} of {:.2e}'.format(float((library_wrw() - dw).abs().max()),
                                                    float(dw.abs().max())))
    # first layer: 1 -> 32 channels, 11x41, stride (2, 2)
    feats = torch.randn(batch, 2 * frames - 1, 80, device='cuda', generator=gen)
    w0 = torch.randn(32, 1, 11, 41, device='cuda', generator=gen) * 0.1
    flops0 = 2.0 * batch * frames * 40 * 32 * 11 * 41
    ms = timed(lambda: hip.conv0_fwd(feats, w0, bias))
    print('conv0_fwd: {:.3f} ms  {:.1f} TFLOP/s'.format(ms, flops0 / ms / 1e9))
    w0_cl = w0.contiguous(memory_format=torch.channels_last)

    def library_conv0():
        padded = torch.nn.functional.pad(feats.unsqueeze(1), (19, 20, 5, 5)) \
            .contiguous(memory_format=torch.channels_last)
        return torch.ops.aten.convolution(padded, w0_cl, bias, [2, 2], [0, 0], [1, 1], False,
                                          [0, 0], 1).contiguous(memory_format=torch.channels_last)
    ms = timed(library_conv0)
    print('MIOpen conv0 fwd incl. padding copy: {:.3f} ms'.format(ms))
    print('conv0 max |diff| {:.2e}'.format(float(
        (library_conv0().permute(0, 2, 3, 1) - hip.conv0_fwd(feats, w0, bias)).abs().max())))
    dz0 = torch.randn(batch, frames, 40, 32, device='cuda', generator=gen)
    dw0 = torch.empty(32, 1, 11, 41, device='cuda')
    ms = timed(lambda: hip.conv0_wrw(dz0, feats, dw0))
    print('conv0_wrw: {:.3f} ms  {:.1f} TFLOP/s'.format(ms, flops0 / ms / 1e9))
    padded0 = torch.nn.functional.pad(feats.unsqueeze(1), (19, 20, 5, 5)) \
        .contiguous(memory_format=torch.channels_last)
    dz0_nchw = dz0.permute(0, 3, 1, 2)

    def library_wrw0():
        return torch.ops.aten.convolution_backward(dz0_nchw, padded0, w0_cl, [32], [2, 2], [0, 0],
                                                   [1, 1], False, [0, 0], 1,
                                                   [False, True, False])[1]
    ms = timed(library_wrw0)
    print('MIOpen conv0 wrw: {:.3f} ms'.format(ms))
    print('conv0 wrw max |diff| {:.2e} of {:.2e}'.format(
        float((library_wrw0() - dw0).abs().max()), float(dw0.abs().max())))
    # the reference stack's third layer: 32 -> 96 channels on 20 frequencies
    w3 = torch.randn(96, 32, 11, 21, device='cuda', generator=gen) * 0.05
    packed3 = hip.conv_s12_pack_weights(w3)
    x3 = torch.randn(batch, frames, 20, 32, device='cuda', generator=gen)
    dz3 = torch.randn(batch, frames, 10, 96, device='cuda', generator=gen)
    flops3 = 2.0 * batch * frames * 10 * 96 * 32 * 11 * 21
    ms = timed(lambda: hip.conv_s12_fwd(x3, packed3, 96))
    print('layer 3 fwd: {:.3f} ms  {:.1f} TFLOP/s'.format(ms, flops3 / ms / 1e9))
    ms = timed(lambda: hip.conv_s12_bwd_data(dz3, packed3))
    print('layer 3 bwd-data: {:.3f} ms  {:.1f} TFLOP/s'.format(ms, flops3 / ms / 1e9))


if __name__ == '__main__':
    main()
