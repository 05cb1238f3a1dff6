"""Times the block-scaled data-gradient kernel (csrc/dgrad16.hip) against the library path it
replaces (row split of dxw + fp16 GEMM over 3 K + rescale) at a recurrent layer's shape, and prints
both errors against float64.  python tools/dgrad16_probe.py [T] [B] [N]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
from ctc_asr_amd import hip, split_gemm            # noqa: E402
from test_gpu_dgrad16 import decades, publish, rel_errors      # noqa: E402

T, B, N = [int(v) for v in sys.argv[1:4]] + [500, 32, 2048][len(sys.argv) - 1:]
H = 1024
DEV = 'cuda'
gen = torch.Generator(device=DEV).manual_seed(1)
dxw = decades(T, B, gen)
w = torch.randn(8 * H, N, device=DEV, generator=gen) / np.sqrt(N)
ws = publish(hip, dxw)
d2 = dxw.view(T * B, 8 * H)
ref = d2.double() @ w.double()


def timed(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    start, stop = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    start.record()
    for _ in range(reps):
        out = fn()
    stop.record()
    torch.cuda.synchronize()
    return start.elapsed_time(stop) / reps, out


packed = hip.dgrad16_pack_weights(w, H, 2048.0)
out = torch.empty(T * B, N, device=DEV)
ms_pack, _ = timed(lambda: hip.dgrad16_pack_weights(w, H, 2048.0, out=packed))
ms_own, got = timed(lambda: hip.dgrad16_blockscaled(ws, T, B, H, packed, 2048.0, N, out=out))
wt = torch.empty(N, 8 * H, device=DEV)
hip.transpose_batched(w.view(1, 8 * H, N), out=wt.view(1, N, 8 * H))
wt16 = split_gemm.split16(wt, split_gemm.W_SCALE, split_gemm.H_B)
ms_lib, lib = timed(lambda: split_gemm.dgrad16(d2, wt16, split_gemm.W_SCALE))
ms_split, _ = timed(lambda: hip.split_f16_rows(d2, split_gemm.H_A))
ms_f32, f32 = timed(lambda: torch.mm(d2, w), reps=5)
flop = 2.0 * T * B * 8 * H * N
print('shape [{} x {}] x [{} x {}]'.format(T * B, 8 * H, 8 * H, N))
print('own kernel       {:.3f} ms  {:.0f} TFLOP/s fp32-equivalent ({:.0f} on the fp16 pipe)  '
      'rms / row error {:.2e} / {:.2e}'.format(ms_own, flop / ms_own * 1e-9, 3 * flop / ms_own * 1e-9,
                                               *rel_errors(got, ref)))
if os.environ.get('DG_PROF'):
    print('phase clocks per stage [wave][dma wait, barrier, issue, multiply]:')
    print(got[300, :32].view(8, 4).cpu().numpy().round(0))
print('pack of W_ih     {:.3f} ms'.format(ms_pack))
print('library path     {:.3f} ms (row split alone {:.3f})  rms / row error {:.2e} / {:.2e}'
      .format(ms_lib, ms_split, *rel_errors(lib, ref)))
print('fp32 library     {:.3f} ms  rms / row error {:.2e} / {:.2e}'.format(ms_f32,
                                                                          *rel_errors(f32, ref)))
