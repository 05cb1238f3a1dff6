#!/usr/bin/env python
"""Time every ctc_asr_amd/csrc/_obj/sg_*.so (tools/split_gemm_variants.sh) on the C3 forward
projection [16000 x 2048] x [8192 x 2048]^T and the data-gradient shape."""
import ctypes
import glob
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
shapes = [(16000, 8192, 2048), (16000, 2048, 8192)]
g = torch.Generator(device='cuda').manual_seed(0)
for path in sorted(glob.glob(os.path.join(ROOT, 'ctc_asr_amd', 'csrc', '_obj', 'sg_*.so'))):
    lib = ctypes.CDLL(path)
    fn = lib.ctcasr_gemm_split_nt
    fn.restype = ctypes.c_int
    fn.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p,
                   ctypes.c_int64] + [ctypes.c_int] * 4 + [ctypes.c_void_p]
    row = []
    for m, n, k in shapes:
        x = torch.randn(m, k, device='cuda', generator=g)
        w = torch.randn(n, k, device='cuda', generator=g) / k ** 0.5
        out = torch.empty(m, n, device='cuda')
        stream = torch.cuda.current_stream().cuda_stream
        call = lambda: fn(x.data_ptr(), k, w.data_ptr(), k, out.data_ptr(), n, m, n, k, 0, stream)
        for _ in range(5):
            assert call() == 0
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(20):
            call()
        b.record()
        torch.cuda.synchronize()
        ms = a.elapsed_time(b) / 20
        ref = x[:256].double() @ w.double().t()
        err = float((out[:256].double() - ref).abs().max())
        row.append('{:.3f} ms ({:.0f} TF, err {:.1e})'.format(ms, 12.0 * m * n * k / ms / 1e9, err))
    print('{:28s} {}'.format(os.path.basename(path), '   '.join(row)), flush=True)
