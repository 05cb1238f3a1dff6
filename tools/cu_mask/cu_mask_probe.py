#!/usr/bin/env python
"""Map CU-mask bits of hipExtStreamCreateWithCUMask to XCDs on this GPU: for a few masks, launch 128
one-wave workgroups that each need a whole CU's LDS and report the XCC_ID histogram.
    hipcc --offload-arch=gfx950 -shared -fPIC tools/cu_mask/cu_mask_probe.hip -o tools/cu_mask/probe.so
    python tools/cu_mask/cu_mask_probe.py"""
import collections
import ctypes
import os

import torch

here = os.path.dirname(os.path.abspath(__file__))
lib = ctypes.CDLL(os.path.join(here, 'probe.so'))
lib.probe_stream_with_mask.restype = ctypes.c_void_p
lib.probe_stream_with_mask.argtypes = [ctypes.c_void_p, ctypes.c_uint32]
lib.probe_where.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p,
                            ctypes.c_void_p, ctypes.c_int]
torch.zeros(1, device='cuda')


def run(name, bits, blocks=128):
    words = (ctypes.c_uint32 * 8)()
    for b in bits:
        words[b // 32] |= 1 << (b % 32)
    stream = lib.probe_stream_with_mask(words, 8)
    assert stream, 'hipExtStreamCreateWithCUMask failed'
    xcc = torch.full((blocks,), -1, dtype=torch.int32, device='cuda')
    hwid = torch.zeros(blocks, dtype=torch.int32, device='cuda')
    torch.cuda.synchronize()
    assert lib.probe_where(stream, blocks, 100 * 1024, xcc.data_ptr(), hwid.data_ptr(), 2000000) == 0
    torch.cuda.synchronize()
    hist = collections.Counter(xcc.cpu().tolist())
    print('{:40s} bits {:3d}  XCC histogram {}'.format(name, len(bits), dict(sorted(hist.items()))),
          flush=True)


run('all 256 bits', range(256), 256)
run('bits 0..127', range(128))
run('bits 128..255', range(128, 256))
run('even bits', range(0, 256, 2))
run('bits with (b % 8) < 4', [b for b in range(256) if b % 8 < 4])
run('bits with (b // 32) % 2 == 0', [b for b in range(256) if (b // 32) % 2 == 0])
run('bits with (b % 16) < 8', [b for b in range(256) if b % 16 < 8])
