#!/usr/bin/env python
"""Round 6: the half-chip backward recurrence (B = 32, H = 1024, staggered tiles) alone and beside
synthetic co-runners on the 128 free CUs - which resource of a neighbour costs it time?
    hipcc --offload-arch=gfx950 -O3 -shared -fPIC tools/beside/beside_kernels.hip -o tools/beside/beside.so
    python tools/beside/beside_probe.py"""
import ctypes, os, sys
import torch
here = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(here)))
from ctc_asr_amd import hip
hip.load()
lib = ctypes.CDLL(os.path.join(here, 'beside.so'))
lib.beside_launch.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_size_t,
                              ctypes.c_void_p, ctypes.c_void_p]
T, B, H = 500, 32, 1024
g = torch.Generator(device='cuda').manual_seed(0)
xw = torch.randn(T, B, 2, 4 * H, device='cuda', generator=g) * 0.5
w = torch.randn(2, 4 * H, H, device='cuda', generator=g) / 32
dy = torch.randn(T, B, 2 * H, device='cuda', generator=g)
wt = hip.transpose_batched(w)
flags = hip.RNN_F16 | hip.RNN_XCD_SPLIT | hip.RNN_STAGGER
y, reserve, ws = hip.rnn_fwd('lstm', xw, w)
dxw = hip.rnn_bwd('lstm', dy, y, wt, reserve, workspace=ws, flags=flags)
big = torch.zeros(1 << 28, device='cuda')            # 1 GB
sink = torch.zeros(4, device='cuda')
side = torch.cuda.Stream()
ticket = [0]
def run(kind, blocks=128):
    times = []
    for _ in range(4):
        torch.cuda.synchronize()
        ticket[0] += 1
        ready = torch.cuda.Event(); ready.record()
        r0, r1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        r0.record()
        hip.rnn_bwd('lstm', dy, y, wt, reserve, dxw=dxw, workspace=ws, flags=flags, ticket=ticket[0])
        r1.record()
        if kind is not None:
            with torch.cuda.stream(side):
                side.wait_event(ready)
                hip.rnn_resident_gate('lstm', ws, T, B, H, ticket[0], 300)
                rc = lib.beside_launch(kind, blocks, 3400, big.data_ptr(), big.numel() * 4, sink.data_ptr(),
                                       side.cuda_stream)
                assert rc == 0
        torch.cuda.synchronize()
        times.append(r0.elapsed_time(r1) * 1e3 / T)
    return min(times), sorted(times)[len(times) // 2]
print('alone                 : {:.2f} us per time step (median {:.2f})'.format(*run(None)))
for kind, name in ((0, 'MFMA on registers'), (1, 'VALU on registers'), (2, 'LDS reads'), (3, 'HBM streaming reads')):
    for blocks in (128, 64):
        print('{:22s}: {:.2f} us per time step (median {:.2f}) beside {} workgroups'.format(
            name, *run(kind, blocks), blocks), flush=True)
hip.rnn_poll_error('lstm', ws, T, B, H)
