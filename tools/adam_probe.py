"""Times ctcasr_adam_step over the C3 arena (122 M parameters): GB/s against the 8 TB/s of HBM3E.
    [CTCASR_LIB=...] python tools/adam_probe.py"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ctc_asr_amd import hip
hip.load(os.environ.get('CTCASR_LIB'))
n = 122_000_000
p, g, m, v = (torch.randn(n, device='cuda') * 0.01 for _ in range(4))
v.abs_()
for _ in range(3):
    hip.adam_step(p, g, m, v, 3)
torch.cuda.synchronize()
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
for i in range(20):
    hip.adam_step(p, g, m, v, 4 + i)
e.record(); torch.cuda.synchronize()
ms = s.elapsed_time(e) / 20
print('{}: {:.3f} ms per step, {:.2f} TB/s (28 bytes per parameter)'.format(
    os.environ.get('CTCASR_LIB', 'default')[-16:], ms, n * 28 / ms / 1e9))
