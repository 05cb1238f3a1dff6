#!/usr/bin/env python
"""Where does prnn_bwd16s_kernel (CTCASR_RNN_STAGGER) differ from the one-barrier kernel?"""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ctc_asr_amd import hip

T, B = (int(v) for v in sys.argv[1:3]) if len(sys.argv) >= 3 else (12, 32)
H, GH = 1024, 4096
hip.load()
g = torch.Generator(device='cuda').manual_seed(43)
xw = torch.randn(T, B, 2, GH, device='cuda', generator=g) * 0.5
w = torch.randn(2, GH, H, device='cuda', generator=g) / np.sqrt(H)
dy = torch.randn(T, B, 2 * H, device='cuda', generator=g)
y, reserve, ws = hip.rnn_fwd('lstm', xw, w)
wt = hip.transpose_batched(w)
a = hip.rnn_bwd('lstm', dy, y, wt, reserve, workspace=ws, flags=hip.RNN_F16)
a2 = hip.rnn_bwd('lstm', dy, y, wt, reserve, workspace=ws, flags=hip.RNN_F16)
b = hip.rnn_bwd('lstm', dy, y, wt, reserve, workspace=ws, flags=hip.RNN_F16 | hip.RNN_STAGGER)
b2 = hip.rnn_bwd('lstm', dy, y, wt, reserve, workspace=ws, flags=hip.RNN_F16 | hip.RNN_STAGGER)
hip.rnn_poll_error('lstm', ws, T, B, H)
print('base repeat equal', torch.equal(a, a2), 'stagger repeat equal', torch.equal(b, b2))
d = (a - b).abs()
print('max abs diff', float(d.max()), 'max |a|', float(a.abs().max()), 'nonzero frac', float((d > 0).float().mean()))
for t in range(T):
    for dr in range(2):
        dd = d[t, :, dr]
        if float(dd.max()) > 0:
            rows = (dd.amax(dim=1) > 0).nonzero().flatten().tolist()
            print('t', t, 'dir', dr, 'max', float(dd.max()), 'rel', float(dd.max() / a[t, :, dr].abs().max()), 'rows', rows[:8], len(rows),
                  'cols', int((dd.amax(dim=0) > 0).sum()))
