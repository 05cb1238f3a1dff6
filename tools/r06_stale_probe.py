"""Round 6: do the staggered-tile kernels read stale exchange lines when the two 16-row tiles share
cache lines (B not a multiple of 8)?  Pass 1 with data A, pass 2 with data B on the SAME workspace,
against data B on a fresh workspace with the one-barrier kernel."""
import sys, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ctc_asr_amd import hip
hip.load(os.environ.get('CTCASR_LIB'))
DEV = 'cuda'
H, gh = 1024, 4096
def data(T, B, seed):
    g = torch.Generator(device=DEV).manual_seed(seed)
    xw = torch.randn(T, B, 2, gh, device=DEV, generator=g) * 0.5
    w = torch.randn(2, gh, H, device=DEV, generator=g) / np.sqrt(H)
    dy = torch.randn(T, B, 2 * H, device=DEV, generator=g)
    return xw, w, dy
def trial(T, B, xcd, extra, name, reps=4, first=None):
    base = hip.RNN_F16 | (hip.RNN_XCD_SPLIT if xcd else 0)
    worst = 0.0
    count = 0
    for rep in range(reps):
        xa, wa, dya = data(T, B, 100 + rep)
        xb, wb, dyb = data(T, B, 200 + rep)
        ya, ra, ws = hip.rnn_fwd('lstm', xa, wa)
        hip.rnn_bwd('lstm', dya, ya, hip.transpose_batched(wa), ra, workspace=ws, flags=base | (extra if first is None else first))
        yb, rb, _ = hip.rnn_fwd('lstm', xb, wb, workspace=ws)
        got = hip.rnn_bwd('lstm', dyb, yb, hip.transpose_batched(wb), rb, workspace=ws, flags=base | extra)
        hip.rnn_poll_error('lstm', ws, T, B, H)
        yf, rf, wsf = hip.rnn_fwd('lstm', xb, wb)
        want = hip.rnn_bwd('lstm', dyb, yf, hip.transpose_batched(wb), rf, workspace=wsf, flags=base)
        hip.rnn_poll_error('lstm', wsf, T, B, H)
        err = float((got - want).abs().max() / want.abs().max())
        worst = max(worst, err)
        count += int(((got - want).abs() > 1e-4 * want.abs().max()).sum())
    print('{:9s} T {:3d} B {:2d} xcd {}: worst |diff| / max {:.2e}, elements off by > 1e-4: {}'.format(
        name, T, B, xcd, worst, count), flush=True)
for B in (17, 19, 20):
    trial(40, B, 0, hip.RNN_KPAIR, os.environ.get('CTCASR_LIB', 'default')[-12:-3])
