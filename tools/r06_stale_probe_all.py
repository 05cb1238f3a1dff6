"""Round 6: every persistent recurrence kernel, a pass over data A then data B on the SAME workspace
against data B on a fresh workspace (same kernel): bit-identical, or something read the pass before."""
import sys, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ctc_asr_amd import hip
hip.load()
DEV = 'cuda'
def data(cell, T, B, H, seed):
    G = hip.CELL_GATES[cell]
    g = torch.Generator(device=DEV).manual_seed(seed)
    xw = torch.randn(T, B, 2, G * H, device=DEV, generator=g) * 0.5
    w = torch.randn(2, G * H, H, device=DEV, generator=g) / np.sqrt(H)
    dy = torch.randn(T, B, 2 * H, device=DEV, generator=g)
    bh = torch.randn(2, G * H, device=DEV, generator=g) * 0.3 if cell == 'gru' else None
    return xw, w, hip.transpose_batched(w), dy, bh
def both(cell, d, ws, ff, bf):
    xw, w, wt, dy, bh = d
    y, res, ws = hip.rnn_fwd(cell, xw, w, workspace=ws, flags=ff, b_hh_n=bh) if ws is not None else \
        hip.rnn_fwd(cell, xw, w, flags=ff, b_hh_n=bh)
    dxw = hip.rnn_bwd(cell, dy, y, wt, res, workspace=ws, flags=bf)
    return y, dxw, ws
for cell, H in (('lstm', 1024), ('gru', 1024), ('rnn_relu', 2048), ('rnn_tanh', 2048), ('lstm', 2048), ('gru', 2048)):
    for B in (17, 19, 27, 32, 9):
        for name, ff, bf in (('fp32', 0, 0), ('fp32 whole', 0, hip.RNN_WHOLE_CHIP), ('fp32 1bar', hip.RNN_ONE_BARRIER, hip.RNN_ONE_BARRIER),
                             ('f16', hip.RNN_F16 | hip.RNN_XCD_SPLIT, hip.RNN_F16 | hip.RNN_XCD_SPLIT | hip.RNN_STAGGER | hip.RNN_KPAIR),
                             ('f16 half fwd', hip.RNN_F16 | hip.RNN_HALF_CHIP, hip.RNN_F16)):
            T = 24
            if not hip.rnn_persistent_supported(cell, T, B, H):
                continue
            bad = []
            for rep in range(2):
                a, b = data(cell, T, B, H, 10 + rep), data(cell, T, B, H, 20 + rep)
                _, _, ws = both(cell, a, None, ff, bf)
                y2, d2, _ = both(cell, b, ws, ff, bf)
                hip.rnn_poll_error(cell, ws, T, B, H)
                y1, d1, ws1 = both(cell, b, None, ff, bf)
                hip.rnn_poll_error(cell, ws1, T, B, H)
                if not torch.equal(y1, y2): bad.append('y {}'.format(int((y1 != y2).sum())))
                if not torch.equal(d1, d2): bad.append('dxw {}'.format(int((d1 != d2).sum())))
            print('{:8s} H {} B {:2d} {:12s}: {}'.format(cell, H, B, name, 'ok' if not bad else 'DIFFERS ' + ', '.join(bad)), flush=True)
