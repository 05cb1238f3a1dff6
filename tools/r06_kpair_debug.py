"""Round 6 debug: which shapes make repeated passes of the K-pair kernel differ?"""
import sys, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ctc_asr_amd import hip
hip.load()
DEV = 'cuda'
def trial(T, B, xcd, reps=6, like_test=False):
    g = torch.Generator(device=DEV).manual_seed(47)
    gh, H = 4096, 1024
    xw = torch.randn(T, B, 2, gh, device=DEV, generator=g) * 0.5
    w = torch.randn(2, gh, H, device=DEV, generator=g) / np.sqrt(H)
    dy = torch.randn(T, B, 2 * H, device=DEV, generator=g)
    if like_test:
        dy = dy * torch.logspace(-5, 0, B, device=DEV).view(1, B, 1)
    y, reserve, ws = hip.rnn_fwd('lstm', xw, w)
    wt = hip.transpose_batched(w)
    flags = hip.RNN_F16 | (hip.RNN_XCD_SPLIT if xcd else 0) | hip.RNN_KPAIR
    outs = []
    if like_test:
        hip.rnn_bwd('lstm', dy, y, wt, reserve, workspace=ws, flags=(flags & ~hip.RNN_KPAIR) | hip.RNN_STAGGER)
    for _ in range(reps):
        if like_test:
            db = torch.zeros(2 * gh, device=DEV)
            colmax = torch.zeros(2 * gh, dtype=torch.int32, device=DEV)
            dxw = torch.full((T, B, 2, gh), float('nan'), device=DEV)
            hip.rnn_bwd('lstm', dy, y, wt, reserve, None, dxw=dxw, dbias=db, workspace=ws, flags=flags, colmax=colmax)
            hip.rnn_poll_error('lstm', ws, T, B, H)
            outs.append(dxw)
        else:
            outs.append(hip.rnn_bwd('lstm', dy, y, wt, reserve, workspace=ws, flags=flags).clone())
    hip.rnn_poll_error('lstm', ws, T, B, H)
    diffs = [int((o != outs[0]).sum()) for o in outs[1:]]
    where = ''
    for o in outs[1:]:
        bad = (o != outs[0]).nonzero()
        if len(bad):
            where = 'steps {} rows {} dirs {}'.format(sorted(set(bad[:, 0].tolist()))[:8],
                                                      sorted(set(bad[:, 1].tolist()))[:8],
                                                      sorted(set(bad[:, 2].tolist())))
            break
    if where:
        o = [x for x in outs[1:] if (x != outs[0]).any()][0]
        bad = (o != outs[0]).nonzero()
        i = tuple(bad[0].tolist())
        where += ' first {} {:.9g} vs {:.9g}; units {}'.format(i, float(o[i]), float(outs[0][i]), sorted(set((bad[:, 3] % 1024).tolist()))[:12])
    print('T {:3d} B {:2d} xcd {}: differing elements per repeat {} {}'.format(T, B, xcd, diffs, where), flush=True)
for T, B in ((7, 17), (8, 17), (7, 32), (13, 32), (12, 32), (61, 27)):
    for xcd in (0, 1):
        trial(T, B, xcd, like_test=True)
