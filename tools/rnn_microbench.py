#!/usr/bin/env python
"""Times ctcasr_rnn_fwd / ctcasr_rnn_bwd alone (C2 shape by default) and, with
CTCASR_RNN_PROF=1, prints the per-phase timings workgroup 0 of the persistent kernel recorded.

    python tools/rnn_microbench.py [T B H]
"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ctc_asr_amd import hip  # noqa: E402


def main():
    T, B, H = (int(v) for v in sys.argv[1:4]) if len(sys.argv) >= 4 else (500, 16, 1024)
    cell = sys.argv[4] if len(sys.argv) >= 5 else 'lstm'
    G = hip.CELL_GATES[cell]
    hip.load(os.environ.get('CTCASR_LIB'))      # A/B builds of the library
    # CTCASR_FULL: backward recurrence on the whole chip (default: half);
    # CTCASR_FWD_HALF: forward recurrence on half of the chip (default: whole)
    bwd_flags = hip.RNN_WHOLE_CHIP if os.environ.get('CTCASR_FULL') else hip.RNN_DEFAULT
    fwd_flags = hip.RNN_HALF_CHIP if os.environ.get('CTCASR_FWD_HALF') else hip.RNN_DEFAULT
    if os.environ.get('CTCASR_ONE_BARRIER'):   # B > 16: round-1 kernels (one barrier for both tiles)
        bwd_flags |= hip.RNN_ONE_BARRIER
        fwd_flags |= hip.RNN_ONE_BARRIER
    if os.environ.get('CTCASR_RS'):            # backward: reduce-scatter form (LSTM-1024)
        bwd_flags |= hip.RNN_REDUCE_SCATTER
    if os.environ.get('CTCASR_F16'):           # both recurrences on the fp16 matrix pipe
        fwd_flags |= hip.RNN_F16
        bwd_flags |= hip.RNN_F16
    if os.environ.get('CTCASR_XCD'):           # fp16 kernels: one direction per half of the XCDs
        fwd_flags |= hip.RNN_XCD_SPLIT
        bwd_flags |= hip.RNN_XCD_SPLIT
    if os.environ.get('CTCASR_STAGGER'):       # fp16 LSTM-1024 backward, 17..32 rows: staggered tiles
        bwd_flags |= hip.RNN_STAGGER
    if os.environ.get('CTCASR_KPAIR'):         # ... with the K axis split over pairs of workgroups
        bwd_flags |= hip.RNN_KPAIR
    g = torch.Generator(device='cuda').manual_seed(0)
    xw = torch.randn(T, B, 2, G * H, device='cuda', generator=g) * 0.5
    w = torch.randn(2, G * H, H, device='cuda', generator=g) / np.sqrt(H)
    dy = torch.randn(T, B, 2 * H, device='cuda', generator=g)
    wt = hip.transpose_batched(w)
    b_hh = torch.randn(2, G * H, device='cuda', generator=g) * 0.3 if cell == 'gru' else None
    y, reserve, ws = hip.rnn_fwd(cell, xw, w, flags=fwd_flags, b_hh_n=b_hh)
    dxw = hip.rnn_bwd(cell, dy, y, wt, reserve, workspace=ws, flags=bwd_flags)
    hip.rnn_poll_error(cell, ws, T, B, H)
    for name, fn in (('fwd', lambda: hip.rnn_fwd(cell, xw, w, y=y, reserve=reserve, workspace=ws,
                                              flags=fwd_flags, b_hh_n=b_hh)),
                     ('bwd', lambda: hip.rnn_bwd(cell, dy, y, wt, reserve, dxw=dxw, workspace=ws,
                                              flags=bwd_flags))):
        start, stop = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 5
        start.record()
        for _ in range(reps):
            fn()
        stop.record()
        torch.cuda.synchronize()
        ms = start.elapsed_time(stop) / reps
        print('{}: {:.3f} ms per call, {:.2f} us per time step'.format(name, ms, ms * 1e3 / T))
        if os.environ.get('CTCASR_RNN_PROF'):
            state = (6 * B * H * 4 + 255) // 256 * 256
            base = state + 9216 + 256 + (0 if name == "fwd" else 32)   # SyncWords: counters, error
            words = ws[base: base + 32].cpu().numpy().view(np.uint64)
            labels = ['wait', 'loads+mfma', 'reduce+gates+publish', 'drain+arrive']
            if name == 'bwd' and os.environ.get('CTCASR_RS'):
                words = ws[base: base + 40].cpu().numpy().view(np.uint64)
                labels = ['wait', 'partial loads+sum', 'gates+A operand', 'mfma+publish',
                          'drain+arrive+dxw']
            if name == 'bwd' and os.environ.get('CTCASR_KPAIR') and H == 1024:
                words = ws[base: base + 72].cpu().numpy().view(np.uint64)
                labels = ['marker wait', 'poll wait', 'main loops (rest)',
                          'reduce+hand-off+gates+publish', 'spins x 100', 'of which hand-off wait',
                          'wait for the first granules', 'request -> first granules',
                          'request -> arrival point (all landed)']
            elif name == 'bwd' and os.environ.get('CTCASR_STAGGER'):
                words = ws[base: base + 40].cpu().numpy().view(np.uint64)
                labels = ['marker wait', 'poll wait', 'main loops (rest)', 'reduce+gates+publish',
                          'spins x 100']
            print('  wg0 phases (us/step): ' + ', '.join(
                '{} {:.2f}'.format(l, float(w_) / 100.0 / T) for l, w_ in zip(labels, words)))
            if name == 'fwd' and B > 16:
                more = ws[state + 9216 + 256: state + 9216 + 256 + 128].cpu().numpy() \
                    .view(np.uint64).astype(np.float64) / 100.0 / T
                print('  chain 0: {}  (partials+barrier inside phase 2: {:.2f})'.format(
                    np.round(more[0:4], 2).tolist(), more[4]))
                print('  chain 1: {}  (partials+barrier inside phase 2: {:.2f})'.format(
                    np.round(more[8:12], 2).tolist(), more[12]))
            if name == 'bwd' and os.environ.get('CTCASR_RS'):
                every = ws[state + 9216 + 256 + 128: state + 9216 + 256 + 128 + 256 * 32] \
                    .cpu().numpy().view(np.uint64).reshape(256, 4).astype(np.float64) / 100.0 / T
                chains = 2 if B > 16 else 1
                every = every[:128 * chains]
                for ch in range(chains):
                    part = every[ch::chains]
                    print('    chain {} over 128 workgroups (us/step)'.format(ch))
                    for k, label in enumerate(['wait', 'partial loads+sum', 'gates+mfma+publish',
                                               'drain+arrive+dxw']):
                        col = part[:, k]
                        print('      {:<20s} min {:.2f}  median {:.2f}  max {:.2f}'.format(
                            label, col.min(), np.median(col), col.max()))
                    busy = part[:, 1] + part[:, 2] + part[:, 3]
                    print('      busy (all but wait)  min {:.2f}  median {:.2f}  max {:.2f}; '
                          'per blockIdx % 8: {}'.format(
                              busy.min(), np.median(busy), busy.max(),
                              np.round([busy[i::8].mean() for i in range(8)], 2).tolist()))
            if name == 'bwd' and H == 2048 and cell == 'lstm' and os.environ.get('CTCASR_F16'):
                # (the last launch of the pass: tile 0 / the last tile, direction 1)
                fbase = state + 9216 + 256
                every = ws[fbase + 128: fbase + 128 + 256 * 32].cpu().numpy().view(np.uint64) \
                    .reshape(256, 4).astype(np.float64) / 100.0 / T
                for k, label in enumerate(labels[:4]):
                    col = every[:, k]
                    print('    all 256 workgroups, {:<22s} min {:.2f}  median {:.2f}  max {:.2f}'
                          .format(label, col.min(), np.median(col), col.max()))
                busy = every[:, 1] + every[:, 2] + every[:, 3]
                print('    busy (all but wait): min {:.2f} median {:.2f} max {:.2f}; per slice % 8: {}; '
                      'per (slice >> 3) & 3: {}'.format(
                          busy.min(), np.median(busy), busy.max(),
                          np.round([busy[i::8].mean() for i in range(8)], 2).tolist(),
                          np.round([busy[[b for b in range(256) if (b >> 3) & 3 == j]].mean()
                                    for j in range(4)], 2).tolist()))
                order = np.argsort(busy)
                print('    busiest workgroups: {}  us {}'.format(
                    order[-8:].tolist(), np.round(busy[order[-8:]], 2).tolist()))
                for k, label in enumerate(labels[:4]):
                    print('      {:<22s} of the busiest 8: {}'.format(
                        label, np.round(every[order[-8:], k], 2).tolist()))
            if name == 'fwd':
                every = ws[base + 128: base + 128 + 256 * 32].cpu().numpy().view(np.uint64) \
                    .reshape(256, 4).astype(np.float64) / 100.0 / T
                for k, label in enumerate(labels):
                    col = every[:, k]
                    print('    all 256 workgroups, {:<22s} min {:.2f}  median {:.2f}  max {:.2f}'
                          .format(label, col.min(), np.median(col), col.max()))
                busy = every[:, 1] + every[:, 2] + every[:, 3]
                order = np.argsort(busy)
                print('    busiest (loads..arrive) workgroups: {}  us {}'.format(
                    order[-8:].tolist(), np.round(busy[order[-8:]], 2).tolist()))
                print('    per blockIdx % 8 mean busy: {}'.format(
                    np.round([busy[i::8].mean() for i in range(8)], 2).tolist()))
    hip.rnn_poll_error(cell, ws, T, B, H)
    # checksum for A/B comparisons between variants
    print('checksum y {:.6f} dxw {:.6f}'.format(float(y.double().abs().sum()),
                                                 float(dxw.double().abs().sum())))


if __name__ == '__main__':
    main()
