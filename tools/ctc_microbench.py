#!/usr/bin/env python
"""Times ctcasr_ctc_loss_fwd_bwd alone at the C2 shape (T'=500, B=16, 29 classes, 150 labels)
and checks it against the C oracle."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ctc_asr_amd import hip  # noqa: E402
from oracle import cref  # noqa: E402


def main():
    T, B, C, L = 500, 16, 29, 150
    rng = np.random.default_rng(0)
    logits = (rng.normal(size=(T, B, C)) * 2).astype(np.float32)
    labels = [list(rng.integers(0, C - 1, size=L)) for _ in range(B)]
    flat = np.concatenate(labels).astype(np.int32)
    offsets = np.arange(0, (B + 1) * L, L, dtype=np.int32)
    seq = np.full(B, T, dtype=np.int32)
    d = lambda a, t=torch.float32: torch.as_tensor(a, dtype=t).cuda()
    lg, fl, of, sl = d(logits), d(flat, torch.int32), d(offsets, torch.int32), d(seq, torch.int32)
    ws = torch.empty(hip.ctc_loss_workspace_bytes(T, B, C, L), dtype=torch.uint8, device='cuda')
    loss, grad, status = hip.ctc_loss_fwd_bwd(lg, fl, of, sl, L, workspace=ws)
    ref_loss, ref_grad, _ = cref.ctc_loss(logits, labels, seq)
    print('max |loss - oracle| {:.3e}, max |grad - oracle| {:.3e}'.format(
        np.abs(loss.cpu().numpy() - ref_loss).max(), np.abs(grad.cpu().numpy() - ref_grad).max()))
    start, stop = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    start.record()
    for _ in range(20):
        hip.ctc_loss_fwd_bwd(lg, fl, of, sl, L, loss=loss, grad=grad, status=status, workspace=ws)
    stop.record()
    torch.cuda.synchronize()
    ms = start.elapsed_time(stop) / 20
    print('ctc_loss_fwd_bwd: {:.3f} ms per call, {:.3f} us per lattice step'.format(
        ms, ms * 1e3 / (2 * T)))


if __name__ == '__main__':
    main()
