#!/usr/bin/env python
"""Times ctcasr_ctc_beam_decode / ctcasr_ctc_greedy_decode alone on logits shaped like a trained
model's (peaked: most frames blank-dominated):  python tools/beam_microbench.py [T B]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ctc_asr_amd import hip  # noqa: E402


def timed(fn, reps=3):
    fn()
    torch.cuda.synchronize()
    start, stop = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    start.record()
    for _ in range(reps):
        fn()
    stop.record()
    torch.cuda.synchronize()
    return start.elapsed_time(stop) / reps


def main():
    T, B = (int(v) for v in sys.argv[1:3]) if len(sys.argv) >= 3 else (500, 16)
    C = 29
    rng = np.random.default_rng(0)
    for name, scale, blank_bias in (('untrained (flat)', 0.3, 0.0), ('trained-like (peaked)', 3.0, 4.0)):
        logits = (rng.normal(size=(T, B, C)) * scale).astype(np.float32)
        logits[:, :, -1] += blank_bias
        lg = torch.as_tensor(logits).cuda()
        sl = torch.full((B,), T, dtype=torch.int32, device='cuda')
        print('{}: T={} B={}'.format(name, T, B))
        print('  greedy           {:9.3f} ms'.format(timed(lambda: hip.ctc_greedy_decode(lg, sl))))
        for width in (1, 16, 64, 256, 1024):
            ms = timed(lambda: hip.ctc_beam_decode(lg, sl, width), reps=2)
            print('  beam width {:5d} {:9.3f} ms  ({:.1f} us per frame)'.format(
                width, ms, ms * 1e3 / T))


if __name__ == '__main__':
    main()
