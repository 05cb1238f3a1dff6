#!/usr/bin/env python
"""Training steps on ONE bucket of the C5 sequence (bench.py --workload c5): the shortest
(`short`), the longest (`long`) or the n-th batch - for kernel traces that show what a short
bucket costs against a long one (profiles/r03_c5_*.md).

    python tools/c5_bucket_steps.py short|long|<index> [steps]
    rocprofv3 --kernel-trace --stats -d /tmp/c5s -o c5 -- python tools/c5_bucket_steps.py short 6
"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from ctc_asr_amd import hip  # noqa: E402
from ctc_asr_amd.engine import Trainer  # noqa: E402
from ctc_asr_amd.labels import encode  # noqa: E402
from ctc_asr_amd.model import CTCModel, ModelConfig  # noqa: E402
from ctc_asr_amd.synth import random_label, random_pcm  # noqa: E402


def main():
    which = sys.argv[1] if len(sys.argv) > 1 else 'short'
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 6
    filters, layers, hidden, dense, batch, _, cell = bench.WORKLOADS['c5']
    sequence = bench.c5_bucket_sequence(batch, 24)
    longest = [int(s.max()) for s in sequence]
    index = {'short': int(np.argmin(longest)), 'long': int(np.argmax(longest))}.get(which)
    if index is None:
        index = int(which)
    nsamp = sequence[index]
    cfg = ModelConfig(used_model='ds2', conv_filters=filters, num_units_dense=dense,
                      num_layers_rnn=layers, num_units_rnn=hidden, rnn_cell=cell, cudnn=True,
                      dense_dropout_rate=0.1)
    trainer = Trainer(cfg, device='cuda:0', seed=0)
    rng = np.random.default_rng(1)
    pcm = torch.zeros((batch, int(nsamp.max())), dtype=torch.int16, device='cuda')
    rows = []
    for b, n in enumerate(nsamp):
        pcm[b, :n] = torch.from_numpy(random_pcm(rng, int(n))).cuda()
        rows.append(list(encode(random_label(rng, max(1, int(n / 16000.0 * 15.0))))))
    nsamp_d = torch.from_numpy(nsamp).cuda()
    labels = CTCModel.pack_labels(rows, trainer.model.device)

    def step():
        feats, lengths = hip.features(pcm, nsamp_d, 'mel', 'local', False, 16000)
        return trainer.train_step(feats, lengths, labels)

    for _ in range(3):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / steps * 1e3
    trainer.drain_checks()
    seconds = float(nsamp.sum()) / 16000.0
    frames = hip.features_num_frames(int(nsamp.max()))
    print('bucket {} of the C5 sequence: {:.2f}-{:.2f} s utterances, T\' = {}, {:.3f} ms per step, '
          '{:.0f} audio-s/s'.format(index, nsamp.min() / 16000.0, nsamp.max() / 16000.0,
                                    (frames + 1) // 2, ms, seconds / (ms * 1e-3)))


if __name__ == '__main__':
    main()
