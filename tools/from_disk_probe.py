#!/usr/bin/env python
"""Where train.py-from-disk loses against the same steps from HBM: times the input pipeline alone
(reader thread, staging, uploads, features), the epoch without logged steps and with them.

    python tools/from_disk_probe.py [utterances]
"""
import os
import sys
import tempfile
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ctc_asr_amd import input_functions as inp, summaries, synth, train  # noqa: E402
from ctc_asr_amd.engine import Trainer  # noqa: E402
from ctc_asr_amd.model import ModelConfig  # noqa: E402
from ctc_asr_amd.params import FLAGS  # noqa: E402


def main():
    utterances = int(sys.argv[1]) if len(sys.argv) > 1 else 400
    with tempfile.TemporaryDirectory() as tmp:
        corpus, csv = os.path.join(tmp, 'corpus'), os.path.join(tmp, 'train.csv')
        rng = np.random.default_rng(77)
        synth.write_corpus(corpus, csv, synth.librispeech_like_durations(rng, utterances, drop=True),
                           seed=78, subdir='train')
        FLAGS.reset()
        FLAGS.update(corpus_dir=corpus, train_csv=csv, train_dir=os.path.join(tmp, 'ckpt'),
                     batch_size=16, num_buckets=8, beam_width=64, log_frequency=20, random_seed=5)
        for rep in range(2):
            t0 = time.perf_counter()
            first = None
            n = 0
            for _ in inp.host_batches(csv, True, inp.get_bucket_boundaries(csv, 8), seed=11):
                if first is None:
                    first = time.perf_counter() - t0
                n += 1
            print('host side only (CSV, probes, shuffle, WAV reads): {} batches in {:.3f} s, first '
                  'after {:.3f} s'.format(n, time.perf_counter() - t0, first))
        for rep in range(2):
            t0 = time.perf_counter()
            n = 0
            for _ in inp.input_fn_generator('train_bucket', seed=11)():
                n += 1
            torch.cuda.synchronize()
            print('input_fn alone (+ staging, uploads, features): {} batches in {:.3f} s'.format(
                n, time.perf_counter() - t0))
        cfg = ModelConfig(used_model='ds2', conv_filters=(32, 32), num_units_dense=2048,
                          num_layers_rnn=2, num_units_rnn=1024, rnn_cell='lstm', cudnn=True,
                          beam_width=64)
        trainer = Trainer(cfg, device='cuda', seed=0)
        writer = summaries.SummaryWriter(FLAGS.train_dir, 'train')
        for label, log_frequency, w in (('warm-up', 10000, None), ('no logged steps but the first', 10000, None),
                                        ('log every 20, no writer', 20, None),
                                        ('log every 20 + decode + summaries', 20, writer)):
            FLAGS.update(log_frequency=log_frequency)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            steps, _ = train.train_epoch(trainer, 'train_bucket', 2, 0, 1, w, seed=11)
            torch.cuda.synchronize()
            print('{}: {} steps in {:.3f} s = {:.2f} ms per step'.format(
                label, steps, time.perf_counter() - t0, (time.perf_counter() - t0) / steps * 1e3))


if __name__ == '__main__':
    main()
