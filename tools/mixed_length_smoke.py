#!/usr/bin/env python
"""BASELINE.json configs[4]-shaped smoke run on one GPU: LibriSpeech-like utterance lengths
(0.7-17 s), bucketed batches, DS2 + 2xBiLSTM-1024, one training epoch per schedule stage and a
dev evaluation with beam width 64.  Prints throughput and the evaluation metrics."""
import os
import sys
import tempfile
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ctc_asr_amd import synth, train  # noqa: E402


def main():
    count = int(sys.argv[1]) if len(sys.argv) > 1 else 96
    batch = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    epochs = int(sys.argv[3]) if len(sys.argv) > 3 else 2
    log_frequency = int(sys.argv[4]) if len(sys.argv) > 4 else 4
    rng = np.random.default_rng(0)
    with tempfile.TemporaryDirectory() as tmp:
        corpus = os.path.join(tmp, 'corpus')
        for name, n, seed in (('train', count, 1), ('dev', 24, 2), ('test', 24, 3)):
            synth.write_corpus(corpus, os.path.join(tmp, name + '.csv'),
                               synth.librispeech_like_durations(rng, n), seed=seed, subdir=name)
        t0 = time.perf_counter()
        train.main(['--corpus_dir', corpus, '--train_csv', os.path.join(tmp, 'train.csv'),
                    '--dev_csv', os.path.join(tmp, 'dev.csv'),
                    '--test_csv', os.path.join(tmp, 'test.csv'),
                    '--train_dir', os.path.join(tmp, 'ckpt'), '--batch_size={}'.format(batch), '--num_buckets=6',
                    '--feature_type=mel', '--used_model=ds2', '--conv_filters=32',
                    '--conv_filters=32', '--num_layers_rnn=2', '--num_units_rnn=1024',
                    '--rnn_cell=lstm', '--num_units_dense=2048', '--max_epochs={}'.format(epochs),
                    '--beam_width=64', '--log_frequency={}'.format(log_frequency), '--random_seed=5'])
        print('total wall time {:.1f} s'.format(time.perf_counter() - t0))


if __name__ == '__main__':
    main()
