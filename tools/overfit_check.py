#!/usr/bin/env python
"""End-to-end learning check: the DS2 model (own convolutions, persistent BiLSTM-1024 kernels,
CTC, Adam) memorises a small batch of noise "utterances" with random transcripts - the loss
falls towards zero and the greedy / beam decodes become the transcripts.
    python tools/overfit_check.py [steps batch seconds]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ctc_asr_amd.engine import Trainer  # noqa: E402
from ctc_asr_amd.labels import decode  # noqa: E402
from ctc_asr_amd.model import CTCModel, ModelConfig  # noqa: E402
from ctc_asr_amd.synth import synthetic_batch  # noqa: E402


class Flags:
    learning_rate, adam_beta1, adam_beta2, adam_epsilon = 3e-4, 0.9, 0.999, 1e-8


def run(steps=300, batch=8, seconds=2.0, verbose=True):
    cfg = ModelConfig(used_model='ds2', conv_filters=(32, 32), num_units_dense=512,
                      num_layers_rnn=2, num_units_rnn=1024, rnn_cell='lstm', cudnn=True,
                      dense_dropout_rate=0.0)
    trainer = Trainer(cfg, flags=Flags, device='cuda', seed=3)
    feats, lengths, labels, texts = synthetic_batch(batch, seconds, seed=5, chars_per_second=6.0)
    feats_d = torch.tensor(feats, device='cuda')
    len_d = torch.tensor(lengths, device='cuda')
    packed = CTCModel.pack_labels(labels, trainer.model.device)
    t0 = time.perf_counter()
    losses = []
    for step in range(steps):
        loss = trainer.train_step(feats_d, len_d, packed, check=(step % 50 == 0))
        if step % 25 == 0 or step == steps - 1:
            losses.append(float(loss))
            if verbose:
                print('step {:4d} loss {:.4f}'.format(step, losses[-1]))
    trainer.model.check_rnn_error()
    logits, seq_len = trainer.model.inference_fn(feats_d, len_d, training=False)
    greedy, _, _ = trainer.model.decode_fn(logits, seq_len, None, greedy=True)
    beam, _, _ = trainer.model.decode_fn(logits, seq_len, None, beam_width=64)
    hits_g = sum(decode(g) == t for g, t in zip(greedy, texts))
    hits_b = sum(decode(b) == t for b, t in zip(beam, texts))
    if verbose:
        print('{:.1f} s; greedy {}/{} exact, beam-64 {}/{} exact; e.g. "{}" vs "{}"'.format(
            time.perf_counter() - t0, hits_g, batch, hits_b, batch, decode(beam[0]), texts[0]))
    return losses, hits_g, hits_b, batch


if __name__ == '__main__':
    args = [float(v) for v in sys.argv[1:4]]
    run(*(int(args[0]) if len(args) > 0 else 300, int(args[1]) if len(args) > 1 else 8,
          args[2] if len(args) > 2 else 2.0))
