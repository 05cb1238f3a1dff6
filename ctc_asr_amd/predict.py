"""Transcribe one WAV file: ``python -m ctc_asr_amd.predict --input file.wav``
(counterpart of ``asr/predict.py:44-67``; returns / prints {'decoded', 'plaintext'})."""

import os
import sys

import numpy as np
import torch

from ctc_asr_amd import storage
from ctc_asr_amd.input_functions import features_from_pcm, read_wav
from ctc_asr_amd.model import CTCModel, ModelConfig
from ctc_asr_amd.params import FLAGS


def predict(model, wav_path):
    feats, lengths = features_from_pcm([read_wav(wav_path)], model.device)
    logits, seq_len = model.inference_fn(feats, lengths, training=False)
    model.check_rnn_error()
    decoded, plaintext, _ = model.decode_fn(logits, seq_len, None)
    return {'decoded': np.array(decoded[0], dtype=np.int32), 'plaintext': plaintext[0]}


def main(argv=None):
    FLAGS.parse(sys.argv[1:] if argv is None else argv)
    if not os.path.isfile(FLAGS.input):
        raise ValueError('The input file "{}" does not exist.'.format(FLAGS.input))
    if not torch.cuda.is_available():
        raise SystemExit('ctc_asr_amd.predict needs an MI355X; no GPU is visible.')
    model = CTCModel(ModelConfig.from_flags(FLAGS), 'cuda', seed=FLAGS.random_seed or 1)
    latest = storage.latest_checkpoint(FLAGS.train_dir)
    if latest is None:
        raise SystemExit('No checkpoint found in {}.'.format(FLAGS.train_dir))
    storage.restore_checkpoint(latest, model)
    print('Inputs: {}'.format(FLAGS.input))
    print(predict(model, FLAGS.input))
    return 0


if __name__ == '__main__':
    sys.exit(main())
