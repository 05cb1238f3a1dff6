#!/usr/bin/env python
"""Convert between this package's checkpoints (``<train_dir>/model-<step>.pt``) and TensorFlow
tensor-bundle checkpoints (``model.ckpt-<step>.index`` / ``.data-00000-of-00001``) as the
reference's ``tf.estimator`` writes them.  Runs on the host (no GPU needed for the conversion;
the shared library is only used for its CRC-32C).  The network is described by the usual flags.

    python tools/tf_checkpoint.py to-tf   --train_dir DIR [--used_model ds2 --rnn_cell lstm ...]
    python tools/tf_checkpoint.py from-tf --train_dir DIR [--checkpoint PREFIX] [flags ...]
    python tools/tf_checkpoint.py list PREFIX
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import torch

from ctc_asr_amd import storage, tf_bundle
from ctc_asr_amd.model import ModelConfig, ParamArena
from ctc_asr_amd.params import FLAGS


def main(argv):
    if len(argv) < 2 or argv[0] not in ('to-tf', 'from-tf', 'list'):
        print(__doc__)
        return 2
    command, rest = argv[0], argv[1:]
    if command == 'list':
        for name, (dtype, shape) in sorted(tf_bundle.list_bundle(rest[0]).items()):
            print('{:<100s} {} {}'.format(name, dtype, list(shape)))
        return 0
    checkpoint = None
    if '--checkpoint' in rest:
        at = rest.index('--checkpoint')
        checkpoint = rest[at + 1]
        rest = rest[:at] + rest[at + 2:]
    FLAGS.parse(rest)
    cfg = ModelConfig.from_flags(FLAGS)
    arena = ParamArena(cfg, 'cpu')
    if command == 'to-tf':
        latest = storage.latest_checkpoint(FLAGS.train_dir)
        if latest is None:
            raise SystemExit('no model-*.pt in {}'.format(FLAGS.train_dir))
        state = torch.load(latest, map_location='cpu', weights_only=False)
        if state['shapes'] != arena.shapes:
            raise SystemExit('{} was written for a different network layout (check the flags)'
                             .format(latest))
        arena.param.copy_(state['param'])
        arena.touch()
        print(storage.export_tf_checkpoint(FLAGS.train_dir, arena, cfg, int(state['step'])))
        return 0
    step = storage.import_tf_checkpoint(checkpoint or FLAGS.train_dir, arena, cfg)
    os.makedirs(FLAGS.train_dir, exist_ok=True)
    path = os.path.join(FLAGS.train_dir, 'model-{}.pt'.format(step))
    torch.save({'step': step, 'epoch': 0, 'param': arena.param, 'm': arena.m, 'v': arena.v,
                'dropout_seed': 0, 'shapes': arena.shapes, 'offsets': arena.offsets,
                'extra': {'imported_from': checkpoint or FLAGS.train_dir}}, path)
    print(path)
    return 0


if __name__ == '__main__':
    sys.exit(main(sys.argv[1:]))
