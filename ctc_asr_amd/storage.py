"""Checkpoints in ``FLAGS.train_dir`` with the reference's semantics: implicit resume from the
latest checkpoint, ``--delete`` wipes the directory first, the newest 5 are kept
(``asr/train.py:23,31-42``, ``asr/util/storage.py:88-108``; the reference delegates the format
to ``tf.estimator``, here it is a ``torch.save`` of the flat arenas).

`export_tf_checkpoint` / `import_tf_checkpoint` additionally write / read the variables in
TensorFlow's own checkpoint format (``model.ckpt-<step>.index`` + ``.data-00000-of-00001`` +
``checkpoint``; `tf_bundle`, names and layouts from `tf_names`), so that the same weights can be
loaded by the reference and by this package."""

import glob
import os
import shutil

import numpy as np
import torch

from ctc_asr_amd import tf_bundle, tf_names

KEEP_CHECKPOINT_MAX = 5


def maybe_delete_checkpoints(path, delete):
    """Delete ``path`` recursively if ``delete`` (and it exists); resume otherwise."""
    if os.path.exists(path) and os.path.isdir(path):
        if delete:
            print('Deleting old checkpoints: {}'.format(path))
            shutil.rmtree(path)
        else:
            print('Found old checkpoint. Resuming training: {}'.format(path))


def checkpoint_paths(train_dir):
    return sorted(glob.glob(os.path.join(train_dir, 'model-*.pt')),
                  key=lambda p: int(os.path.basename(p)[6:-3]))


def latest_checkpoint(train_dir):
    paths = checkpoint_paths(train_dir)
    return paths[-1] if paths else None


def save_checkpoint(train_dir, model, epoch, extra=None):
    os.makedirs(train_dir, exist_ok=True)
    arena = model.arena
    state = {'step': model.step_count, 'epoch': epoch, 'param': arena.param.cpu(),
             'm': arena.m.cpu(), 'v': arena.v.cpu(), 'dropout_seed': model.dropout_seed,
             'shapes': arena.shapes, 'offsets': arena.offsets, 'extra': extra or {}}
    path = os.path.join(train_dir, 'model-{}.pt'.format(model.step_count))
    tmp = path + '.tmp'
    torch.save(state, tmp)
    os.replace(tmp, path)
    for old in checkpoint_paths(train_dir)[:-KEEP_CHECKPOINT_MAX]:
        os.remove(old)
    return path


def restore_checkpoint(path, model):
    state = torch.load(path, map_location='cpu', weights_only=False)
    arena = model.arena
    if state['shapes'] != arena.shapes:
        raise ValueError('Checkpoint {} was written for a different network layout.'.format(path))
    arena.param.copy_(state['param'])
    arena.touch()
    arena.m.copy_(state['m'])
    arena.v.copy_(state['v'])
    model.step_count = int(state['step'])
    model.dropout_seed = int(state['dropout_seed'])
    return int(state['epoch'])


def export_tf_checkpoint(train_dir, arena, cfg, global_step):
    """Write the parameters of ``arena`` (a `ParamArena`) as a TensorFlow checkpoint
    ``<train_dir>/model.ckpt-<global_step>`` and point ``<train_dir>/checkpoint`` at it.
    Optimizer slots are not written (the reference's cuDNN layers keep theirs as opaque
    blobs); returns the checkpoint prefix."""
    os.makedirs(train_dir, exist_ok=True)
    variables = tf_names.to_tf_variables(arena.export(), cfg)
    variables['global_step'] = np.array(int(global_step), dtype=np.int64)
    name = 'model.ckpt-{}'.format(int(global_step))
    prefix = os.path.join(train_dir, name)
    tf_bundle.write_bundle(prefix, variables)
    tf_bundle.write_checkpoint_state(train_dir, name)
    return prefix


def import_tf_checkpoint(path, arena, cfg):
    """Load the model variables of a TensorFlow checkpoint (a prefix, or a directory holding a
    ``checkpoint`` file) into ``arena``; Adam moments restart at zero.  Returns global_step.
    Optimizer slot variables (``.../Adam``, ``beta1_power`` ...) in the file are ignored."""
    prefix = tf_bundle.latest_checkpoint(path) if os.path.isdir(path) else path
    if prefix is None or not os.path.isfile(prefix + '.index'):
        raise ValueError('No TensorFlow checkpoint at {}'.format(path))
    present = tf_bundle.list_bundle(prefix)
    tf_names.check_names(present, cfg)      # ValueError naming missing / unexpected variables
    wanted = [n for n in present if n == 'global_step' or n in set(tf_names.expected_names(cfg))]
    variables = tf_bundle.read_bundle(prefix, wanted)
    step = variables.pop('global_step', None)
    arena.load(tf_names.from_tf_variables(variables, cfg))
    variables['global_step'] = 0 if step is None else step
    arena.m.zero_()
    arena.v.zero_()
    return int(variables['global_step'])
