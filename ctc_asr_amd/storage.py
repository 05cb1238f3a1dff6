"""Checkpoints in ``FLAGS.train_dir`` with the reference's semantics: implicit resume from the
latest checkpoint, ``--delete`` wipes the directory first, the newest 5 are kept
(``asr/train.py:23,31-42``, ``asr/util/storage.py:88-108``; the reference delegates the format
to ``tf.estimator``, here it is a ``torch.save`` of the flat arenas)."""

import glob
import os
import shutil

import torch

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
    arena.m.copy_(state['m'])
    arena.v.copy_(state['v'])
    model.step_count = int(state['step'])
    model.dropout_seed = int(state['dropout_seed'])
    return int(state['epoch'])
