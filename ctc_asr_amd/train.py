"""Training driver: ``python -m ctc_asr_amd.train [--flag=value ...]`` on one GPU, or under
``python -m torch.distributed.run --nproc-per-node N -m ctc_asr_amd.train ...`` for data-parallel
training over the GPUs of a node (RCCL all-reduce).

Epoch schedule of the reference (``asr/train.py:19-74``): epoch 1 on 'train_batch' (CSV order,
i.e. SortaGrad for a length-sorted manifest) -> dev evaluation -> epochs 2..max_epochs on
'train_bucket' (shuffled, bucketed) each followed by a dev evaluation.  Checkpoints, resume and
``--delete`` behave like the estimator's (see ``storage.py``).  A NaN/inf loss stops training
(the reference's NanTensorHook, ``asr/model.py:368``).
"""

import math
import os
import sys
import time

import torch

from ctc_asr_amd import storage, tf_bundle
from ctc_asr_amd.engine import Trainer, init_distributed
from ctc_asr_amd.evaluate import evaluate_dataset
from ctc_asr_amd.input_functions import input_fn_generator
from ctc_asr_amd.model import CTCModel, ModelConfig
from ctc_asr_amd.params import FLAGS, get_parameters


class NanLossDuringTrainingError(RuntimeError):
    """Raised when the training loss is NaN or infinite."""


def train_epoch(trainer, target, epoch, rank, world):
    model = trainer.model
    input_fn = input_fn_generator(target, device=model.device, rank=rank, world_size=world,
                                  seed=(FLAGS.random_seed or 1) * 1000 + epoch if world > 1
                                  else None)
    window_loss, window_audio, window_start = 0.0, 0.0, time.perf_counter()
    steps = 0
    for batch in input_fn():
        features, labels = batch
        loss = trainer.train_step(features['spectrogram'], features['spectrogram_length'],
                                  labels)
        steps += 1
        window_audio += batch.audio_seconds * world
        if model.step_count % FLAGS.log_frequency == 0 or steps == 1:
            value = float(trainer.global_mean(loss))
            if not math.isfinite(value):
                raise NanLossDuringTrainingError('NaN loss during training.')
            elapsed = time.perf_counter() - window_start
            if rank == 0:
                print('epoch {} step {:,d}: loss = {:.4f} ({:.1f} audio-s/s)'.format(
                    epoch, model.step_count, value, window_audio / max(elapsed, 1e-9)))
            window_loss, window_audio, window_start = value, 0.0, time.perf_counter()
    return steps, window_loss


def main(argv=None):
    FLAGS.parse(sys.argv[1:] if argv is None else argv)
    rank, local_rank, world = init_distributed()
    if not torch.cuda.is_available():
        raise SystemExit('ctc_asr_amd.train needs an MI355X; no GPU is visible.')
    torch.cuda.set_device(local_rank)
    seed = FLAGS.random_seed if FLAGS.random_seed != 0 else int(time.time())
    if rank == 0:
        storage.maybe_delete_checkpoints(FLAGS.train_dir, FLAGS.delete)
        print('torch {} on {} GPU(s); parameters:{}'.format(torch.__version__, world,
                                                             get_parameters()))
    if world > 1:
        torch.distributed.barrier()
    cfg = ModelConfig.from_flags(FLAGS)
    # corpora have utterances of every length: convolutions as fixed-shape tiles, autotuned once
    trainer = Trainer(cfg, flags=FLAGS, device='cuda:{}'.format(local_rank), seed=seed,
                      world_size=world, rank=rank,
                      conv_mode=os.environ.get('CTCASR_CONV_MODE', 'tiled'))
    model = trainer.model
    start_epoch = 1
    latest = storage.latest_checkpoint(FLAGS.train_dir)
    if latest is not None:
        start_epoch = storage.restore_checkpoint(latest, model) + 1
        if rank == 0:
            print('Restored {} (step {:,d}); continuing with epoch {}.'.format(
                latest, model.step_count, start_epoch))
    elif tf_bundle.latest_checkpoint(FLAGS.train_dir) is not None:
        # a train_dir written by the reference (tf.estimator).  Taking its variables over is
        # opt-in: the variable-name mapping (tf_names.py) has never seen a file TensorFlow wrote.
        if os.environ.get('CTCASR_IMPORT_TF_CHECKPOINT') == '1':
            model.step_count = storage.import_tf_checkpoint(FLAGS.train_dir, model.arena, cfg)
            if rank == 0:
                print('Imported TensorFlow checkpoint {} (global_step {:,d}).'.format(
                    tf_bundle.latest_checkpoint(FLAGS.train_dir), model.step_count))
        elif rank == 0:
            print('Note: {} holds a TensorFlow checkpoint; set CTCASR_IMPORT_TF_CHECKPOINT=1 to '
                  'start from its variables.  Starting from a fresh initialisation.'
                  .format(FLAGS.train_dir))

    for epoch in range(start_epoch, FLAGS.max_epochs + 1):
        target = 'train_batch' if epoch == 1 else 'train_bucket'
        if rank == 0:
            print('Starting epoch {} on {}.'.format(epoch, target))
        train_epoch(trainer, target, epoch, rank, world)
        if rank == 0:
            storage.save_checkpoint(FLAGS.train_dir, model, epoch)
            if os.environ.get('CTCASR_EXPORT_TF_CHECKPOINT') == '1':
                storage.export_tf_checkpoint(FLAGS.train_dir, model.arena, cfg, model.step_count)
        result = evaluate_dataset(model, 'dev', rank, world)
        if rank == 0:
            print('Evaluation result after epoch {}: {}'.format(epoch, result))
    if rank == 0:
        print('Completed all epochs.')
    return 0


if __name__ == '__main__':
    sys.exit(main())
