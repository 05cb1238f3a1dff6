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

import numpy as np
import torch

from ctc_asr_amd import storage, summaries, tf_bundle
from ctc_asr_amd.engine import NanLossDuringTrainingError, Trainer, init_distributed
from ctc_asr_amd.evaluate import evaluate_dataset
from ctc_asr_amd.input_functions import input_fn_generator
from ctc_asr_amd.model import ModelConfig
from ctc_asr_amd.params import FLAGS, get_parameters


def train_epoch(trainer, target, epoch, rank, world, writer=None, seed=None):
    """One pass over ``target``.  Every ``FLAGS.log_frequency`` steps (and at the first step)
    rank 0 prints the reference's LoggerHook line and records the summaries the reference
    records: loss, learning rate, and - from a beam-search decode of the batch just trained on -
    mean edit distance, word error rate and ``num_samples_to_report`` decoded / original texts."""
    model = trainer.model
    if seed is None and world > 1:      # (every rank must walk the same shuffled order)
        seed = (FLAGS.random_seed or 1) * 1000 + epoch
    input_fn = input_fn_generator(target, device=model.device, rank=rank, world_size=world,
                                  seed=seed)
    logger = summaries.ThroughputLogger(FLAGS.log_frequency, FLAGS.batch_size * world)
    window_loss, steps = 0.0, 0
    for batch in input_fn():
        features, labels = batch
        # (the labels as the reader thread has uploaded them: no host-to-device copy here)
        loss = trainer.train_step(features['spectrogram'], features['spectrogram_length'],
                                  batch.packed_labels)
        steps += 1
        logger.add_audio(batch.audio_seconds * world)
        if model.step_count % FLAGS.log_frequency == 0 or steps == 1:
            value = float(trainer.global_mean(loss))
            trainer.drain_checks()      # the loss read above synchronised anyway
            if not math.isfinite(value):
                raise NanLossDuringTrainingError('NaN loss during training.')
            window_loss = value
            if rank == 0:
                line, examples_per_sec, audio_per_sec = logger.line(model.step_count, value)
                print('epoch {} '.format(epoch) + line)
                if writer is not None:
                    decoded, plaintext, summary = model.decode_fn(
                        model.last_logits, model.last_seq_length,
                        np.array([t.encode('utf-8') for t in features['label_plaintext']],
                                 dtype=object))
                    _, mean_ed, _, wer = model.error_rates_fn(
                        labels, features['label_plaintext'], decoded, plaintext)
                    step = model.step_count
                    writer.scalar('loss', value, step)
                    writer.scalar('learning_rate', trainer.lr, step)
                    writer.scalar('Metrics/mean_edit_distance', mean_ed, step)
                    writer.scalar('Metrics/word_error_rate', wer, step)
                    writer.scalar('examples_per_sec', examples_per_sec, step)
                    writer.scalar('audio_seconds_per_sec', audio_per_sec, step)
                    writer.text('decoded_text', summary[:, :FLAGS.num_samples_to_report], step)
            else:
                logger.line(model.step_count, value)
            # the decode and the summary records above are not training time
            logger.restart()
    trainer.drain_checks()              # nothing unchecked reaches the checkpoint
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
    trainer = Trainer(cfg, flags=FLAGS, device='cuda:{}'.format(local_rank), seed=seed,
                      world_size=world, rank=rank)
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

    writer = summaries.SummaryWriter(FLAGS.train_dir, 'train') if rank == 0 else None
    eval_writer = summaries.SummaryWriter(FLAGS.train_dir, 'eval_dev') if rank == 0 else None
    for epoch in range(start_epoch, FLAGS.max_epochs + 1):
        target = 'train_batch' if epoch == 1 else 'train_bucket'
        if rank == 0:
            print('Starting epoch {} on {}.'.format(epoch, target))
        train_epoch(trainer, target, epoch, rank, world, writer)
        if rank == 0:
            storage.save_checkpoint(FLAGS.train_dir, model, epoch)
            if os.environ.get('CTCASR_EXPORT_TF_CHECKPOINT') == '1':
                storage.export_tf_checkpoint(FLAGS.train_dir, model.arena, cfg, model.step_count)
        result = evaluate_dataset(model, 'dev', rank, world)
        if rank == 0:
            for tag in ('loss', 'mean_edit_distance', 'word_error_rate'):
                eval_writer.scalar(tag, result[tag], model.step_count)
            print('Evaluation result after epoch {}: {}'.format(epoch, result))
    if rank == 0:
        print('Completed all epochs.')
    return 0


if __name__ == '__main__':
    sys.exit(main())
