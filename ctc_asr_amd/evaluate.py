"""Evaluation driver: ``python -m ctc_asr_amd.evaluate [--dev] [--flag=value ...]``.

Counterpart of ``asr/evaluate.py:18-43``: one pass over dev.csv (``--dev``) or test.csv with the
model in evaluation mode; reports the CTC loss and the two ``eval_metric_ops`` of
``asr/model.py:111-118`` — mean edit distance and word error rate, each the unweighted mean over
batches of the per-batch mean.  Decoding is the CTC beam search of width ``FLAGS.beam_width``.
"""

import sys

import numpy as np
import torch

from ctc_asr_amd import storage, summaries
from ctc_asr_amd.input_functions import input_fn_generator
from ctc_asr_amd.model import CTCModel, ModelConfig
from ctc_asr_amd.params import FLAGS


def evaluate_dataset(model, target, rank=0, world=1, max_batches=None, report_samples=True):
    """{'loss', 'mean_edit_distance', 'word_error_rate', 'batches'} for one pass over
    ``target`` ('dev' or 'test'); with ``world > 1`` every rank scores its shard and the
    per-batch means are averaged over ranks."""
    input_fn = input_fn_generator(target, device=model.device, rank=rank, world_size=world,
                                  seed=(FLAGS.random_seed or 1) if world > 1 else None)
    losses, meds, wers = [], [], []
    # Decoding is deferred: the beam search runs one workgroup per utterance, so the logits of
    # several batches are decoded in ONE launch (`CTCModel.decode_many`) - same results, and a
    # group costs about what a single batch does.  The metrics stay per-batch means.
    pending, group, samples = [], None, None

    def score_pending():
        nonlocal samples
        results = model.decode_many([(logits, seq_len, originals)
                                     for logits, seq_len, originals, _, _ in pending])
        for (decoded, plaintext, summary), (_, _, _, labels, texts) in zip(results, pending):
            _, mean_ed, _, wer = model.error_rates_fn(labels, texts, decoded, plaintext)
            meds.append(float(mean_ed))
            wers.append(float(wer))
            if samples is None:
                samples = summary
        del pending[:]

    for index, batch in enumerate(input_fn()):
        if max_batches is not None and index >= max_batches:
            break
        features, labels = batch
        logits, seq_len = model.inference_fn(features['spectrogram'],
                                             features['spectrogram_length'], training=False)
        loss = model.loss_fn(logits, seq_len, labels)
        model.check_rnn_error()       # a timed-out persistent recurrence would score garbage
        losses.append(float(loss))
        originals = np.array([t.encode('utf-8') for t in features['label_plaintext']],
                             dtype=object)
        pending.append((logits, seq_len, originals, labels, features['label_plaintext']))
        # (the longest utterances bound the prefix-tree pool: size the group on them)
        group = model.decode_group_size(max(int(item[0].shape[0]) for item in pending),
                                        int(logits.shape[1]))
        if len(pending) >= group:
            score_pending()
    if pending:
        score_pending()
    if report_samples and rank == 0 and samples is not None:
        for dec, orig in list(zip(samples[0], samples[1]))[:FLAGS.num_samples_to_report]:
            print('  decoded: "{}"\n  original: "{}"'.format(dec, orig))
    stats = torch.tensor([np.sum(losses), np.sum(meds), np.sum(wers), len(losses)],
                         dtype=torch.float64, device=model.device)
    if world > 1:
        torch.distributed.all_reduce(stats)
    count = max(float(stats[3]), 1.0)
    return {'loss': float(stats[0]) / count, 'mean_edit_distance': float(stats[1]) / count,
            'word_error_rate': float(stats[2]) / count, 'batches': int(stats[3])}


def main(argv=None):
    FLAGS.parse(sys.argv[1:] if argv is None else argv)
    if not torch.cuda.is_available():
        raise SystemExit('ctc_asr_amd.evaluate needs an MI355X; no GPU is visible.')
    model = CTCModel(ModelConfig.from_flags(FLAGS), 'cuda', seed=FLAGS.random_seed or 1)
    latest = storage.latest_checkpoint(FLAGS.train_dir)
    if latest is None:
        raise SystemExit('No checkpoint found in {}.'.format(FLAGS.train_dir))
    storage.restore_checkpoint(latest, model)
    target = 'dev' if FLAGS.dev else 'test'
    print('Evaluating checkpoint {} on the {} set.'.format(latest, target))
    result = evaluate_dataset(model, target)
    writer = summaries.SummaryWriter(FLAGS.train_dir, 'eval_' + target)
    for tag in ('loss', 'mean_edit_distance', 'word_error_rate'):       # eval_metric_ops + loss
        writer.scalar(tag, result[tag], model.step_count)
    print('Evaluation result: {}'.format(result))
    return 0


if __name__ == '__main__':
    sys.exit(main())
