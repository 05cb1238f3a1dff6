"""Scalar / text summaries and the throughput log line of a run.

The reference records ``tf.summary.scalar('mean_edit_distance' | 'word_error_rate', family=
'Metrics')``, ``tf.summary.text('decoded_text', ...)`` and the loss every ``log_frequency`` steps
(``asr/model.py:96-101``, ``asr/train.py:37`` ``save_summary_steps``), reports the evaluation
metrics as ``eval_metric_ops`` (``asr/model.py:111-118``) and prints a throughput line from its
``LoggerHook`` (``asr/util/hooks.py:446-477``).  TensorBoard event files are out of scope; the same
records go to JSON-lines files instead: ``<train_dir>/summaries/<run>.jsonl``, one object per
record - ``{"step", "wall_time", "tag", "value"}`` (scalars) or ``{..., "text": [...]}``.
"""

import json
import os
import time
from datetime import datetime


class SummaryWriter:
    """Appends summary records of one run ('train', 'eval_dev', 'eval_test') to a JSONL file."""

    def __init__(self, train_dir, run):
        self.path = os.path.join(train_dir, 'summaries', run + '.jsonl')
        os.makedirs(os.path.dirname(self.path), exist_ok=True)

    def _append(self, record):
        with open(self.path, 'a', encoding='utf-8') as handle:
            handle.write(json.dumps(record) + '\n')

    def scalar(self, tag, value, step):
        self._append({'step': int(step), 'wall_time': time.time(), 'tag': tag,
                      'value': float(value)})

    def text(self, tag, rows, step):
        self._append({'step': int(step), 'wall_time': time.time(), 'tag': tag,
                      'text': [[str(cell) for cell in row] for row in rows]})


def read_summaries(train_dir, run):
    """All records of a run, in file order."""
    path = os.path.join(train_dir, 'summaries', run + '.jsonl')
    if not os.path.exists(path):
        return []
    with open(path, encoding='utf-8') as handle:
        return [json.loads(line) for line in handle if line.strip()]


class ThroughputLogger:
    """The reference's ``LoggerHook``: every ``log_frequency`` steps one line with the loss,
    examples/sec, sec/batch and batch/sec over the window since the previous line - plus
    audio-seconds/s, the metric of this package's benchmark."""

    def __init__(self, log_frequency, batch_size):
        self.log_frequency, self.batch_size = int(log_frequency), int(batch_size)
        self._start, self._audio, self._steps = time.time(), 0.0, 0

    def add_audio(self, seconds):
        """One training step over ``seconds`` of audio went into the current window."""
        self._audio += float(seconds)
        self._steps += 1

    def restart(self):
        """Start the window now (after work that is not training: summaries, decoding)."""
        self._start = time.time()

    def line(self, global_step, loss_value):
        """Returns (text line, examples/sec, audio-s/s) and starts the next window."""
        now = time.time()
        duration = max(now - self._start, 1e-9)
        # the steps actually taken in this window (the reference's hook assumes log_frequency of
        # them, which overstates the first line of an epoch, logged after ONE step)
        steps = max(self._steps, 1)
        examples_per_sec = steps * self.batch_size / duration
        audio_per_sec = self._audio / duration
        text = ('{:%Y-%m-%d %H:%M:%S}: (step={:,d}); loss={:.4f}; {:.1f} examples/sec '
                '({:.3f} sec/batch) ({:.2f} batch/sec); {:.1f} audio-s/s'.format(
                    datetime.now(), global_step, loss_value, examples_per_sec,
                    duration / float(steps), steps / duration, audio_per_sec))
        self._start, self._audio, self._steps = now, 0.0, 0
        return text, examples_per_sec, audio_per_sec
