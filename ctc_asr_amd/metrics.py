"""Scoring helpers: text rendering, Levenshtein distance, WER and label edit distance.

Behavioural mirror of ``asr/util/metrics.py:9-141`` (``dense_to_text``, ``wer``, ``wer_batch``,
``levenshtein``) and of the ``tf.edit_distance(decoded, labels)`` call in
``asr/model.py:338`` (normalised Levenshtein over integer labels).  Host-side Python like the
reference's ``tf.py_func`` bodies; nothing here is on the GPU hot path.
"""

import numpy as np

from ctc_asr_amd.labels import itoc
from ctc_asr_amd.params import NP_FLOAT


def levenshtein(a, b):
    """Edit distance between two sequences (strings or lists of words / ints).

    Single-row dynamic programme; unit costs for insert / delete / substitute, as in
    ``asr/util/metrics.py:110-141``.
    """
    if len(a) < len(b):
        a, b = b, a
    # `b` is now the shorter sequence; one row of len(b)+1 cells.
    row = list(range(len(b) + 1))
    for i, item_a in enumerate(a, start=1):
        diagonal, row[0] = row[0], i
        for j, item_b in enumerate(b, start=1):
            substitute = diagonal + (item_a != item_b)
            diagonal = row[j]
            row[j] = min(substitute, row[j] + 1, row[j - 1] + 1)
    return row[len(b)]


def wer(original, result):
    """Word error rate = word-level Levenshtein / number of words in ``original``
    (``asr/util/metrics.py:52-76``).  Raises ``ZeroDivisionError`` for an empty original, like the
    reference."""
    if isinstance(original, bytes):
        original = original.decode('utf-8')
    if isinstance(result, bytes):
        result = result.decode('utf-8')
    original_words = original.split()
    result_words = result.split()
    return np.array(levenshtein(original_words, result_words) / float(len(original_words)),
                    dtype=NP_FLOAT)


def wer_batch(originals, results):
    """Per-sample WER ``f32[B]`` and their mean ``f32[]`` (``asr/util/metrics.py:81-105``)."""
    if len(originals) != len(results):
        raise AssertionError('wer_batch(): originals and results differ in length.')
    rates = np.array([wer(o, r) for o, r in zip(originals, results)], dtype=NP_FLOAT)
    mean = np.array(float(np.sum(rates.astype(np.float64))) / float(len(originals)),
                    dtype=NP_FLOAT)
    return rates, mean


def dense_to_text(decoded, originals):
    """Render dense integer rows as strings (0 -> '') and stack them with the originals.

    Returns ``(decoded_strings object[B], summary object[2, B])`` like
    ``asr/util/metrics.py:9-47``; ``originals`` may be empty, giving ``'n/a'`` placeholders.
    (The reference uses the removed ``np.object`` alias; plain ``object`` is the same dtype.)
    """
    decoded_strings = [''.join(itoc(int(i)) for i in row) for row in decoded]
    if len(originals) > 0:
        original_strings = [o.decode('utf-8') if isinstance(o, bytes) else str(o)
                            for o in originals]
    else:
        original_strings = ['n/a'] * len(decoded_strings)
    decoded_arr = np.array(decoded_strings, dtype=object)
    summary = np.vstack([decoded_arr, np.array(original_strings, dtype=object)])
    return decoded_arr, summary


def edit_distance(hypothesis, truth, normalize=True):
    """``tf.edit_distance`` for one pair of integer sequences: Levenshtein(hyp, truth), divided
    by ``len(truth)`` when ``normalize``.  An empty truth gives ``inf`` for a non-empty
    hypothesis and 0 for an empty one (TensorFlow's convention)."""
    hypothesis, truth = list(hypothesis), list(truth)
    dist = float(levenshtein(hypothesis, truth))
    if not normalize:
        return dist
    if len(truth) == 0:
        return float('inf') if dist != 0.0 else 0.0
    return dist / float(len(truth))


def edit_distance_batch(hypotheses, truths, normalize=True):
    """Batch form: ``f32[B]`` of per-utterance distances and their mean (``asr/model.py:338-339``)."""
    if len(hypotheses) != len(truths):
        raise ValueError('edit_distance_batch(): batch sizes differ.')
    dists = np.array([edit_distance(h, t, normalize) for h, t in zip(hypotheses, truths)],
                     dtype=NP_FLOAT)
    return dists, np.array(np.mean(dists.astype(np.float64)) if len(dists) else 0.0,
                           dtype=NP_FLOAT)
