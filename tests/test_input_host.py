"""Host-side logic of the input pipeline that needs no GPU: manifest quirks, WAV checks,
bucketing / batching / sharding semantics (asr/input_functions.py:21-153)."""

import random

import numpy as np
import pytest
from scipy.io import wavfile

from ctc_asr_amd import input_functions as inp
from ctc_asr_amd import synth
from ctc_asr_amd.params import FLAGS


@pytest.fixture()
def tiny_corpus(tmp_path):
    FLAGS.reset()
    rows = synth.write_corpus(str(tmp_path / 'corpus'), str(tmp_path / 'train.csv'),
                              [0.7, 0.9, 1.1, 1.3, 1.5], seed=3, chars_per_second=5.0)
    FLAGS.update(corpus_dir=str(tmp_path / 'corpus'), train_csv=str(tmp_path / 'train.csv'))
    yield tmp_path, rows
    FLAGS.reset()


def test_manifest_drops_header_and_last_row(tiny_corpus):
    tmp_path, rows = tiny_corpus
    assert len(rows) == 6                       # 5 utterances + the sacrificial duplicate
    manifest = inp.read_manifest(FLAGS.train_csv)
    assert [r['path'] for r in manifest] == [r[0] for r in rows[:5]]
    assert all(set(r['label']) <= set(' abcdefghijklmnopqrstuvwxyz') for r in manifest)
    # without the sacrificial row the reference loses the last real example
    with open(FLAGS.train_csv) as handle:
        lines = handle.read().splitlines()
    with open(FLAGS.train_csv, 'w') as handle:
        handle.write('\n'.join(lines[:-1]) + '\n')
    assert len(inp.read_manifest(FLAGS.train_csv)) == 4


def test_read_wav_checks(tiny_corpus, tmp_path):
    tmp, rows = tiny_corpus
    audio = inp.read_wav(str(tmp / 'corpus' / rows[0][0]))
    assert audio.dtype == np.int16 and len(audio) == 11200
    with pytest.raises(ValueError):
        inp.read_wav(str(tmp / 'corpus' / 'missing.wav'))
    short = tmp_path / 'short.wav'
    wavfile.write(str(short), 16000, np.zeros(400, dtype=np.int16))
    with pytest.raises(RuntimeError):
        inp.read_wav(str(short))
    wrong_rate = tmp_path / 'rate.wav'
    wavfile.write(str(wrong_rate), 8000, np.zeros(4000, dtype=np.int16))
    with pytest.raises(RuntimeError):
        inp.read_wav(str(wrong_rate))
    # the header-only probe agrees with the full read and applies the same checks
    assert inp.probe_wav(str(tmp / 'corpus' / rows[0][0])) == 11200
    with pytest.raises(ValueError):
        inp.probe_wav(str(tmp / 'corpus' / 'missing.wav'))
    with pytest.raises(RuntimeError):
        inp.probe_wav(str(short))
    with pytest.raises(RuntimeError):
        inp.probe_wav(str(wrong_rate))
    stereo = tmp_path / 'stereo.wav'
    wavfile.write(str(stereo), 16000, np.zeros((4000, 2), dtype=np.int16))
    with pytest.raises(RuntimeError):
        inp.probe_wav(str(stereo))
    with pytest.raises(RuntimeError):
        inp.read_wav(str(stereo))
    floats = tmp_path / 'float.wav'            # IEEE-float WAV: `wave` refuses, fall back
    wavfile.write(str(floats), 16000, np.zeros(4000, dtype=np.float32))
    with pytest.raises(RuntimeError):
        inp.probe_wav(str(floats))
    with pytest.raises(ValueError):
        inp._check_feature_args('fbank', None)
    with pytest.raises(ValueError):
        inp._check_feature_args(None, 'global')
    with pytest.raises(ValueError):
        inp.input_fn_generator('training')


def _items(lengths):
    return [(None, [1], 'x{}'.format(i), n) for i, n in enumerate(lengths)]


def test_plain_batches_keep_order_and_drop_the_remainder():
    groups = list(inp._group_batches(iter(_items(range(10))), False, [], 4))
    assert [[it[3] for it in g] for g in groups] == [[0, 1, 2, 3], [4, 5, 6, 7]]


def test_bucketing_matches_tf_bucket_by_sequence_length():
    # boundaries [5, 10]: buckets (-inf, 5), [5, 10), [10, inf); full buckets emit at once,
    # leftovers are emitted at the end in bucket order and are not dropped
    lengths = [1, 12, 6, 2, 7, 13, 3, 9, 4, 5, 11]
    groups = list(inp._group_batches(iter(_items(lengths)), True, [5, 10], 3))
    got = [[it[3] for it in g] for g in groups]
    assert got == [[1, 2, 3], [6, 7, 9], [12, 13, 11], [4], [5]]
    assert sorted(v for g in got for v in g) == sorted(lengths)


def test_shuffle_buffer_is_a_permutation_and_seeded():
    items = list(range(100))
    a = list(inp._shuffle_buffer(iter(items), 16, random.Random(1)))
    b = list(inp._shuffle_buffer(iter(items), 16, random.Random(1)))
    assert sorted(a) == items and a == b and a != items
    # an element can only surface once the buffer has filled: position >= index - buffer
    assert all(pos >= val - 16 for pos, val in enumerate(a))


def test_random_label_shape():
    rng = np.random.default_rng(0)
    for n in (1, 2, 15, 150):
        text = synth.random_label(rng, n)
        assert len(text) == n and not text.startswith(' ') and not text.endswith(' ')
        assert '  ' not in text
    durations = synth.librispeech_like_durations(rng, 1000)
    assert durations.min() >= 0.7 and durations.max() <= 17.0 and 8.0 < durations.mean() < 14.0


def test_data_parallel_ranks_share_each_group(tmp_path):
    """world_size 2: at every step both ranks draw from the SAME group of 2 x batch_size
    utterances (same bucket, so the padded time extent matches and nobody waits in the
    all-reduce), keep disjoint halves, and together see exactly what one process would."""
    FLAGS.reset()
    durations = [0.7 + 0.05 * i for i in range(41)]
    synth.write_corpus(str(tmp_path / 'corpus'), str(tmp_path / 'train.csv'), durations, seed=5,
                       chars_per_second=5.0)
    FLAGS.update(corpus_dir=str(tmp_path / 'corpus'), train_csv=str(tmp_path / 'train.csv'),
                 batch_size=3, num_buckets=4, random_seed=11)
    try:
        from ctc_asr_amd.csv_helper import get_bucket_boundaries
        boundaries = get_bucket_boundaries(FLAGS.train_csv, FLAGS.num_buckets)
        for use_buckets in (True, False):
            bounds = boundaries if use_buckets else []
            ranks = [list(inp.host_batches(FLAGS.train_csv, use_buckets, bounds, rank, 2))
                     for rank in (0, 1)]
            assert len(ranks[0]) == len(ranks[1]) > 0
            FLAGS.update(batch_size=6)          # what ONE process with the global batch forms
            single = list(inp.host_batches(FLAGS.train_csv, use_buckets, bounds, 0, 1,
                                           seed=FLAGS.random_seed))
            FLAGS.update(batch_size=3)
            assert len(single) >= len(ranks[0])
            for step, (a, b) in enumerate(zip(*ranks)):
                assert len(a) == len(b) <= 3
                texts = [item[2] for item in a + b]
                assert len(set(texts)) == len(texts)                 # disjoint halves
                merged = [item[2] for item in single[step]]
                assert texts == merged[:len(texts)]                  # same group, same order
                for (pcm, ids, text), ref in zip(a + b, single[step]):
                    assert np.array_equal(pcm, ref[0]) and ids == ref[1]
                if use_buckets:     # one bucket per group
                    keys = {np.searchsorted(boundaries, inp.num_frames(len(item[0])),
                                            side='right') for item in a + b}
                    assert len(keys) == 1
    finally:
        FLAGS.reset()


def test_pipelined_input_projection_plan_covers_every_row_once():
    """`xw_pipeline_plan` (the schedule of `CTCModel._rnn_fwd_pipelined`): replayed in numpy,
    the per-launch half-K products reproduce y W^T + b for odd and even T' and any chunk count -
    every row is initialised (bias form) exactly once and before anything is accumulated into
    it, and only rows whose y is final after that launch are touched.  (Round-1 bug: odd T' with
    an even chunk count initialised a range twice.)"""
    from ctc_asr_amd.model import xw_pipeline_plan
    rng = np.random.default_rng(0)
    hidden, n_out, batch = 3, 5, 2
    for t_out in (8, 9, 49, 65, 66, 251, 499, 500, 501):
        for chunks in (1, 2, 3, 4, 5, 6):
            y = rng.normal(size=(t_out, batch, 2 * hidden))
            w = rng.normal(size=(n_out, 2 * hidden))
            bias = rng.normal(size=n_out)
            want = (y.reshape(t_out * batch, -1) @ w.T + bias).reshape(t_out, batch, n_out)
            got = np.full((t_out, batch, n_out), np.nan)
            bounds, plan = xw_pipeline_plan(t_out, chunks)
            assert bounds[0] == 0 and bounds[-1] == t_out and len(plan) == chunks
            assert all(b1 >= b0 for b0, b1 in zip(bounds, bounds[1:]))
            for c, ops in enumerate(plan):
                hi = bounds[c + 1]
                for d, a, b, first in ops:
                    # forward direction: times < hi are final; backward direction: times >= T'-hi
                    assert (d == 0 and b <= hi) or (d == 1 and a >= t_out - hi)
                    part = y[a:b, :, d * hidden:(d + 1) * hidden] @ \
                        w[:, d * hidden:(d + 1) * hidden].T
                    if first:
                        assert np.isnan(got[a:b]).all()
                        got[a:b] = part + bias
                    else:
                        assert not np.isnan(got[a:b]).any()
                        got[a:b] += part
            assert np.abs(got - want).max() < 1e-12, (t_out, chunks)
