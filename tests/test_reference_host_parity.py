"""Host-side mirrors against fixtures recorded from the reference's own Python
(tests/golden/reference_python.json, made by tools/make_golden_from_reference.py)."""

import json
import os

import numpy as np
import pytest

from ctc_asr_amd import csv_helper, labels, metrics, params
from oracle import features as ofeat

GOLD = json.load(open(os.path.join(os.path.dirname(__file__), 'golden', 'reference_python.json')))


def test_alphabet():
    assert labels.num_classes() == GOLD['labels']['num_classes'] == 29
    for char, idx in GOLD['labels']['ctoi'].items():
        assert labels.ctoi(char) == idx
    for idx, char in GOLD['labels']['itoc'].items():
        assert labels.itoc(int(idx)) == char
    assert labels.BLANK_ID == 28 and labels.itoc(0) == ''
    with pytest.raises(ValueError):
        labels.ctoi('A')
    with pytest.raises(ValueError):
        labels.ctoi('ab')
    with pytest.raises(ValueError):
        labels.itoc(29)
    with pytest.raises(KeyError):
        labels.itoc(28)          # the blank has no character, like the reference


def test_flag_names_and_defaults():
    params.FLAGS.reset()
    ours = params.FLAGS.defaults_dict()
    for name, default in GOLD['flags'].items():
        assert name in ours, name
        assert ours[name] == default, (name, ours[name], default)
    for name in ('train_dir', 'corpus_dir', 'train_csv', 'test_csv', 'dev_csv', 'dev', 'input'):
        assert name in ours


def test_constants_and_summary_string():
    for name, value in GOLD['constants'].items():
        assert getattr(params, name) == value, name
    params.FLAGS.reset()
    assert params.get_parameters() == GOLD['get_parameters']


def test_levenshtein_and_wer():
    for case in GOLD['levenshtein']:
        assert metrics.levenshtein(case['a'], case['b']) == case['distance']
    for case in GOLD['wer']:
        assert float(metrics.wer(case['original'], case['result'])) == pytest.approx(case['wer'])
    batch = GOLD['wer_batch']
    rates, mean = metrics.wer_batch(batch['originals'], batch['results'])
    assert rates.dtype == np.float32 and mean.dtype == np.float32
    assert np.allclose(rates, batch['rates']) and float(mean) == pytest.approx(batch['mean'])
    with pytest.raises(ZeroDivisionError):
        metrics.wer('', 'a')
    # values quoted in SURVEY.md 8a-a19
    assert metrics.levenshtein('kitten', 'sitting') == 3
    assert float(metrics.wer('the cat sat', 'the cat sat on')) == pytest.approx(1 / 3)


def test_bucket_boundaries(tmp_path):
    for i, case in enumerate(GOLD['bucket_boundaries']):
        path = tmp_path / 'c{}.csv'.format(i)
        with open(path, 'w') as handle:
            handle.write('path;label;length\n')
            for j, seconds in enumerate(case['lengths']):
                handle.write('x/{}.wav;abc;{}\n'.format(j, seconds))
        assert csv_helper.get_bucket_boundaries(str(path), case['num_buckets']) == \
            case['boundaries']


def test_feature_normalization_matches_reference():
    for case in GOLD['feature_normalization']:
        x = np.array(case['input'], dtype=np.float32)
        for method in ('none', 'local', 'local_scalar'):
            got = ofeat.normalize(x, method)
            assert np.allclose(got, np.array(case[method], dtype=np.float32), atol=1e-6)
    with pytest.raises(ValueError):
        ofeat.normalize(np.zeros((2, 2)), 'global')


def test_dense_to_text_and_edit_distance():
    decoded = np.array([[2, 3, 1, 0, 0], [27, 0, 0, 0, 0]], dtype=np.int32)
    strings, summary = metrics.dense_to_text(decoded, np.array([b'ab ', b'z'], dtype=object))
    assert list(strings) == ['ab ', 'z'] and summary.shape == (2, 2)
    assert list(summary[1]) == ['ab ', 'z']
    _, summary = metrics.dense_to_text(decoded, np.array([], dtype=np.int32))
    assert list(summary[1]) == ['n/a', 'n/a']
    # tf.edit_distance(normalize=True) conventions
    assert metrics.edit_distance([1, 2, 3], [1, 3]) == pytest.approx(0.5)
    assert metrics.edit_distance([], []) == 0.0
    assert metrics.edit_distance([1], []) == float('inf')
    dists, mean = metrics.edit_distance_batch([[1, 2], [3]], [[1, 2], [4, 5]])
    assert np.allclose(dists, [0.0, 1.0]) and float(mean) == pytest.approx(0.5)
