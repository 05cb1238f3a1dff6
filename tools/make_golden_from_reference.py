#!/usr/bin/env python
"""Generates tests/golden/reference_python.json by IMPORTING the reference (read-only, from
/root/reference) in the build container and recording inputs + outputs of the pure-Python pieces
of the hot path that run without TensorFlow / python_speech_features:

  asr/labels.py (ctoi, itoc, num_classes), asr/params.py (flag names/defaults, constants,
  get_parameters), asr/util/metrics.py (levenshtein, wer, wer_batch),
  asr/util/csv_helper.py (get_bucket_boundaries), asr/input_functions.py
  (__feature_normalization).

`tensorflow` and `python_speech_features` are not installed here; the modules above only need
`tf.flags.DEFINE_*` / `tf.flags.FLAGS` / `tf.float32` at import time, so an in-memory stand-in
that records flag definitions is registered under those names (it computes nothing).  The
arithmetic that lives inside TensorFlow / psf is NOT pinned by this file.
The reference never travels: only this script and the JSON data it writes are committed.
"""

import json
import os
import sys
import tempfile
import types

import numpy as np

REFERENCE = '/root/reference'
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests', 'golden',
                   'reference_python.json')


def install_stubs():
    defined = {}

    class _Flags:
        def __getattr__(self, name):
            return defined[name]

    tf = types.ModuleType('tensorflow')
    flags = types.SimpleNamespace()
    for kind in ('string', 'integer', 'float', 'boolean', 'multi_integer'):
        setattr(flags, 'DEFINE_' + kind,
                lambda name, default, help_text, _k=kind: defined.__setitem__(name, default))
    flags.FLAGS = _Flags()
    tf.flags = flags
    tf.float32 = 'float32'
    sys.modules['tensorflow'] = tf
    sys.modules['python_speech_features'] = types.ModuleType('python_speech_features')
    return defined


def main():
    defined = install_stubs()
    sys.path.insert(0, REFERENCE)
    if not hasattr(np, 'object'):
        pass   # dense_to_text uses the removed np.object alias; it is not exercised here
    from asr import labels, params
    from asr.util import csv_helper, metrics
    from asr import input_functions

    rng = np.random.default_rng(20260928)
    gold = {'generator': 'tools/make_golden_from_reference.py', 'reference': 'mdangschat/ctc-asr'}

    gold['labels'] = {
        'num_classes': labels.num_classes(),
        'ctoi': {c: labels.ctoi(c) for c in ' abcdefghijklmnopqrstuvwxyz'},
        'itoc': {str(i): labels.itoc(i) for i in range(0, 28)},
    }
    flag_defaults = {}
    for name, default in defined.items():
        if name in ('train_dir', 'corpus_dir', 'train_csv', 'test_csv', 'dev_csv'):
            continue   # absolute paths of the build container
        flag_defaults[name] = default
    gold['flags'] = flag_defaults
    gold['constants'] = {k: getattr(params, k) for k in (
        'MIN_EXAMPLE_LENGTH', 'MAX_EXAMPLE_LENGTH', 'WIN_LENGTH', 'WIN_STEP', 'NUM_FEATURES',
        'CSV_HEADER_PATH', 'CSV_HEADER_LABEL', 'CSV_HEADER_LENGTH', 'CSV_FIELDNAMES',
        'CSV_DELIMITER')}
    gold['get_parameters'] = params.get_parameters()

    words = ['the', 'cat', 'sat', 'on', 'a', 'mat', 'dog', 'ran', 'far', 'away', 'speech', 'ctc']
    lev, wer_cases = [], []
    for _ in range(40):
        a = ''.join(rng.choice(list('abcde '), size=rng.integers(0, 12)))
        b = ''.join(rng.choice(list('abcde '), size=rng.integers(0, 12)))
        lev.append({'a': a, 'b': b, 'distance': int(metrics.levenshtein(a, b))})
    for _ in range(25):
        orig = ' '.join(rng.choice(words, size=rng.integers(1, 8)))
        hyp = ' '.join(rng.choice(words, size=rng.integers(0, 8)))
        wer_cases.append({'original': orig, 'result': hyp, 'wer': float(metrics.wer(orig, hyp))})
    originals = [c['original'] for c in wer_cases[:6]]
    results = [c['result'] for c in wer_cases[:6]]
    rates, mean = metrics.wer_batch(originals, results)
    gold['levenshtein'] = lev
    gold['wer'] = wer_cases
    gold['wer_batch'] = {'originals': originals, 'results': results,
                         'rates': [float(r) for r in rates], 'mean': float(mean)}

    buckets = []
    with tempfile.TemporaryDirectory() as tmp:
        for case, (count, num_buckets, sort) in enumerate([(20, 4, True), (100, 16, True),
                                                           (57, 8, False), (300, 96, True)]):
            lengths = np.round(rng.uniform(0.7, 17.0, size=count), 4)
            if sort:
                lengths = np.sort(lengths)
            path = os.path.join(tmp, 'c{}.csv'.format(case))
            with open(path, 'w') as handle:
                handle.write('path;label;length\n')
                for i, seconds in enumerate(lengths):
                    handle.write('x/{}.wav;abc;{}\n'.format(i, seconds))
            buckets.append({'lengths': [float(v) for v in lengths], 'num_buckets': num_buckets,
                            'boundaries': csv_helper.get_bucket_boundaries(path, num_buckets)})
    gold['bucket_boundaries'] = buckets

    norm_fn = input_functions.__dict__['__feature_normalization']
    norms = []
    for shape in [(7, 4), (13, 80)]:
        x = rng.normal(size=shape).astype(np.float32) * 3 + 1
        case = {'input': x.tolist()}
        for method in ('none', 'local', 'local_scalar'):
            case[method] = np.asarray(norm_fn(x, method), dtype=np.float32).tolist()
        norms.append(case)
    gold['feature_normalization'] = norms

    with open(OUT, 'w') as handle:
        json.dump(gold, handle, indent=1, sort_keys=True)
    print('wrote', OUT, os.path.getsize(OUT), 'bytes')


if __name__ == '__main__':
    main()
