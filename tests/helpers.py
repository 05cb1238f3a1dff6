"""Shared builders for tests: random parameter dicts in the shared layout, label packing."""

import numpy as np

from oracle import nn as onn


def make_params(rng, used_model='ds2', cell='lstm', hidden=64, dense=32, layers=2,
                conv_filters=(4, 4, 6), classes=29, num_features=80, scale=1.0):
    params = {}
    if used_model == 'ds2':
        c_in, conv = 1, []
        for filt, ksz in zip(conv_filters, onn.DEFAULT_KERNEL_SIZES):
            fan = ksz[0] * ksz[1] * c_in
            conv.append(((rng.normal(size=(ksz[0], ksz[1], c_in, filt)) / np.sqrt(fan) * scale)
                         .astype(np.float32), (rng.normal(size=filt) * 0.1).astype(np.float32)))
            c_in = filt
        params['conv'] = conv
        freq = num_features
        for _ in conv_filters:
            freq = -(-freq // 2)
        in_size = freq * conv_filters[-1]
    else:
        dims = [num_features, dense, dense, dense]
        params['dense'] = [((rng.normal(size=(dims[i], dims[i + 1])) / np.sqrt(dims[i]))
                            .astype(np.float32),
                            (rng.normal(size=dims[i + 1]) * 0.1).astype(np.float32))
                           for i in range(3)]
        in_size = dense
    gates = onn.GATES[cell]
    rnn = []
    for _ in range(layers):
        rnn.append(dict(
            w_ih=(rng.normal(size=(2, gates * hidden, in_size)) / np.sqrt(in_size))
            .astype(np.float32),
            w_hh=(rng.normal(size=(2, gates * hidden, hidden)) / np.sqrt(hidden))
            .astype(np.float32),
            b_ih=(rng.normal(size=(2, gates * hidden)) * 0.1).astype(np.float32),
            b_hh=(rng.normal(size=(2, gates * hidden)) * 0.1).astype(np.float32)))
        in_size = 2 * hidden
    params['rnn'] = rnn
    params['dense4'] = ((rng.normal(size=(2 * hidden, dense)) / np.sqrt(2 * hidden))
                        .astype(np.float32), (rng.normal(size=dense) * 0.1).astype(np.float32))
    params['logits'] = ((rng.normal(size=(dense, classes)) / np.sqrt(dense)).astype(np.float32),
                        (rng.normal(size=classes) * 0.1).astype(np.float32))
    return params


def pack_labels(labels):
    offsets = np.zeros(len(labels) + 1, dtype=np.int32)
    offsets[1:] = np.cumsum([len(row) for row in labels])
    flat = np.array([v for row in labels for v in row] or [0], dtype=np.int32)
    return flat, offsets
