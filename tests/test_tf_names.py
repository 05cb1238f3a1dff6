"""Round trip of the (unverified) TensorFlow variable-name mapping and its numerical neutrality:
the re-imported parameters give the same logits in the CPU oracle."""

import numpy as np

from ctc_asr_amd.model import ModelConfig, init_params, to_oracle_layout
from ctc_asr_amd.tf_names import from_tf_variables, to_tf_variables
from oracle import nn as onn


def test_round_trip_is_numerically_neutral():
    rng = np.random.default_rng(0)
    for kwargs in (dict(used_model='ds2', conv_filters=(4, 4, 6), rnn_cell='lstm', cudnn=True),
                   dict(used_model='ds1', rnn_cell='rnn_tanh', cudnn=False)):
        cfg = ModelConfig(num_units_dense=12, num_layers_rnn=2, num_units_rnn=8, **kwargs)
        flat = init_params(cfg, 1)
        for name in flat:
            flat[name] = (flat[name] + rng.normal(size=flat[name].shape) * 0.1).astype(np.float32)
        variables = to_tf_variables(flat, cfg)
        assert 'logits/dense/kernel' in variables and 'dense4/dense/bias' in variables
        if cfg.used_model == 'ds2':
            assert variables['conv/conv2d_1/kernel'].shape == (11, 21, 4, 4)
            key = ('rnn/cudnn_lstm/stack_bidirectional_rnn/cell_1/bidirectional_rnn/bw/'
                   'cudnn_compatible_lstm_cell/kernel')
            assert variables[key].shape == (2 * 8 + 8, 4 * 8)
        back = from_tf_variables(variables, cfg)
        feats = rng.normal(size=(2, 21, 80))
        lengths = np.array([21, 17])
        a, _ = onn.inference(feats, lengths, to_oracle_layout(flat, cfg), cfg.used_model,
                             cfg.rnn_cell, cfg.cudnn)
        b, _ = onn.inference(feats, lengths, to_oracle_layout(back, cfg), cfg.used_model,
                             cfg.rnn_cell, cfg.cudnn)
        assert np.abs(a - b).max() < 1e-5
