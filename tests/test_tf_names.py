"""Round trip of the (unverified) TensorFlow variable-name mapping and its numerical neutrality:
the re-imported parameters give the same logits in the CPU oracle."""

import numpy as np

from ctc_asr_amd.model import ModelConfig, init_params, to_oracle_layout
from ctc_asr_amd.tf_names import from_tf_variables, to_tf_variables
from oracle import nn as onn


def test_round_trip_is_numerically_neutral():
    rng = np.random.default_rng(0)
    for kwargs in (dict(used_model='ds2', conv_filters=(4, 4, 6), rnn_cell='lstm', cudnn=True),
                   dict(used_model='ds2', conv_filters=(4, 4), rnn_cell='gru', cudnn=True),
                   dict(used_model='ds2', conv_filters=(4, 4), rnn_cell='rnn_relu', cudnn=True),
                   dict(used_model='ds1', rnn_cell='rnn_tanh', cudnn=False)):
        cfg = ModelConfig(num_units_dense=12, num_layers_rnn=2, num_units_rnn=8, **kwargs)
        flat = init_params(cfg, 1)
        for name in flat:
            flat[name] = (flat[name] + rng.normal(size=flat[name].shape) * 0.1).astype(np.float32)
        variables = to_tf_variables(flat, cfg)
        assert 'logits/dense/kernel' in variables and 'dense4/dense/bias' in variables
        if cfg.rnn_cell == 'gru':
            # canonical CudnnCompatibleGRUCell form: gates + candidate input / hidden projections
            scope = ('rnn/cudnn_gru/stack_bidirectional_rnn/cell_0/bidirectional_rnn/fw/'
                     'cudnn_compatible_gru_cell/')
            assert variables[scope + 'gates/kernel'].shape == (cfg.rnn_input_size() + 8, 16)
            assert variables[scope + 'candidate/hidden_projection/kernel'].shape == (8, 8)
            assert variables[scope + 'candidate/hidden_projection/bias'].shape == (8,)
        if cfg.rnn_cell == 'rnn_relu':
            assert any(k.startswith('rnn/cudnn_rnn_relu/') for k in variables)
        if cfg.used_model == 'ds2' and cfg.rnn_cell == 'lstm':
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


def test_name_check_lists_missing_and_unexpected_variables():
    import pytest
    from ctc_asr_amd.tf_names import check_names, expected_names
    cfg = ModelConfig(num_units_dense=12, num_layers_rnn=1, num_units_rnn=8, used_model='ds2',
                      conv_filters=(4, 4), rnn_cell='lstm', cudnn=True)
    names = expected_names(cfg)
    check_names(names + ['global_step', 'beta1_power', 'dense4/dense/kernel/Adam'], cfg)
    with pytest.raises(ValueError) as err:
        check_names([n for n in names if not n.startswith('logits/')] + ['foo/bar'], cfg)
    assert 'logits/dense/kernel' in str(err.value) and 'foo/bar' in str(err.value)
