"""Best-effort mapping between this package's parameter dict and the variable names / layouts a
TensorFlow-1.12 Estimator checkpoint of the reference would hold.

UNVERIFIED: TensorFlow cannot run in the build container, so the names below are recalled from
TensorFlow's variable-scoping rules for the scopes the reference opens (``conv``, ``rnn``,
``dense``, ``dense4``, ``logits``: ``asr/model.py:156,166,219,231``,
``asr/util/tf_contrib.py:50``) and from the canonical (``CudnnLSTMSaveable``) checkpoint form of
``tf.contrib.cudnn_rnn``: per layer and direction a ``cudnn_compatible_lstm_cell`` with
``kernel [I + H, 4H]`` in TensorFlow gate order (i, c, f, o) and ONE ``bias [4H]`` = b_W + b_R.
The cuDNN split of the bias into two vectors is not recoverable from such a checkpoint; on import
the whole bias goes to ``b_ih`` and ``b_hh`` is zero (numerically identical).

Use: ``numpy.savez(path, **to_tf_variables(model.arena.export(), cfg))`` and
``model.arena.load(from_tf_variables(dict(numpy.load(path)), cfg))``.
"""

import numpy as np

# our LSTM gate order (cuDNN / torch): i, f, g, o; TensorFlow LSTM cells: i, c(=g), f, o
_TF_GATE_ORDER = (0, 2, 1, 3)


def _suffix(index):
    return '' if index == 0 else '_{}'.format(index)


def _rnn_scope(cfg, layer, direction):
    side = 'fw' if direction == 0 else 'bw'
    if cfg.cudnn:
        cell = {'lstm': 'cudnn_compatible_lstm_cell', 'gru': 'cudnn_compatible_gru_cell',
                'rnn_relu': 'basic_rnn_cell', 'rnn_tanh': 'basic_rnn_cell'}[cfg.rnn_cell]
        return ('rnn/cudnn_{}/stack_bidirectional_rnn/cell_{}/bidirectional_rnn/{}/{}'
                .format(cfg.rnn_cell if cfg.rnn_cell in ('lstm', 'gru') else 'rnn', layer, side,
                        cell))
    return 'rnn/stack_bidirectional_rnn/cell_{}/bidirectional_rnn/{}/basic_rnn_cell'.format(
        layer, side)


def to_tf_variables(params, cfg):
    """name->array (shared layout) -> {tensorflow variable name: array}."""
    out = {}
    if cfg.used_model == 'ds2':
        for i in range(len(cfg.conv_filters)):
            out['conv/conv2d{}/kernel'.format(_suffix(i))] = params['conv{}/kernel'.format(i)]
            out['conv/conv2d{}/bias'.format(_suffix(i))] = params['conv{}/bias'.format(i)]
    else:
        for i in range(3):
            out['dense/dense{}/kernel'.format(_suffix(i))] = params['dense{}/kernel'.format(i)]
            out['dense/dense{}/bias'.format(_suffix(i))] = params['dense{}/bias'.format(i)]
    hidden = cfg.num_units_rnn
    for layer in range(cfg.num_layers_rnn):
        for direction in (0, 1):
            w_ih = params['rnn{}/w_ih'.format(layer)][direction]      # [G*H, I]
            w_hh = params['rnn{}/w_hh'.format(layer)][direction]      # [G*H, H]
            bias = params['rnn{}/b_ih'.format(layer)][direction] + \
                params['rnn{}/b_hh'.format(layer)][direction]
            if cfg.cell == 'lstm':
                order = _TF_GATE_ORDER
                w_ih = np.concatenate([w_ih[g * hidden:(g + 1) * hidden] for g in order])
                w_hh = np.concatenate([w_hh[g * hidden:(g + 1) * hidden] for g in order])
                bias = np.concatenate([bias[g * hidden:(g + 1) * hidden] for g in order])
            scope = _rnn_scope(cfg, layer, direction)
            out[scope + '/kernel'] = np.concatenate([w_ih.T, w_hh.T], axis=0)   # [I + H, G*H]
            out[scope + '/bias'] = bias
    out['dense4/dense/kernel'] = params['dense4/kernel']
    out['dense4/dense/bias'] = params['dense4/bias']
    out['logits/dense/kernel'] = params['logits/kernel']
    out['logits/dense/bias'] = params['logits/bias']
    return {k: np.asarray(v, dtype=np.float32) for k, v in out.items()}


def from_tf_variables(variables, cfg):
    """{tensorflow variable name: array} -> name->array dict in the shared layout."""
    params = {}
    if cfg.used_model == 'ds2':
        for i in range(len(cfg.conv_filters)):
            params['conv{}/kernel'.format(i)] = variables['conv/conv2d{}/kernel'.format(_suffix(i))]
            params['conv{}/bias'.format(i)] = variables['conv/conv2d{}/bias'.format(_suffix(i))]
    else:
        for i in range(3):
            params['dense{}/kernel'.format(i)] = variables['dense/dense{}/kernel'.format(_suffix(i))]
            params['dense{}/bias'.format(i)] = variables['dense/dense{}/bias'.format(_suffix(i))]
    hidden = cfg.num_units_rnn
    inverse = np.argsort(_TF_GATE_ORDER)
    for layer in range(cfg.num_layers_rnn):
        w_ih, w_hh, b_ih = [], [], []
        for direction in (0, 1):
            scope = _rnn_scope(cfg, layer, direction)
            kernel = np.asarray(variables[scope + '/kernel'])
            bias = np.asarray(variables[scope + '/bias'])
            in_size = kernel.shape[0] - hidden
            wi, wh = kernel[:in_size].T, kernel[in_size:].T
            if cfg.cell == 'lstm':
                wi = np.concatenate([wi[g * hidden:(g + 1) * hidden] for g in inverse])
                wh = np.concatenate([wh[g * hidden:(g + 1) * hidden] for g in inverse])
                bias = np.concatenate([bias[g * hidden:(g + 1) * hidden] for g in inverse])
            w_ih.append(wi)
            w_hh.append(wh)
            b_ih.append(bias)
        params['rnn{}/w_ih'.format(layer)] = np.stack(w_ih)
        params['rnn{}/w_hh'.format(layer)] = np.stack(w_hh)
        params['rnn{}/b_ih'.format(layer)] = np.stack(b_ih)
        params['rnn{}/b_hh'.format(layer)] = np.zeros_like(params['rnn{}/b_ih'.format(layer)])
    params['dense4/kernel'] = variables['dense4/dense/kernel']
    params['dense4/bias'] = variables['dense4/dense/bias']
    params['logits/kernel'] = variables['logits/dense/kernel']
    params['logits/bias'] = variables['logits/dense/bias']
    return {k: np.asarray(v, dtype=np.float32) for k, v in params.items()}
