"""Best-effort mapping between this package's parameter dict and the variable names / layouts a
TensorFlow-1.12 Estimator checkpoint of the reference would hold.

UNVERIFIED: TensorFlow cannot run in the build container, so the names below are recalled from
TensorFlow's variable-scoping rules for the scopes the reference opens (``conv``, ``rnn``,
``dense``, ``dense4``, ``logits``: ``asr/model.py:156,166,219,231``,
``asr/util/tf_contrib.py:50``) and from the canonical (``CudnnLSTMSaveable``) checkpoint form of
``tf.contrib.cudnn_rnn``: per layer and direction a ``cudnn_compatible_lstm_cell`` with
``kernel [I + H, 4H]`` in TensorFlow gate order (i, c, f, o) and ONE ``bias [4H]`` = b_W + b_R.
The cuDNN split of the bias into two vectors is not recoverable from such a checkpoint; on import
the whole bias goes to ``b_ih`` and ``b_hh`` is zero (numerically identical).  The GRU keeps the
canonical form of ``CudnnCompatibleGRUCell``: ``gates/{kernel [I+H, 2H], bias [2H]}`` (reset,
update), ``candidate/input_projection/{kernel [I, H], bias [H]}`` and
``candidate/hidden_projection/{kernel [H, H], bias [H]}`` - the candidate's recurrent bias sits
inside r * (R_n h + b_Rn), so both of its bias vectors are real and are kept apart.

`check_names` compares a checkpoint's variable list with what this mapping expects and raises a
ValueError naming the missing and the unexpected variables instead of a bare KeyError.

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
        # layer scope = snake-cased class name: CudnnLSTM, CudnnGRU, CudnnRNNRelu, CudnnRNNTanh
        layer_scope = {'lstm': 'cudnn_lstm', 'gru': 'cudnn_gru', 'rnn_relu': 'cudnn_rnn_relu',
                       'rnn_tanh': 'cudnn_rnn_tanh'}[cfg.rnn_cell]
        return ('rnn/{}/stack_bidirectional_rnn/cell_{}/bidirectional_rnn/{}/{}'
                .format(layer_scope, layer, side, cell))
    return 'rnn/stack_bidirectional_rnn/cell_{}/bidirectional_rnn/{}/basic_rnn_cell'.format(
        layer, side)


def _front_and_head(params_or_vars, cfg, to_tf):
    """The non-recurrent layers map one to one (same shapes, HWIO conv kernels)."""
    pairs = []
    if cfg.used_model == 'ds2':
        for i in range(len(cfg.conv_filters)):
            for leaf in ('kernel', 'bias'):
                pairs.append(('conv{}/{}'.format(i, leaf),
                              'conv/conv2d{}/{}'.format(_suffix(i), leaf)))
    else:
        for i in range(3):
            for leaf in ('kernel', 'bias'):
                pairs.append(('dense{}/{}'.format(i, leaf),
                              'dense/dense{}/{}'.format(_suffix(i), leaf)))
    for ours, theirs in (('dense4', 'dense4/dense'), ('logits', 'logits/dense')):
        for leaf in ('kernel', 'bias'):
            pairs.append(('{}/{}'.format(ours, leaf), '{}/{}'.format(theirs, leaf)))
    if to_tf:
        return {theirs: params_or_vars[ours] for ours, theirs in pairs}
    return {ours: params_or_vars[theirs] for ours, theirs in pairs}


def to_tf_variables(params, cfg):
    """name->array (shared layout) -> {tensorflow variable name: array}."""
    out = _front_and_head(params, cfg, True)
    hidden = cfg.num_units_rnn
    for layer in range(cfg.num_layers_rnn):
        for direction in (0, 1):
            w_ih = params['rnn{}/w_ih'.format(layer)][direction]      # [G*H, I]
            w_hh = params['rnn{}/w_hh'.format(layer)][direction]      # [G*H, H]
            b_ih = params['rnn{}/b_ih'.format(layer)][direction]
            b_hh = params['rnn{}/b_hh'.format(layer)][direction]
            scope = _rnn_scope(cfg, layer, direction)
            if cfg.cell == 'gru':       # cuDNN / torch gate order r, z, n
                out[scope + '/gates/kernel'] = np.concatenate(
                    [w_ih[:2 * hidden].T, w_hh[:2 * hidden].T], axis=0)          # [I + H, 2H]
                out[scope + '/gates/bias'] = b_ih[:2 * hidden] + b_hh[:2 * hidden]
                out[scope + '/candidate/input_projection/kernel'] = w_ih[2 * hidden:].T
                out[scope + '/candidate/input_projection/bias'] = b_ih[2 * hidden:]
                out[scope + '/candidate/hidden_projection/kernel'] = w_hh[2 * hidden:].T
                out[scope + '/candidate/hidden_projection/bias'] = b_hh[2 * hidden:]
                continue
            bias = b_ih + b_hh
            if cfg.cell == 'lstm':
                order = _TF_GATE_ORDER
                w_ih = np.concatenate([w_ih[g * hidden:(g + 1) * hidden] for g in order])
                w_hh = np.concatenate([w_hh[g * hidden:(g + 1) * hidden] for g in order])
                bias = np.concatenate([bias[g * hidden:(g + 1) * hidden] for g in order])
            out[scope + '/kernel'] = np.concatenate([w_ih.T, w_hh.T], axis=0)   # [I + H, G*H]
            out[scope + '/bias'] = bias
    return {k: np.asarray(v, dtype=np.float32) for k, v in out.items()}


def expected_names(cfg):
    """The model-variable names a reference checkpoint of layout ``cfg`` is expected to hold."""
    from ctc_asr_amd.model import param_spec
    zeros = {name: np.zeros(shape, dtype=np.float32) for name, shape in param_spec(cfg)}
    return sorted(to_tf_variables(zeros, cfg))


def check_names(present, cfg):
    """Raise a ValueError listing what is missing / unexpected when the model variables in
    ``present`` (a checkpoint's variable names; optimizer slots and ``global_step`` are ignored)
    are not exactly the ones this mapping expects for ``cfg``."""
    model_vars = {n for n in present if '/Adam' not in n and not n.startswith('beta') and
                  n != 'global_step' and not n.endswith('_power')}
    want = set(expected_names(cfg))
    missing, extra = sorted(want - model_vars), sorted(model_vars - want)
    if missing or extra:
        raise ValueError(
            'TensorFlow checkpoint does not match the (unverified) variable-name mapping of '
            'ctc_asr_amd.tf_names for this network layout.\n  missing: {}\n  unexpected: {}'
            .format(missing or '-', extra or '-'))


def from_tf_variables(variables, cfg):
    """{tensorflow variable name: array} -> name->array dict in the shared layout."""
    check_names(variables.keys(), cfg)
    params = _front_and_head(variables, cfg, False)
    hidden = cfg.num_units_rnn
    inverse = np.argsort(_TF_GATE_ORDER)
    for layer in range(cfg.num_layers_rnn):
        w_ih, w_hh, b_ih, b_hh = [], [], [], []
        for direction in (0, 1):
            scope = _rnn_scope(cfg, layer, direction)
            if cfg.cell == 'gru':
                gates = np.asarray(variables[scope + '/gates/kernel'])
                in_size = gates.shape[0] - hidden
                cand_in = np.asarray(variables[scope + '/candidate/input_projection/kernel'])
                cand_h = np.asarray(variables[scope + '/candidate/hidden_projection/kernel'])
                w_ih.append(np.concatenate([gates[:in_size].T, cand_in.T]))
                w_hh.append(np.concatenate([gates[in_size:].T, cand_h.T]))
                b_ih.append(np.concatenate(
                    [np.asarray(variables[scope + '/gates/bias']),
                     np.asarray(variables[scope + '/candidate/input_projection/bias'])]))
                b_hh.append(np.concatenate(
                    [np.zeros(2 * hidden, dtype=np.float32),
                     np.asarray(variables[scope + '/candidate/hidden_projection/bias'])]))
                continue
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
            b_hh.append(np.zeros_like(bias))      # the sum is all TensorFlow keeps
        params['rnn{}/w_ih'.format(layer)] = np.stack(w_ih)
        params['rnn{}/w_hh'.format(layer)] = np.stack(w_hh)
        params['rnn{}/b_ih'.format(layer)] = np.stack(b_ih)
        params['rnn{}/b_hh'.format(layer)] = np.stack(b_hh)
    return {k: np.asarray(v, dtype=np.float32) for k, v in params.items()}
