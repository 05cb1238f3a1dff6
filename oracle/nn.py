"""ORACLE (test infrastructure, not product code): numpy restatement of the network layers.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
this module.  PARITY UNPINNED by the reference (no tests; TensorFlow/cuDNN not runnable here) —
the equations are the TensorFlow 1.12 / cuDNN semantics listed in SURVEY.md 8a and are
cross-checked against independent ``torch`` CPU operators in ``tests/test_oracle_nn.py``.

Restates (forward only, float64 unless the caller passes float32):

* ``tf_contrib.dense_layers`` / dense4 / logits — ``asr/util/tf_contrib.py:34-61``,
  ``asr/model.py:219-233``
* ``tf_contrib.conv_layers`` (NHWC, SAME, ReLU, min(.,20), reshape) —
  ``asr/util/tf_contrib.py:64-146``
* cuDNN bidirectional RNN stack (LSTM / GRU / ReLU / tanh, no sequence lengths) —
  ``asr/model.py:186-216``
* ``stack_bidirectional_dynamic_rnn`` over ``BasicRNNCell(tanh)`` with sequence lengths —
  ``asr/model.py:169-184``, ``asr/util/tf_contrib.py:149-194``
* TensorFlow-form Adam — ``asr/model.py:80-83``

Parameter layout (shared with the product code, ``ctc_asr_amd/model.py``):
conv kernel ``[kt, kf, Cin, Cout]`` (TensorFlow HWIO); dense kernel ``[in, out]``;
RNN ``w_ih [2, G*H, I]``, ``w_hh [2, G*H, H]``, ``b_ih, b_hh [2, G*H]`` with cuDNN gate order
(LSTM i,f,g,o; GRU r,z,n) — direction 0 = forward, 1 = backward.
"""

import math

import numpy as np


def sigmoid(x):
    return 1.0 / (1.0 + np.exp(-x))


def relu_clip(x, cutoff=20.0):
    return np.minimum(np.maximum(x, 0.0), cutoff)


def dense(x, kernel, bias):
    return x @ kernel + bias


def same_padding(size, kernel, stride):
    """TensorFlow SAME: out = ceil(in / s); total pad = max((out-1)*s + k - in, 0); the extra
    element goes to the *end*."""
    out = -(-size // stride)
    total = max((out - 1) * stride + kernel - size, 0)
    return out, total // 2, total - total // 2


def conv2d_same(x, kernel, bias, strides):
    """NHWC conv with TensorFlow SAME padding.  x [B, T, F, Cin], kernel [kt, kf, Cin, Cout]."""
    batch, size_t, size_f, c_in = x.shape
    k_t, k_f, _, c_out = kernel.shape
    out_t, pad_t0, pad_t1 = same_padding(size_t, k_t, strides[0])
    out_f, pad_f0, pad_f1 = same_padding(size_f, k_f, strides[1])
    padded = np.pad(x, ((0, 0), (pad_t0, pad_t1), (pad_f0, pad_f1), (0, 0)))
    out = np.zeros((batch, out_t, out_f, c_out), dtype=x.dtype)
    flat_kernel = kernel.reshape(k_t * k_f * c_in, c_out)
    for i in range(out_t):
        rows = padded[:, i * strides[0]: i * strides[0] + k_t]
        for j in range(out_f):
            patch = rows[:, :, j * strides[1]: j * strides[1] + k_f, :]
            out[:, i, j, :] = patch.reshape(batch, -1) @ flat_kernel
    return out + bias


DEFAULT_KERNEL_SIZES = ((11, 41), (11, 21), (11, 21))
DEFAULT_STRIDES = ((2, 2), (1, 2), (1, 2))


def conv_layers(sequences, conv_params, relu_cutoff=20.0):
    """``conv_layers`` with dropout rate 0.  ``sequences`` [B, T, F]; ``conv_params`` list of
    (kernel, bias).  Returns (output [B, T', F'*C] freq-major / channel-minor, seq_len [B] all
    equal to T' — true lengths are discarded like in the reference)."""
    if len(conv_params) > len(DEFAULT_STRIDES):
        raise ValueError('conv_layers(): at most three convolutional layers are defined.')
    out = sequences[..., None]
    for (kernel, bias), stride in zip(conv_params, DEFAULT_STRIDES):
        out = relu_clip(conv2d_same(out, kernel, bias, stride), relu_cutoff)
    batch, out_t = out.shape[0], out.shape[1]
    out = out.reshape(batch, out_t, -1)
    return out, np.full(batch, out_t, dtype=np.int32)


def dense_layers(sequences, dense_params, relu_cutoff=20.0):
    """DS1 front-end with dropout rate 0: 3 x (dense + ReLU + min 20)."""
    out = sequences
    for kernel, bias in dense_params:
        out = relu_clip(dense(out, kernel, bias), relu_cutoff)
    return out


# ------------------------------------------------------------------------------------------
# Recurrent layers, time-major x [T, B, I]
# ------------------------------------------------------------------------------------------
GATES = {'lstm': 4, 'gru': 3, 'rnn_relu': 1, 'rnn_tanh': 1}


def _cell_step(cell, gates_x, h, c, w_hh, b_hh, hidden):
    """One step of one direction.  ``gates_x`` = W x + b_ih [B, G*H]."""
    if cell == 'lstm':
        pre = gates_x + h @ w_hh.T + b_hh
        i = sigmoid(pre[:, 0 * hidden:1 * hidden])
        f = sigmoid(pre[:, 1 * hidden:2 * hidden])
        g = np.tanh(pre[:, 2 * hidden:3 * hidden])
        o = sigmoid(pre[:, 3 * hidden:4 * hidden])
        c = f * c + i * g
        return o * np.tanh(c), c
    if cell == 'gru':
        rec = h @ w_hh.T + b_hh
        r = sigmoid(gates_x[:, :hidden] + rec[:, :hidden])
        z = sigmoid(gates_x[:, hidden:2 * hidden] + rec[:, hidden:2 * hidden])
        n = np.tanh(gates_x[:, 2 * hidden:] + r * rec[:, 2 * hidden:])
        return (1.0 - z) * n + z * h, c
    pre = gates_x + h @ w_hh.T + b_hh
    if cell == 'rnn_relu':
        return np.maximum(pre, 0.0), c
    if cell == 'rnn_tanh':
        return np.tanh(pre), c
    raise ValueError('Unsupported cell "{}"'.format(cell))


def birnn_layer(x, params, cell, seq_len=None):
    """One bidirectional layer.  ``params`` = dict(w_ih, w_hh, b_ih, b_hh).

    ``seq_len is None``  -> cuDNN semantics: every row runs all T steps; the backward direction
    starts at t = T-1, i.e. inside the zero padding.
    ``seq_len`` given    -> ``dynamic_rnn`` semantics: steps ``t >= seq_len[b]`` emit zeros and
    carry the state; the backward direction is reversed per row by its own length.
    Returns y [T, B, 2H] = [fw || bw].
    """
    num_steps, batch, _ = x.shape
    hidden = params['w_hh'].shape[2]
    out = np.zeros((num_steps, batch, 2 * hidden), dtype=x.dtype)
    for direction in (0, 1):
        w_ih, w_hh = params['w_ih'][direction], params['w_hh'][direction]
        b_ih, b_hh = params['b_ih'][direction], params['b_hh'][direction]
        h = np.zeros((batch, hidden), dtype=x.dtype)
        c = np.zeros((batch, hidden), dtype=x.dtype)
        if seq_len is None:
            order = range(num_steps) if direction == 0 else range(num_steps - 1, -1, -1)
            for t in order:
                h, c = _cell_step(cell, x[t] @ w_ih.T + b_ih, h, c, w_hh, b_hh, hidden)
                out[t, :, direction * hidden:(direction + 1) * hidden] = h
        else:
            for step in range(num_steps):
                # row b reads its own time index: forward = step, backward = len_b - 1 - step
                t_idx = np.full(batch, step) if direction == 0 else \
                    np.asarray(seq_len) - 1 - step
                alive = step < np.asarray(seq_len)
                t_safe = np.clip(t_idx, 0, num_steps - 1)
                x_t = x[t_safe, np.arange(batch)]
                h_new, c_new = _cell_step(cell, x_t @ w_ih.T + b_ih, h, c, w_hh, b_hh, hidden)
                h = np.where(alive[:, None], h_new, h)
                c = np.where(alive[:, None], c_new, c)
                for b in range(batch):
                    if alive[b]:
                        out[t_idx[b], b, direction * hidden:(direction + 1) * hidden] = h[b]
    return out


def birnn_stack(x, layers, cell, seq_len=None):
    """Stacked bidirectional layers; each consumes the previous ``[fw || bw]`` (no dropout)."""
    out = x
    for params in layers:
        out = birnn_layer(out, params, cell, seq_len)
    return out


# ------------------------------------------------------------------------------------------
# Whole network (forward): returns time-major logits like ``CTCModel.inference_fn``
# ------------------------------------------------------------------------------------------
def inference(features, feature_len, params, used_model='ds2', rnn_cell='lstm', cudnn=True,
              relu_cutoff=20.0):
    """``features`` [B, T, 80] -> (logits [T', B, 29], seq_len [B]).  Dropout rates 0.

    ``params`` = dict(conv=[(k, b)...] | dense=[(k, b)...], rnn=[layer dicts], dense4=(k, b),
    logits=(k, b)).  With ``cudnn=False`` the RNN is the length-aware tanh ``BasicRNNCell``
    regardless of ``rnn_cell`` (``asr/params.py:47-50`` TODO)."""
    if used_model == 'ds1':
        out = dense_layers(features, params['dense'], relu_cutoff)
        seq_len = np.asarray(feature_len, dtype=np.int32)
    elif used_model == 'ds2':
        out, seq_len = conv_layers(features, params['conv'], relu_cutoff)
    else:
        raise ValueError('Unsupported model "{}" in flags.'.format(used_model))
    x = np.transpose(out, (1, 0, 2))
    if cudnn:
        y = birnn_stack(x, params['rnn'], rnn_cell, None)
    else:
        y = birnn_stack(x, params['rnn'], 'rnn_tanh', seq_len)
    d4 = relu_clip(dense(y, *params['dense4']), relu_cutoff)
    return dense(d4, *params['logits']), seq_len


def adam_step(param, grad, m, v, step, lr=1e-5, beta1=0.9, beta2=0.999, eps=1e-8):
    """TensorFlow-form Adam (epsilon outside the bias correction).  ``step`` counts from 1."""
    lr_t = lr * math.sqrt(1.0 - beta2 ** step) / (1.0 - beta1 ** step)
    m = beta1 * m + (1.0 - beta1) * grad
    v = beta2 * v + (1.0 - beta2) * grad * grad
    return param - lr_t * m / (np.sqrt(v) + eps), m, v
