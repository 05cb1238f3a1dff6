"""The numpy layer oracle against the independent torch restatement, plus the TensorFlow
conventions it encodes (SAME padding, reshape order, Adam form)."""

import numpy as np
import pytest
import torch

from oracle import nn as onn
from oracle import torch_ref
from tests.helpers import make_params


def test_same_padding_values_of_the_reference_stack():
    # SURVEY.md 8a-a11
    assert onn.same_padding(999, 11, 2) == (500, 5, 5)
    assert onn.same_padding(80, 41, 2) == (40, 19, 20)
    assert onn.same_padding(40, 21, 2) == (20, 9, 10)
    assert onn.same_padding(20, 21, 2) == (10, 9, 10)
    assert onn.same_padding(500, 11, 1) == (500, 5, 5)


@pytest.mark.parametrize('used_model,cell,cudnn,filters', [
    ('ds2', 'lstm', True, (4, 4, 6)), ('ds2', 'lstm', True, (4, 4)), ('ds2', 'gru', True, (4, 4, 6)),
    ('ds2', 'rnn_relu', True, (4, 4, 6)), ('ds1', 'rnn_tanh', True, ()),
    ('ds1', 'lstm', False, ()), ('ds2', 'rnn_tanh', False, (4, 4, 6))])
def test_numpy_oracle_equals_torch_restatement(used_model, cell, cudnn, filters):
    rng = np.random.default_rng(0)
    params = make_params(rng, used_model, cell if cudnn else 'rnn_tanh', hidden=8, dense=12,
                         conv_filters=filters or (4, 4, 6))
    feats = rng.normal(size=(3, 37, 80))
    lengths = np.array([37, 20, 29])
    logits, seq_len = onn.inference(feats, lengths, params, used_model, cell, cudnn)
    model = torch_ref.TorchRefModel(params, used_model, cell, cudnn, dtype=torch.float64)
    t_logits, t_len = model(torch.tensor(feats), lengths)
    assert logits.shape == tuple(t_logits.shape)
    assert np.abs(logits - t_logits.detach().numpy()).max() < 1e-5
    assert (seq_len == t_len.numpy()).all()
    if used_model == 'ds2':
        assert (seq_len == 19).all()          # padded conv length for every row
    else:
        assert (seq_len == lengths).all()


def test_conv_reshape_is_frequency_major_channel_minor():
    rng = np.random.default_rng(2)
    conv = [(rng.normal(size=(11, 41, 1, 3)) * 0.1, np.zeros(3))]
    x = rng.normal(size=(1, 9, 80))
    out, _ = onn.conv_layers(x, conv)
    nhwc = onn.relu_clip(onn.conv2d_same(x[..., None], conv[0][0], conv[0][1], (2, 2)))
    assert out.shape == (1, 5, 40 * 3)
    assert np.array_equal(out[0, 2, 7 * 3 + 1], nhwc[0, 2, 7, 1])


def test_length_aware_rnn_emits_zeros_after_the_end_and_reverses_per_row():
    rng = np.random.default_rng(3)
    hidden, steps = 5, 9
    layer = dict(w_ih=rng.normal(size=(2, hidden, 4)), w_hh=rng.normal(size=(2, hidden, hidden)),
                 b_ih=np.zeros((2, hidden)), b_hh=np.zeros((2, hidden)))
    x = rng.normal(size=(steps, 2, 4))
    y = onn.birnn_layer(x, layer, 'rnn_tanh', seq_len=[9, 4])
    assert np.all(y[4:, 1] == 0.0) and np.any(y[3, 1] != 0.0)
    # backward direction of the short row starts at its own last frame
    first_bw = np.tanh(x[3, 1] @ layer['w_ih'][1].T)
    assert np.allclose(y[3, 1, hidden:], first_bw)
    # cuDNN semantics run through the padding instead
    y_cudnn = onn.birnn_layer(x, layer, 'rnn_tanh', seq_len=None)
    assert np.any(y_cudnn[8, 1] != 0.0)


def test_adam_is_tensorflow_form():
    p, g = np.array([1.0, -2.0]), np.array([0.5, 0.25])
    new_p, m, v = onn.adam_step(p, g, np.zeros(2), np.zeros(2), 1, lr=0.1, eps=1e-8)
    lr_t = 0.1 * np.sqrt(1 - 0.999) / (1 - 0.9)
    assert np.allclose(m, 0.1 * g) and np.allclose(v, 0.001 * g * g)
    assert np.allclose(new_p, p - lr_t * m / (np.sqrt(v) + 1e-8))
    # torch.optim.Adam puts epsilon inside the bias-corrected root: differs for tiny gradients
    tiny = np.array([1e-9])
    tf_p, _, _ = onn.adam_step(np.zeros(1), tiny, np.zeros(1), np.zeros(1), 1, lr=0.1)
    t = torch.zeros(1, dtype=torch.float64, requires_grad=True)
    opt = torch.optim.Adam([t], lr=0.1)
    t.grad = torch.tensor(tiny)
    opt.step()
    assert abs(tf_p[0] - float(t)) > 1e-3
