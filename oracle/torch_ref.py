"""ORACLE (test infrastructure, not product code): the same graph in stock ``torch`` CPU operators.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
this module.  It serves two purposes:

1. an *independent* cross-check of ``oracle/nn.py`` / ``oracle/ctc.py`` (``F.conv2d`` with explicit
   asymmetric SAME padding, ``nn.LSTM`` / ``nn.GRU`` / ``nn.RNN`` — same gate order and double bias
   as cuDNN —, ``F.ctc_loss``), and the source of reference *gradients* through autograd;
2. the CPU baseline of ``bench.py`` (kind "port"): the reference's TensorFlow CPU path cannot be
   run, so the identical graph is timed on oneDNN/MKL-backed torch operators.

Follows ``CTCModel.inference_fn`` / ``loss_fn`` (``asr/model.py:123-269``) and
``tf_contrib.conv_layers`` / ``dense_layers`` (``asr/util/tf_contrib.py:34-146``).
Parameter layout: see ``oracle/nn.py``.
"""

import math

import torch
import torch.nn.functional as F

from oracle.nn import DEFAULT_STRIDES, same_padding

_RNN_CLASSES = {'lstm': torch.nn.LSTM, 'gru': torch.nn.GRU, 'rnn_relu': torch.nn.RNN,
                'rnn_tanh': torch.nn.RNN}


def _as_tensor(value, dtype):
    return torch.as_tensor(value, dtype=dtype).clone().detach()


class TorchRefModel(torch.nn.Module):
    """DS1/DS2 graph over a parameter dict in the shared layout (numpy arrays or tensors)."""

    def __init__(self, params, used_model='ds2', rnn_cell='lstm', cudnn=True, relu_cutoff=20.0,
                 dtype=torch.float32):
        super().__init__()
        self.used_model, self.rnn_cell, self.cudnn = used_model, rnn_cell, cudnn
        self.relu_cutoff = relu_cutoff
        front = params['conv'] if used_model == 'ds2' else params['dense']
        self.front_kernels = torch.nn.ParameterList(
            [torch.nn.Parameter(_as_tensor(k, dtype)) for k, _ in front])
        self.front_biases = torch.nn.ParameterList(
            [torch.nn.Parameter(_as_tensor(b, dtype)) for _, b in front])
        cell = rnn_cell if cudnn else 'rnn_tanh'
        self.cell = cell
        hidden = params['rnn'][0]['w_hh'].shape[2]
        self.hidden = hidden
        self.rnns = torch.nn.ModuleList()
        for layer in params['rnn']:
            in_size = layer['w_ih'].shape[2]
            kwargs = dict(input_size=in_size, hidden_size=hidden, num_layers=1,
                          bidirectional=True)
            if cell in ('rnn_relu', 'rnn_tanh'):
                kwargs['nonlinearity'] = 'relu' if cell == 'rnn_relu' else 'tanh'
            rnn = _RNN_CLASSES[cell](**kwargs).to(dtype)
            with torch.no_grad():
                for direction, suffix in ((0, ''), (1, '_reverse')):
                    getattr(rnn, 'weight_ih_l0' + suffix).copy_(
                        _as_tensor(layer['w_ih'][direction], dtype))
                    getattr(rnn, 'weight_hh_l0' + suffix).copy_(
                        _as_tensor(layer['w_hh'][direction], dtype))
                    getattr(rnn, 'bias_ih_l0' + suffix).copy_(
                        _as_tensor(layer['b_ih'][direction], dtype))
                    getattr(rnn, 'bias_hh_l0' + suffix).copy_(
                        _as_tensor(layer['b_hh'][direction], dtype))
            self.rnns.append(rnn)
        self.dense4_kernel = torch.nn.Parameter(_as_tensor(params['dense4'][0], dtype))
        self.dense4_bias = torch.nn.Parameter(_as_tensor(params['dense4'][1], dtype))
        self.logits_kernel = torch.nn.Parameter(_as_tensor(params['logits'][0], dtype))
        self.logits_bias = torch.nn.Parameter(_as_tensor(params['logits'][1], dtype))

    def _relu_clip(self, x):
        return torch.clamp(x, min=0.0, max=self.relu_cutoff)

    def forward(self, features, feature_len):
        """features [B, T, 80] -> (logits [T', B, C], seq_len LongTensor [B])."""
        batch = features.shape[0]
        if self.used_model == 'ds2':
            out = features.unsqueeze(1)  # NCHW: [B, 1, T, F]
            for kernel, bias, stride in zip(self.front_kernels, self.front_biases,
                                            DEFAULT_STRIDES):
                k_t, k_f = kernel.shape[0], kernel.shape[1]
                _, pt0, pt1 = same_padding(out.shape[2], k_t, stride[0])
                _, pf0, pf1 = same_padding(out.shape[3], k_f, stride[1])
                out = F.pad(out, (pf0, pf1, pt0, pt1))
                out = F.conv2d(out, kernel.permute(3, 2, 0, 1), bias, stride=stride)
                out = self._relu_clip(out)
            # [B, C, T', F'] -> [B, T', F', C] -> [B, T', F'*C]
            out = out.permute(0, 2, 3, 1).reshape(batch, out.shape[2], -1)
            seq_len = torch.full((batch,), out.shape[1], dtype=torch.long)
        else:
            out = features
            for kernel, bias in zip(self.front_kernels, self.front_biases):
                out = self._relu_clip(out @ kernel + bias)
            seq_len = torch.as_tensor(feature_len, dtype=torch.long)
        x = out.transpose(0, 1)  # time-major
        for rnn in self.rnns:
            if self.cudnn:
                x, _ = rnn(x)
            else:
                packed = torch.nn.utils.rnn.pack_padded_sequence(x, seq_len.cpu(),
                                                                 enforce_sorted=False)
                x, _ = torch.nn.utils.rnn.pad_packed_sequence(rnn(packed)[0],
                                                              total_length=x.shape[0])
        d4 = self._relu_clip(x @ self.dense4_kernel + self.dense4_bias)
        return d4 @ self.logits_kernel + self.logits_bias, seq_len

    def loss(self, logits, seq_len, labels):
        """Mean over the batch of -ln p(label | x); ``labels`` = list of B label lists."""
        flat = torch.tensor([v for row in labels for v in row], dtype=torch.long)
        lengths = torch.tensor([len(row) for row in labels], dtype=torch.long)
        per_utt = F.ctc_loss(F.log_softmax(logits, dim=-1), flat, seq_len, lengths,
                             blank=logits.shape[-1] - 1, reduction='none', zero_infinity=False)
        return per_utt.mean(), per_utt

    def grads_in_shared_layout(self):
        """Gradients re-packed into the parameter-dict layout of ``oracle/nn.py``."""
        front = [(k.grad, b.grad) for k, b in zip(self.front_kernels, self.front_biases)]
        rnn = []
        for module in self.rnns:
            layer = {}
            for key, name in (('w_ih', 'weight_ih_l0'), ('w_hh', 'weight_hh_l0'),
                              ('b_ih', 'bias_ih_l0'), ('b_hh', 'bias_hh_l0')):
                layer[key] = torch.stack([getattr(module, name).grad,
                                          getattr(module, name + '_reverse').grad])
            rnn.append(layer)
        return {('conv' if self.used_model == 'ds2' else 'dense'): front, 'rnn': rnn,
                'dense4': (self.dense4_kernel.grad, self.dense4_bias.grad),
                'logits': (self.logits_kernel.grad, self.logits_bias.grad)}


class TFAdam:
    """TensorFlow-form Adam over ``module.parameters()`` (``asr/model.py:80-83``)."""

    def __init__(self, parameters, lr=1e-5, beta1=0.9, beta2=0.999, eps=1e-8):
        self.params = [p for p in parameters]
        self.lr, self.beta1, self.beta2, self.eps = lr, beta1, beta2, eps
        self.m = [torch.zeros_like(p) for p in self.params]
        self.v = [torch.zeros_like(p) for p in self.params]
        self.step_count = 0

    @torch.no_grad()
    def step(self):
        self.step_count += 1
        lr_t = self.lr * math.sqrt(1.0 - self.beta2 ** self.step_count) / \
            (1.0 - self.beta1 ** self.step_count)
        for p, m, v in zip(self.params, self.m, self.v):
            if p.grad is None:
                continue
            m.mul_(self.beta1).add_(p.grad, alpha=1.0 - self.beta1)
            v.mul_(self.beta2).addcmul_(p.grad, p.grad, value=1.0 - self.beta2)
            p.addcdiv_(m, v.sqrt().add_(self.eps), value=-lr_t)

    def zero_grad(self):
        for p in self.params:
            p.grad = None
