"""End-to-end parity of the MI355X model (forward, CTC loss, backward, Adam) against the CPU
oracle: logits and loss within 1e-3 (north_star bar, fp32), gradients to 1e-3 relative."""

import os

import numpy as np
import pytest
import torch

from ctc_asr_amd.model import CTCModel, ModelConfig, init_params, to_oracle_layout
from oracle import ctc as octc
from oracle import nn as onn
from oracle import torch_ref

pytestmark = pytest.mark.gpu

CASES = {
    'ds2_lstm_3conv': dict(used_model='ds2', conv_filters=(4, 4, 6), rnn_cell='lstm', cudnn=True),
    'ds2_lstm_2conv': dict(used_model='ds2', conv_filters=(4, 4), rnn_cell='lstm', cudnn=True),
    # 32 -> 32 channels: the layer with its own data-gradient kernel (ctcasr_conv_s12_bwd_data)
    'ds2_lstm_32ch': dict(used_model='ds2', conv_filters=(32, 32), rnn_cell='lstm', cudnn=True),
    'ds2_lstm_32ch_3conv': dict(used_model='ds2', conv_filters=(32, 32, 8), rnn_cell='lstm',
                                cudnn=True),
    # the reference's own filter counts: layers 2 and 3 both on the own kernels
    'ds2_lstm_ref_convs': dict(used_model='ds2', conv_filters=(32, 32, 96), rnn_cell='lstm',
                               cudnn=True),
    'ds2_gru': dict(used_model='ds2', conv_filters=(4, 4, 6), rnn_cell='gru', cudnn=True),
    'ds2_relu': dict(used_model='ds2', conv_filters=(4, 4, 6), rnn_cell='rnn_relu', cudnn=True),
    'ds1_tanh_cudnn': dict(used_model='ds1', rnn_cell='rnn_tanh', cudnn=True),
    'ds1_basic_rnn': dict(used_model='ds1', rnn_cell='lstm', cudnn=False),   # C1 semantics
}


def _setup(case, seed=0, batch=3, frames=61, hidden=64, dense=32, layers=2):
    cfg = ModelConfig(num_units_dense=dense, num_layers_rnn=layers, num_units_rnn=hidden,
                      dense_dropout_rate=0.0, **CASES[case])
    rng = np.random.default_rng(seed)
    flat = init_params(cfg, seed)
    for name in flat:   # non-zero biases and livelier weights than the tiny-σ initialiser
        flat[name] = (flat[name] + rng.normal(size=flat[name].shape) * 0.05).astype(np.float32)
    feats = rng.normal(size=(batch, frames, 80)).astype(np.float32)
    flen = np.array([frames] + list(rng.integers(frames // 2, frames, size=batch - 1)),
                    dtype=np.int32)
    for b in range(batch):
        feats[b, flen[b]:] = 0.0
    labels = [list(rng.integers(1, 28, size=rng.integers(1, 9))) for _ in range(batch)]
    return cfg, flat, feats, flen, labels


@pytest.mark.parametrize('case', sorted(CASES))
def test_logits_loss_and_gradients(case):
    _check_logits_loss_and_gradients(case)


@pytest.mark.parametrize('chunks', [1, 3])
def test_gradients_through_the_persistent_recurrence(chunks):
    """H = 1024 LSTM layers take the LDS-resident kernels; with ``bwd_chunks`` > 1 the backward
    recurrence is cut into launches and the weight gradients are accumulated range by range on
    the side stream.  Same bars as the small cases."""
    model = _check_logits_loss_and_gradients('ds2_lstm_2conv', bwd_chunks=chunks, hidden=1024,
                                             frames=71, batch=2)
    from ctc_asr_amd import hip
    assert hip.rnn_persistent_supported('lstm', 36, 2, 1024)
    hip.rnn_poll_error('lstm', model._acts['rnn_ws'], 36, 2, 1024)


@pytest.mark.parametrize('case,hidden,batch', [('ds2_gru', 1024, 2), ('ds2_gru', 2048, 3),
                                               ('ds2_lstm_2conv', 2048, 2),
                                               ('ds2_lstm_2conv', 1024, 19),
                                               # 33..64 rows: two persistent launches per pass
                                               # over blocks of 32 and 8 rows
                                               ('ds2_lstm_2conv', 1024, 40)])
def test_gradients_through_the_other_persistent_kernels(case, hidden, batch):
    """Whole-model parity (logits, loss, every gradient) for the shapes that take the round-2
    persistent kernels: GRU at H = 1024 / 2048, the LSTM at H = 2048 (one direction per launch),
    a batch of 19 rows (two 16-row tiles with their own barriers) and one of 40 rows (blocks of
    32 + 8 rows, each with its own barrier words / exchange buffer / carry; backward in step
    ranges with the weight gradients on the side stream)."""
    from ctc_asr_amd import hip
    cell = 'gru' if 'gru' in case else 'lstm'
    # (the fp64 CPU oracle dominates the run time: fewer frames for the H = 2048 models)
    frames = 39 if hidden == 2048 else 63
    assert hip.rnn_persistent_supported(cell, (frames + 1) // 2, batch, hidden)
    model = _check_logits_loss_and_gradients(case, hidden=hidden, frames=frames, batch=batch)
    model.check_rnn_error()


@pytest.mark.parametrize('frames,pipelined', [(131, 4), (129, 4), (97, 2), (141, 3)])
def test_pipelined_forward_equals_single_launch_forward(frames, pipelined):
    """fwd_chunks > 1 (layer 1's forward recurrence on half of the chip in step ranges, layer
    2's input projection accumulated range by range on the side stream) against fwd_chunks = 1
    (whole-chip launch, one GEMM): logits, loss and gradients agree to fp32 rounding, in
    training and in evaluation mode.  frames = 129 / 97 give an ODD T' (65 / 49): with an even
    chunk count the two directions' time ranges then do not coincide (round-1 bug: rows were
    initialised twice, logits off by O(1))."""
    cfg, flat, feats, flen, labels = _setup('ds2_lstm_2conv', hidden=1024, frames=frames, batch=2)
    t_out = (frames + 1) // 2
    out = {}
    for chunks in (1, pipelined):
        model = CTCModel(cfg, 'cuda', params=flat)
        model.fwd_chunks = chunks
        assert model._pipeline_forward(0, 'lstm', t_out, 2, 1024, None, 0.0) == (chunks > 1)
        logits, seq_len = model.inference_fn(torch.tensor(feats), torch.tensor(flen),
                                             training=True)
        loss = model.loss_fn(logits, seq_len, labels)
        model.backward()
        eval_logits, _ = model.inference_fn(torch.tensor(feats), torch.tensor(flen),
                                            training=False)
        model.check_rnn_error()
        out[chunks] = (logits.cpu().numpy(), float(loss), model.arena.export('grad'),
                       eval_logits.cpu().numpy())
    piped = out[pipelined]
    assert np.abs(piped[0] - out[1][0]).max() < 1e-5
    assert abs(piped[1] - out[1][1]) < 1e-4
    assert np.abs(piped[3] - out[1][3]).max() < 1e-5
    for name, ref_g in out[1][2].items():
        err = np.abs(piped[2][name] - ref_g).max()
        assert err < 1e-5 * max(1.0, np.abs(ref_g).max()), (name, err)


def _check_logits_loss_and_gradients(case, bwd_chunks=None, **setup):
    cfg, flat, feats, flen, labels = _setup(case, **setup)
    model = CTCModel(cfg, 'cuda', params=flat)
    if bwd_chunks is not None:
        model.bwd_chunks = bwd_chunks
    logits, seq_len = model.inference_fn(torch.tensor(feats), torch.tensor(flen), training=True)
    loss = model.loss_fn(logits, seq_len, labels)
    model.backward()

    # numpy oracle (forward semantics) ...
    ref_logits, ref_len = onn.inference(feats.astype(np.float64), flen,
                                        to_oracle_layout(flat, cfg), cfg.used_model,
                                        cfg.rnn_cell, cfg.cudnn)
    assert (seq_len.cpu().numpy() == ref_len).all()
    assert np.abs(logits.cpu().numpy() - ref_logits).max() < 1e-3
    ref_loss, _ = octc.ctc_loss(ref_logits, labels, ref_len)
    assert abs(float(loss) - ref_loss.mean()) < 1e-3

    # ... and the torch restatement for gradients
    ref = torch_ref.TorchRefModel(to_oracle_layout(flat, cfg), cfg.used_model, cfg.rnn_cell,
                                  cfg.cudnn, dtype=torch.float64)
    t_logits, t_len = ref(torch.tensor(feats, dtype=torch.float64), flen)
    t_loss, _ = ref.loss(t_logits, t_len, labels)
    t_loss.backward()
    assert abs(float(loss) - float(t_loss)) < 1e-3
    ref_grads = ref.grads_in_shared_layout()
    got = model.arena.export('grad')
    front = 'conv' if cfg.used_model == 'ds2' else 'dense'
    pairs = []
    for i, (gk, gb) in enumerate(ref_grads[front]):
        pairs += [('{}{}/kernel'.format(front, i), gk), ('{}{}/bias'.format(front, i), gb)]
    for i, layer in enumerate(ref_grads['rnn']):
        pairs += [('rnn{}/{}'.format(i, k), layer[k]) for k in ('w_ih', 'w_hh', 'b_ih', 'b_hh')]
    pairs += [('dense4/kernel', ref_grads['dense4'][0]), ('dense4/bias', ref_grads['dense4'][1]),
              ('logits/kernel', ref_grads['logits'][0]), ('logits/bias', ref_grads['logits'][1])]
    for name, ref_g in pairs:
        ref_g = ref_g.numpy()
        err = np.abs(got[name] - ref_g).max()
        assert err < 1e-3 * max(1.0, np.abs(ref_g).max()), (name, err)
    return model


def test_training_steps_track_the_oracle():
    """Three TF-form Adam steps on the same batch: losses follow the torch CPU restatement."""
    cfg, flat, feats, flen, labels = _setup('ds2_lstm_2conv', seed=3)
    model = CTCModel(cfg, 'cuda', params=flat)
    ref = torch_ref.TorchRefModel(to_oracle_layout(flat, cfg), cfg.used_model, cfg.rnn_cell,
                                  cfg.cudnn, dtype=torch.float64)
    opt = torch_ref.TFAdam(ref.parameters(), lr=1e-3)
    for _ in range(3):
        loss = model.forward_backward(torch.tensor(feats), torch.tensor(flen), labels)
        model.apply_gradients(learning_rate=1e-3)
        opt.zero_grad()
        t_logits, t_len = ref(torch.tensor(feats, dtype=torch.float64), flen)
        t_loss, _ = ref.loss(t_logits, t_len, labels)
        t_loss.backward()
        opt.step()
        assert abs(float(loss) - float(t_loss)) < 1e-3
    final = model.arena.export('param')
    assert np.abs(final['logits/kernel'] - ref.logits_kernel.detach().numpy()).max() < 1e-4


def test_dropout_training_path_runs_and_eval_is_deterministic():
    cfg, flat, feats, flen, labels = _setup('ds2_lstm_3conv')
    cfg.dense_dropout_rate = 0.1
    model = CTCModel(cfg, 'cuda', params=flat)
    l1, s1 = model.inference_fn(torch.tensor(feats), torch.tensor(flen), training=False)
    l2, _ = model.inference_fn(torch.tensor(feats), torch.tensor(flen), training=False)
    assert torch.equal(l1, l2)
    l3, _ = model.inference_fn(torch.tensor(feats), torch.tensor(flen), training=True)
    assert not torch.equal(l1, l3)
    loss = model.loss_fn(l3, s1, labels)
    model.backward()
    assert torch.isfinite(loss) and torch.isfinite(model.arena.grad).all()


def test_greedy_decode_strings_match_oracle():
    cfg, flat, feats, flen, labels = _setup('ds2_lstm_3conv', batch=4, frames=120)
    model = CTCModel(cfg, 'cuda', params=flat)
    logits, seq_len = model.inference_fn(torch.tensor(feats), torch.tensor(flen), training=False)
    decoded, plaintext, summary = model.decode_fn(logits, seq_len, None, greedy=True)
    ref = octc.greedy_decode(logits.cpu().numpy(), seq_len.cpu().numpy())
    assert decoded == ref
    from ctc_asr_amd.labels import decode
    assert list(plaintext) == [decode(r) for r in ref]
    assert summary.shape == (2, 4) and summary[1, 0] == 'n/a'


@pytest.mark.parametrize('cudnn', [True, False])
def test_rnn_dropout_runs_and_is_off_in_eval(cudnn):
    cfg, flat, feats, flen, labels = _setup('ds2_lstm_2conv' if cudnn else 'ds1_basic_rnn')
    cfg.rnn_dropout_rate = 0.2
    model = CTCModel(cfg, 'cuda', params=flat)
    e1, s1 = model.inference_fn(torch.tensor(feats), torch.tensor(flen), training=False)
    e2, _ = model.inference_fn(torch.tensor(feats), torch.tensor(flen), training=False)
    assert torch.equal(e1, e2)
    t1, _ = model.inference_fn(torch.tensor(feats), torch.tensor(flen), training=True)
    assert not torch.equal(e1, t1)
    loss = model.loss_fn(t1, s1, labels)
    model.backward()
    assert torch.isfinite(loss) and torch.isfinite(model.arena.grad).all()
    assert float(model.arena.g['rnn0/w_ih'].abs().max()) > 0


def test_full_size_c2_shape_logits_loss_and_decode():
    """BASELINE configs[1] shape (DS2 2-conv + 2xBiLSTM-1024, 10 s = 999 frames -> T' = 500) at
    batch 2: logits / loss against the float64 torch restatement, greedy strings identical.
    Exercises the persistent recurrence kernels over all 500 steps."""
    cfg = ModelConfig(used_model='ds2', conv_filters=(32, 32), num_units_dense=2048,
                      num_layers_rnn=2, num_units_rnn=1024, rnn_cell='lstm', cudnn=True,
                      dense_dropout_rate=0.0)
    rng = np.random.default_rng(11)
    flat = init_params(cfg, 11)
    feats = rng.normal(size=(2, 999, 80)).astype(np.float32)
    flen = np.array([999, 999], dtype=np.int32)
    labels = [list(rng.integers(1, 28, size=150)), list(rng.integers(1, 28, size=150))]
    model = CTCModel(cfg, 'cuda', params=flat)
    logits, seq_len = model.inference_fn(torch.tensor(feats), torch.tensor(flen), training=True)
    loss = model.loss_fn(logits, seq_len, labels)
    model.backward()
    ref = torch_ref.TorchRefModel(to_oracle_layout(flat, cfg), 'ds2', 'lstm', True,
                                  dtype=torch.float64)
    with torch.no_grad():
        t_logits, t_len = ref(torch.tensor(feats, dtype=torch.float64), flen)
        t_loss, _ = ref.loss(t_logits, t_len, labels)
    assert logits.shape == (500, 2, 29) and (seq_len.cpu().numpy() == 500).all()
    assert np.abs(logits.cpu().numpy() - t_logits.numpy()).max() < 1e-3
    assert abs(float(loss) - float(t_loss)) < 1e-3 * max(1.0, abs(float(t_loss)))
    decoded, _, _ = model.decode_fn(logits, seq_len, None, greedy=True)
    assert decoded == octc.greedy_decode(t_logits.numpy(), [500, 500])
    assert torch.isfinite(model.arena.grad).all()
    from ctc_asr_amd import hip
    hip.rnn_poll_error('lstm', model._acts['rnn_ws'], 500, 2, 1024)


def test_full_size_c1_shape_with_gradients():
    """BASELINE configs[0]: DS1, 1 x BiRNN-256 (tanh BasicRNNCell, length aware), batch 2, 3 s
    = 299 frames; the reference's CPU plumbing config."""
    cfg = ModelConfig(used_model='ds1', num_units_dense=256, num_layers_rnn=1, num_units_rnn=256,
                      rnn_cell='rnn_tanh', cudnn=False, dense_dropout_rate=0.0)
    rng = np.random.default_rng(12)
    flat = init_params(cfg, 12)
    feats = rng.normal(size=(2, 299, 80)).astype(np.float32)
    flen = np.array([299, 251], dtype=np.int32)
    feats[1, 251:] = 0.0
    labels = [list(rng.integers(1, 28, size=45)), list(rng.integers(1, 28, size=30))]
    model = CTCModel(cfg, 'cuda', params=flat)
    logits, seq_len = model.inference_fn(torch.tensor(feats), torch.tensor(flen), training=True)
    loss = model.loss_fn(logits, seq_len, labels)
    model.backward()
    ref = torch_ref.TorchRefModel(to_oracle_layout(flat, cfg), 'ds1', 'rnn_tanh', False,
                                  dtype=torch.float64)
    t_logits, t_len = ref(torch.tensor(feats, dtype=torch.float64), flen)
    t_loss, _ = ref.loss(t_logits, t_len, labels)
    t_loss.backward()
    assert (seq_len.cpu().numpy() == flen).all()
    assert np.abs(logits.cpu().numpy() - t_logits.detach().numpy()).max() < 1e-3
    assert abs(float(loss) - float(t_loss.detach())) < 1e-3
    got = model.arena.export('grad')
    ref_g = ref.grads_in_shared_layout()
    for name, ref_t in (('rnn0/w_hh', ref_g['rnn'][0]['w_hh']),
                        ('dense0/kernel', ref_g['dense'][0][0]),
                        ('logits/kernel', ref_g['logits'][0])):
        err = np.abs(got[name] - ref_t.numpy()).max()
        assert err < 1e-3 * max(1.0, np.abs(ref_t.numpy()).max()), (name, err)


def _assembled_against_torch_ref(cfg, batch, frames, label_len, seed, grad_names,
                                 kinked=False):
    rng = np.random.default_rng(seed)
    flat = init_params(cfg, seed)
    feats = rng.normal(size=(batch, frames, 80)).astype(np.float32)
    flen = np.full(batch, frames, dtype=np.int32)
    labels = [list(rng.integers(1, 28, size=label_len)) for _ in range(batch)]
    model = CTCModel(cfg, 'cuda', params=flat)
    logits, seq_len = model.inference_fn(torch.tensor(feats), torch.tensor(flen), training=True)
    loss = model.loss_fn(logits, seq_len, labels)
    model.backward()
    model.check_rnn_error()
    ref = torch_ref.TorchRefModel(to_oracle_layout(flat, cfg), cfg.used_model, cfg.rnn_cell,
                                  cfg.cudnn, dtype=torch.float64)
    t_logits, t_len = ref(torch.tensor(feats, dtype=torch.float64), flen)
    t_loss, _ = ref.loss(t_logits, t_len, labels)
    t_loss.backward()
    t_out = cfg.output_time(frames)
    assert logits.shape == (t_out, batch, 29) and (seq_len.cpu().numpy() == t_out).all()
    assert np.abs(logits.cpu().numpy() - t_logits.detach().numpy()).max() < 1e-3
    assert abs(float(loss) - float(t_loss.detach())) < 1e-3 * max(1.0, abs(float(t_loss)))
    decoded, _, _ = model.decode_fn(logits, seq_len, None, greedy=True)
    assert decoded == octc.greedy_decode(t_logits.detach().numpy(), [t_out] * batch)
    got = model.arena.export('grad')
    ref_g = ref.grads_in_shared_layout()
    for name in grad_names:
        layer, leaf = name.split('/')
        if layer.startswith('rnn'):
            want = ref_g['rnn'][int(layer[3:])][leaf]
        elif layer.startswith('conv'):
            want = ref_g['conv'][int(layer[4:])][0 if leaf == 'kernel' else 1]
        else:
            want = ref_g[layer][0 if leaf == 'kernel' else 1]
        want = want.numpy()
        err = np.abs(got[name] - want).max()
        if kinked:
            # A stack of ReLU cells is not a smooth function of its inputs: wherever a
            # pre-activation sits within rounding distance of zero, two fp32-grade evaluations
            # land on different sides and a handful of gradient entries differ at the 1e-3 level
            # whatever the arithmetic (measured with the convolutions on the fp32 AND on the fp16
            # pipe: relative L2 error 7e-4 / 7e-4 for rnn0/w_ih, a few hundred of 4 M entries
            # off by > 1e-4 in either, largest single entry 5e-4 .. 1.3e-3 depending on which
            # units flip).  The bar for such a model is the gradient as a whole.
            rel = np.linalg.norm(got[name] - want) / max(np.linalg.norm(want), 1e-30)
            assert rel < 2e-3, (name, rel)
            assert err < 1e-2 * max(1.0, np.abs(want).max()), (name, err)
        else:
            assert err < 1e-3 * max(1.0, np.abs(want).max()), (name, err)
    return model


def test_assembled_c3_model_batch_32():
    """BASELINE configs[2] assembled: DS2 2-conv + FIVE BiLSTM-1024 layers at batch 32 (the
    per-GPU unit of the 8-GPU config) - the batch that takes the two-batch-tile kernels and no
    forward pipelining - at T' = 100: logits, loss, greedy strings and gradients of the first /
    middle / last layers against the float64 torch restatement; no recurrence time-out."""
    cfg = ModelConfig(used_model='ds2', conv_filters=(32, 32), num_units_dense=2048,
                      num_layers_rnn=5, num_units_rnn=1024, rnn_cell='lstm', cudnn=True,
                      dense_dropout_rate=0.0)
    _assembled_against_torch_ref(
        cfg, batch=32, frames=199, label_len=30, seed=21,
        grad_names=('conv0/kernel', 'conv1/kernel', 'rnn0/w_ih', 'rnn0/w_hh', 'rnn2/w_hh',
                    'rnn4/w_ih', 'rnn4/b_ih', 'dense4/kernel', 'logits/bias'))


@pytest.mark.timeout(1200)
def test_assembled_c3_model_at_the_benchmark_shape():
    """The shape `bench.py` times (BASELINE configs[2]: DS2 2-conv + 5 x BiLSTM-1024, batch 32,
    999 frames -> T' = 500, 150 labels per utterance) through the default path - fp16-split
    projection GEMMs, both recurrences on the fp16 matrix pipe - against the float64 torch
    restatement: logits and loss to 1e-3, greedy strings identical, and the gradients of the
    bottom layer's input weights, the top layer's recurrent weights and the first convolution
    (each has crossed the whole depth x length of the stack in one direction or the other) to
    1e-3 of their largest element."""
    cfg = ModelConfig(used_model='ds2', conv_filters=(32, 32), num_units_dense=2048,
                      num_layers_rnn=5, num_units_rnn=1024, rnn_cell='lstm', cudnn=True,
                      dense_dropout_rate=0.0)
    model = _assembled_against_torch_ref(
        cfg, batch=32, frames=999, label_len=150, seed=23,
        grad_names=('conv0/kernel', 'conv1/kernel', 'rnn0/w_ih', 'rnn0/w_hh', 'rnn2/w_hh',
                    'rnn4/w_hh', 'rnn4/w_ih', 'rnn4/b_ih', 'dense4/kernel', 'logits/kernel'))
    what = model.arithmetic()
    assert what['rnn0/input_projection'] == what['rnn4/input_projection'] == 'fp16x3'
    assert what['rnn2/recurrence_fwd'] == what['rnn2/recurrence_bwd'] == 'fp16x3'


@pytest.mark.timeout(900)
def test_fifty_training_steps_track_the_all_fp32_path(monkeypatch):
    """Drift: 50 Adam steps of the C3 model at the benchmark's shape on the same batch, once
    through the default arithmetic (fp16 / bf16 pieces on the 16-bit matrix pipe for the
    projection GEMMs, both recurrences and every convolution kernel) and once with every product
    on the fp32 pipe (CTCASR_SPLIT_GEMM=0, CTCASR_RNN_FWD_F16=0, CTCASR_RNN_BWD_F16=0,
    CTCASR_CONV_F16=0): the loss trajectories
    agree to 1e-3 relative at every step (they start bit-close and part only as fast as fp32
    round-off of either path is amplified by training itself)."""
    cfg = ModelConfig(used_model='ds2', conv_filters=(32, 32), num_units_dense=2048,
                      num_layers_rnn=5, num_units_rnn=1024, rnn_cell='lstm', cudnn=True,
                      dense_dropout_rate=0.0)
    rng = np.random.default_rng(29)
    feats = torch.tensor(rng.normal(size=(32, 999, 80)).astype(np.float32), device='cuda')
    flen = torch.full((32,), 999, dtype=torch.int32)
    labels = CTCModel.pack_labels([list(rng.integers(1, 28, size=150)) for _ in range(32)], 'cuda')

    def trajectory(fp32_everything):
        for name in ('CTCASR_SPLIT_GEMM', 'CTCASR_RNN_FWD_F16', 'CTCASR_RNN_BWD_F16',
                     'CTCASR_CONV_F16'):
            if fp32_everything:
                monkeypatch.setenv(name, '0')
            else:
                monkeypatch.delenv(name, raising=False)
        model = CTCModel(cfg, 'cuda', seed=3)
        assert model.split_gemm != fp32_everything and model.rnn_bwd_f16 != fp32_everything
        assert model.conv_f16 != fp32_everything
        losses = []
        for _ in range(50):
            losses.append(model.forward_backward(feats, flen, labels, check=False))
            model.apply_gradients(learning_rate=1e-4)
        model.check_rnn_error()
        for key in ('rnn2/recurrence_bwd', 'conv0/forward', 'conv1/forward'):
            assert model.arithmetic()[key] == ('fp32' if fp32_everything else 'fp16x3'), key
        out = torch.stack(losses).double().cpu().numpy()
        del model
        torch.cuda.empty_cache()
        return out

    split, plain = trajectory(False), trajectory(True)
    assert np.isfinite(split).all() and np.isfinite(plain).all()
    assert plain[-1] < 0.9 * plain[0]                      # it does train
    rel = np.abs(split - plain) / np.abs(plain)
    assert rel.max() < 1e-3, (int(rel.argmax()), float(rel.max()))


@pytest.mark.parametrize('frames,label_len', [(79, 12), (319, 40)])
def test_assembled_reference_default_model(frames, label_len):
    """The reference's own flag defaults (asr/params.py): 3 convolutions (32, 32, 96) + 4 x
    bidirectional ReLU-RNN-2048 + dense 2048, batch 16, at T' = 40 and at T' = 160 (VERDICT r04:
    the only gradient test of the reference's real default ran at 40 steps)."""
    cfg = ModelConfig(used_model='ds2', conv_filters=(32, 32, 96), num_units_dense=2048,
                      num_layers_rnn=4, num_units_rnn=2048, rnn_cell='rnn_relu', cudnn=True,
                      dense_dropout_rate=0.0)
    _assembled_against_torch_ref(
        cfg, batch=16, frames=frames, label_len=label_len, seed=22,
        grad_names=('conv2/kernel', 'rnn0/w_ih', 'rnn1/w_hh', 'rnn3/w_hh', 'dense4/kernel'),
        kinked=True)


@pytest.mark.timeout(1500)
@pytest.mark.parametrize('cell', ['rnn_relu', 'lstm'])
def test_the_reference_models_at_full_length(cell):
    """the reference's OWN models at the length the benchmark times them -
    `ref_default` (asr/params.py: 3 conv + 4 x BiRNN(relu)-2048) and `ref_best` (testruns.md:
    3 conv + 4 x BiLSTM-2048), batch 16, 999 frames -> T' = 500, 150 labels per utterance - through
    the default path (fp16-split projections, the fp16-pipe recurrences where they exist) against
    the float64 torch restatement: logits and loss to 1e-3, greedy strings identical.  (Forward
    and loss only: the float64 backward pass of 4 x 2048 units over 500 steps is the better part
    of an hour on the host; the gradients of these models are tested at T' <= 160 above and, for
    the kernels they run, step by step in test_gpu_kernels.py.)"""
    cfg = ModelConfig(used_model='ds2', conv_filters=(32, 32, 96), num_units_dense=2048,
                      num_layers_rnn=4, num_units_rnn=2048, rnn_cell=cell, cudnn=True,
                      dense_dropout_rate=0.0)
    batch, frames, label_len = 16, 999, 150
    rng = np.random.default_rng(31)
    flat = init_params(cfg, 31)
    feats = rng.normal(size=(batch, frames, 80)).astype(np.float32)
    flen = np.full(batch, frames, dtype=np.int32)
    labels = [list(rng.integers(1, 28, size=label_len)) for _ in range(batch)]
    model = CTCModel(cfg, 'cuda', params=flat)
    logits, seq_len = model.inference_fn(torch.tensor(feats), torch.tensor(flen), training=True)
    loss = model.loss_fn(logits, seq_len, labels)
    model.check_rnn_error()
    t_out = cfg.output_time(frames)
    assert t_out == 500 and logits.shape == (t_out, batch, 29)
    threads = torch.get_num_threads()
    torch.set_num_threads(max(threads, min(32, os.cpu_count() or 8)))
    try:
        ref = torch_ref.TorchRefModel(to_oracle_layout(flat, cfg), cfg.used_model, cfg.rnn_cell,
                                      cfg.cudnn, dtype=torch.float64)
        with torch.no_grad():
            t_logits, t_len = ref(torch.tensor(feats, dtype=torch.float64), flen)
            t_loss, _ = ref.loss(t_logits, t_len, labels)
    finally:
        torch.set_num_threads(threads)
    assert np.abs(logits.cpu().numpy() - t_logits.numpy()).max() < 1e-3
    assert abs(float(loss) - float(t_loss)) < 1e-3 * max(1.0, abs(float(t_loss)))
    decoded, _, _ = model.decode_fn(logits, seq_len, None, greedy=True)
    assert decoded == octc.greedy_decode(t_logits.numpy(), [t_out] * batch)


@pytest.mark.parametrize('cell,hidden,batch', [('lstm', 1024, 32), ('lstm', 1024, 20), ('lstm', 1024, 24),
                                               ('lstm', 2048, 16), ('rnn_relu', 2048, 16)])
def test_a_pass_reads_nothing_of_the_pass_before(cell, hidden, batch):
    """Workspaces, exchange buffers, packed operands and piece buffers are reused from pass to pass
    at the same addresses; a kernel that meets a cached or left-over copy of the pass before shows
    only when the data changes (round 6: tests that repeat one computation cannot see it).  One
    model: forward + backward over batch A, then over batch B; a second model with the same
    parameters: batch B only.  Loss, logits and every gradient slice of B agree to the noise of the
    CTC gradient's atomics."""
    cfg = ModelConfig(used_model='ds2', conv_filters=(32, 32), num_units_dense=2048,
                      num_layers_rnn=2, num_units_rnn=hidden, rnn_cell=cell, cudnn=True,
                      dense_dropout_rate=0.0, conv_dropout_rate=0.0)
    rng = np.random.default_rng(41)
    flat = init_params(cfg, 41)
    frames = 199

    def sample():
        feats = torch.tensor(rng.normal(size=(batch, frames, 80)).astype(np.float32), device='cuda')
        labels = [list(rng.integers(1, 28, size=25)) for _ in range(batch)]
        return feats, torch.full((batch,), frames, dtype=torch.int32), labels

    a, b = sample(), sample()
    first = CTCModel(cfg, 'cuda', params=flat)
    first.forward_backward(*a)
    loss1 = float(first.forward_backward(*b))
    first.check_rnn_error()
    grad1 = first.arena.grad.clone()
    second = CTCModel(cfg, 'cuda', params=flat)
    loss2 = float(second.forward_backward(*b))
    second.check_rnn_error()
    assert abs(loss1 - loss2) <= 1e-6 * abs(loss2), (loss1, loss2)
    for name, lo, hi in second.arena.layer_slices:
        want = second.arena.grad[lo:hi].double()
        rel = float((grad1[lo:hi].double() - want).norm() / want.norm().clamp_min(1e-30))
        assert rel < 1e-5, (name, rel)


def test_decode_many_equals_batch_by_batch_decoding():
    """`CTCModel.decode_many` (several batches in one beam-search launch, what
    `evaluate_dataset` uses) returns exactly what `decode_fn` returns batch by batch: different
    T', ragged lengths, a final smaller batch."""
    cfg, flat, _, _, _ = _setup('ds2_lstm_2conv')
    model = CTCModel(cfg, 'cuda', params=flat)
    rng = np.random.default_rng(77)
    batches = []
    for steps, batch in ((37, 4), (52, 4), (21, 3)):
        logits = (rng.normal(size=(steps, batch, cfg.num_classes)) * 2).astype(np.float32)
        logits[:, :, -1] += 1.5
        seq_len = rng.integers(1, steps + 1, size=batch).astype(np.int32)
        seq_len[0] = steps
        originals = np.array(['utt {}'.format(i).encode('utf-8') for i in range(batch)],
                             dtype=object)
        batches.append((torch.tensor(logits, device='cuda'),
                        torch.tensor(seq_len, device='cuda'), originals))
    for width in (8, 64):
        joint = model.decode_many(batches, beam_width=width)
        for (logits, seq_len, originals), got in zip(batches, joint):
            ref = model.decode_fn(logits, seq_len, originals, beam_width=width)
            assert got[0] == ref[0]
            assert list(got[1]) == list(ref[1])
            assert np.array_equal(got[2], ref[2])
    assert model.decode_group_size(500, 16, beam_width=1024) == 11
    assert model.decode_group_size(500, 16, beam_width=64) == 16
    assert model.decode_group_size(850, 32, beam_width=1024, budget_bytes=8 << 30) == 1
