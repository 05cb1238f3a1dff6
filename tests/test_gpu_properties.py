"""Size-independent properties of the HIP path at BASELINE.json's full sizes (C2/C3: T'=500,
batch 16/32, H=1024; C5: 17 s utterances, T'=850, beam width 64), where a line-by-line comparison
with the Python oracle would take minutes: invariances, symmetries, cross-kernel consistency and
determinism.  Integer / index results are compared bit-exactly, floating point within the stated
tolerances."""

import numpy as np
import pytest
import torch

from oracle import cref
from tests.helpers import pack_labels

pytestmark = pytest.mark.gpu
DEV = 'cuda'


def _t(a, dtype=torch.float32):
    return torch.as_tensor(np.ascontiguousarray(a), dtype=dtype).to(DEV)


def _ctc(hip, logits, labels, seq_len, scale=1.0):
    flat, offsets = pack_labels(labels)
    max_len = max(1, max(len(row) for row in labels))
    return hip.ctc_loss_fwd_bwd(_t(logits), _t(flat, torch.int32), _t(offsets, torch.int32),
                                _t(seq_len, torch.int32), max_len, grad_scale=scale)


@pytest.mark.parametrize('shape', [(500, 32, 150), (850, 8, 255)])     # C3 / C5 lattices
def test_ctc_full_size_properties(hip, shape):
    num_steps, batch, label_len = shape
    rng = np.random.default_rng(41)
    logits = (rng.normal(size=(num_steps, batch, 29)) * 2).astype(np.float32)
    labels = [list(rng.integers(0, 28, size=rng.integers(label_len // 2, label_len + 1)))
              for _ in range(batch)]
    seq_len = rng.integers(num_steps * 3 // 4, num_steps + 1, size=batch).astype(np.int32)
    seq_len[0] = num_steps
    loss, grad, status = _ctc(hip, logits, labels, seq_len)
    loss_np, grad_np = loss.cpu().numpy(), grad.cpu().numpy()
    assert (status.cpu().numpy() == 0).all() and np.isfinite(loss_np).all() and (loss_np > 0).all()
    # against the C oracle (float64 log-space recursion): loss 1e-3, gradient 1e-4
    ref_loss, ref_grad, _ = cref.ctc_loss(logits, labels, seq_len)
    assert np.abs(loss_np - ref_loss).max() < 1e-3
    assert np.abs(grad_np - ref_grad).max() < 1e-4
    # every live frame: softmax minus a distribution over the classes -> rows sum to zero;
    # frames beyond the utterance carry exactly no gradient
    for b in range(batch):
        assert np.abs(grad_np[:seq_len[b], b].sum(axis=1)).max() < 2e-5
        assert (grad_np[seq_len[b]:, b] == 0.0).all()
    # grad_scale is a plain factor (0.5: exact in binary floating point)
    _, half, _ = _ctc(hip, logits, labels, seq_len, scale=0.5)
    assert torch.equal(half, grad * 0.5)
    # a per-frame constant added to the logits changes nothing (softmax shift invariance)
    shift = rng.normal(size=(num_steps, batch, 1)).astype(np.float32) * 3
    loss_s, grad_s, _ = _ctc(hip, logits + shift, labels, seq_len)
    assert np.abs(loss_s.cpu().numpy() - loss_np).max() < 2e-3
    assert np.abs(grad_s.cpu().numpy() - grad_np).max() < 1e-5
    # utterances are independent: permuting the batch permutes the results bit for bit
    perm = rng.permutation(batch)
    loss_p, grad_p, _ = _ctc(hip, logits[:, perm], [labels[i] for i in perm], seq_len[perm])
    assert torch.equal(loss_p, loss[_t(perm, torch.int64)])
    assert torch.equal(grad_p, grad[:, _t(perm, torch.int64)])
    # deterministic run to run
    loss_2, grad_2, _ = _ctc(hip, logits, labels, seq_len)
    assert torch.equal(loss_2, loss)
    assert np.abs((grad_2 - grad).cpu().numpy()).max() < 1e-6   # LDS atomics: order may differ


def test_beam_search_is_consistent_with_the_ctc_loss(hip):
    """The log-probability the beam search reports for its best labelling sums a SUBSET of that
    labelling's alignments (those that stayed inside the beam), so it is bounded above by the
    exact ln p(labelling | x) = -ctc_loss, and approaches it as the beam widens."""
    rng = np.random.default_rng(42)
    num_steps, batch = 500, 4
    logits = (rng.normal(size=(num_steps, batch, 29)) * 3).astype(np.float32)
    logits[:, :, -1] += 3.0
    seq_len = np.array([500, 500, 377, 450], dtype=np.int32)
    def decode_and_score(lg, width):
        out, out_len, logp = hip.ctc_beam_decode(_t(lg), _t(seq_len, torch.int32), width,
                                                 normalization='log_softmax')
        out, out_len, logp = out.cpu().numpy(), out_len.cpu().numpy(), logp.cpu().numpy()
        labels = [out[b, :out_len[b]].tolist() for b in range(batch)]
        assert all(len(row) > 0 for row in labels)
        loss, _, status = _ctc(hip, lg, labels, seq_len)
        assert (status.cpu().numpy() == 0).all()
        return logp, -loss.cpu().numpy()

    found = {}
    for width in (1, 4, 64, 1024):
        logp, exact = decode_and_score(logits, width)
        assert (logp <= exact + 1e-2).all(), (width, logp, exact)
        found[width] = logp
    # a wider beam never reports a less probable best labelling
    assert (found[1024] >= found[4] - 1e-3).all() and (found[64] >= found[1] - 1e-3).all()
    # confident posteriors (what a trained network emits): nearly all of the labelling's
    # probability mass stays inside a wide beam, the bound becomes tight
    logp, exact = decode_and_score(logits * 4.0, 1024)
    assert (logp <= exact + 1e-2).all() and float((exact - logp).max()) < 0.5   # 3 nats above


@pytest.mark.parametrize('cell,hidden,batch,num_steps', [
    ('lstm', 1024, 16, 500), ('lstm', 1024, 32, 500), ('rnn_relu', 2048, 16, 500),
    # C5: 17 s utterances (T' = 850) through the BiLSTM-1024 kernels, one and two batch tiles
    ('lstm', 1024, 16, 850), ('lstm', 1024, 32, 850),
    # the round-2 kernels at full length: LSTM-2048 (one direction per launch), GRU-1024 / -2048
    ('lstm', 2048, 16, 500), ('gru', 1024, 16, 500), ('gru', 2048, 16, 500),
    ('gru', 1024, 32, 500),
    # 33..64 rows: two launches over blocks of 32 and 16 rows (own barrier words / exchange / carry)
    ('lstm', 1024, 48, 500)])
def test_recurrence_full_size_symmetries(hip, cell, hidden, batch, num_steps):
    """T'=500 / 850 through the persistent kernels: determinism, explicit full lengths == no
    lengths, a backward pass cut into step ranges == one launch, and the mirror symmetry of the
    two directions (time-reversed input with the directions' weights swapped gives the
    time-reversed output with the halves swapped) - all bit for bit.  Exercises the exchange
    buffer indexing, the carry between launches and the counter reset (`counters_done`) over
    the full length for every persistent kernel family."""
    gates = hip.CELL_GATES[cell]
    assert hip.rnn_persistent_supported(cell, num_steps, batch, hidden)
    gen = torch.Generator(device=DEV).manual_seed(7)
    scale = 0.5 if cell == 'lstm' else 0.05
    xw = torch.randn(num_steps, batch, 2, gates * hidden, device=DEV, generator=gen) * scale
    w_hh = torch.randn(2, gates * hidden, hidden, device=DEV, generator=gen) / np.sqrt(hidden)
    if cell not in ('lstm', 'gru'):
        w_hh *= 0.5
    dy = torch.randn(num_steps, batch, 2 * hidden, device=DEV, generator=gen)
    # (GRU: the candidate gate's recurrent bias is an argument of the forward pass)
    kw = {}
    if cell == 'gru':
        kw['b_hh_n'] = torch.randn(2, 3 * hidden, device=DEV, generator=gen) * 0.1
    kw_mirror = {k: v.flip(0).contiguous() for k, v in kw.items()}
    y, reserve, ws = hip.rnn_fwd(cell, xw, w_hh, **kw)
    hip.rnn_poll_error(cell, ws, num_steps, batch, hidden)
    assert torch.isfinite(y).all() and float(y.abs().max()) > 1e-3
    y_again, _, _ = hip.rnn_fwd(cell, xw, w_hh, **kw)
    assert torch.equal(y_again, y)
    full = torch.full((batch,), num_steps, dtype=torch.int32, device=DEV)
    y_len, _, _ = hip.rnn_fwd(cell, xw, w_hh, full, **kw)
    assert torch.equal(y_len, y)
    # a forward pass cut into step ranges == one launch (state carried through the workspace)
    y_cut, reserve_cut = torch.empty_like(y), torch.empty_like(reserve)
    cuts = [0, num_steps // 4, num_steps // 2 + 1, num_steps - 3, num_steps]
    for lo, hi in zip(cuts[:-1], cuts[1:]):
        hip.rnn_fwd(cell, xw, w_hh, y=y_cut, reserve=reserve_cut, workspace=ws, steps=(lo, hi),
                    **kw)
    hip.rnn_poll_error(cell, ws, num_steps, batch, hidden)
    assert torch.equal(y_cut, y)
    # mirror symmetry
    y_mirror, _, _ = hip.rnn_fwd(cell, xw.flip(0).flip(2).contiguous(), w_hh.flip(0).contiguous(),
                                 **kw_mirror)
    expect = torch.cat([y[..., hidden:], y[..., :hidden]], dim=-1).flip(0)
    assert torch.equal(y_mirror, expect)
    # backward: one launch vs four step ranges vs explicit lengths
    w_hh_t = hip.transpose_batched(w_hh)
    dxw = hip.rnn_bwd(cell, dy, y, w_hh_t, reserve, workspace=ws)
    hip.rnn_poll_error(cell, ws, num_steps, batch, hidden)
    assert torch.isfinite(dxw).all()
    dxw_cut = torch.empty_like(dxw)
    quarter = num_steps // 4
    for hi, lo in ((num_steps, 3 * quarter), (3 * quarter, 2 * quarter), (2 * quarter, quarter),
                   (quarter, 0)):
        hip.rnn_bwd(cell, dy, y, w_hh_t, reserve, dxw=dxw_cut, workspace=ws, steps=(lo, hi))
    hip.rnn_poll_error(cell, ws, num_steps, batch, hidden)
    assert torch.equal(dxw_cut, dxw)
    dxw_len = hip.rnn_bwd(cell, dy, y, w_hh_t, reserve, full, workspace=ws)
    assert torch.equal(dxw_len, dxw)


def test_features_at_the_maximum_utterance_length(hip):
    """17 s (MAX_EXAMPLE_LENGTH) next to 0.7 s (MIN) in one batch: frame counts, exact zero
    padding, unit statistics after 'local' normalisation, and the long utterance against the
    numpy oracle."""
    from oracle import features as ofeat
    rng = np.random.default_rng(43)
    lengths = np.array([272000, 11200, 160000], dtype=np.int32)
    pcm = np.zeros((3, lengths.max()), dtype=np.int16)
    for b, n in enumerate(lengths):
        pcm[b, :n] = np.clip(rng.normal(size=n) * 3000, -32768, 32767).astype(np.int16)
    out, out_len = hip.features(_t(pcm, torch.int16), _t(lengths, torch.int32), 'mel', 'local')
    out, out_len = out.cpu().numpy(), out_len.cpu().numpy()
    assert out_len.tolist() == [1699, 69, 999] and out.shape == (3, 1699, 80)
    for b in range(3):
        valid = out[b, :out_len[b]]
        assert (out[b, out_len[b]:] == 0.0).all()
        assert np.abs(valid.mean(axis=0)).max() < 1e-4
        assert np.abs(valid.std(axis=0) - 1.0).max() < 1e-3
    ref, ref_len = ofeat.load_sample_from_pcm(pcm[0, :lengths[0]], feature_type='mel',
                                              feature_normalization='local')
    assert ref_len == 1699
    assert np.abs(out[0] - ref).max() < 2e-3


def test_adam_full_size_against_plain_torch(hip):
    """122 M parameters (C3): the TF-form update against the same formula in torch ops."""
    n = 122_200_000
    gen = torch.Generator(device=DEV).manual_seed(3)
    param = torch.randn(n, device=DEV, generator=gen) * 0.05
    grad = torch.randn(n, device=DEV, generator=gen) * 0.01
    m = torch.randn(n, device=DEV, generator=gen) * 0.001
    v = torch.rand(n, device=DEV, generator=gen) * 1e-4
    lr, b1, b2, eps, step, scale = 1e-5, 0.9, 0.999, 1e-8, 17, 0.125
    g = grad * scale
    m_ref = b1 * m + (1 - b1) * g
    v_ref = b2 * v + (1 - b2) * g * g
    lr_t = lr * np.sqrt(1 - b2 ** step) / (1 - b1 ** step)
    p_ref = param - lr_t * m_ref / (v_ref.sqrt() + eps)
    hip.adam_step(param, grad, m, v, step, lr, b1, b2, eps, grad_scale=scale)
    assert float((m - m_ref).abs().max()) < 1e-9
    assert float((v - v_ref).abs().max()) < 2e-11       # one fp32 ulp at 1e-4 (FMA contraction)
    assert float((param - p_ref).abs().max()) < 6e-8    # 2 ulp at |param| ~ 0.25
    # "checksum of checksums": the update moved the parameters by the expected total amount
    assert abs(float(param.double().sum()) - float(p_ref.double().sum())) < 1e-3


@pytest.mark.parametrize('layer,batch,frames', [((40, 32), 32, 500), ((20, 96), 16, 500),
                                                # C5: the longest utterance (17 s, T' = 850)
                                                ((40, 32), 16, 850), ((20, 96), 16, 850)])
def test_convolution_kernels_are_adjoint_at_full_size(hip, layer, batch, frames):
    """C3-sized convolution (batch 32, T' = 500): forward, data gradient and kernel gradient are
    three views of one bilinear form, so  <dz, conv(x; w)> = <x, bwd_data(dz; w)> =
    <w, wrw(dz, x)>  ties the three own kernels together without any reference (fp64 sums of
    fp32 results: relative 1e-4); the kernel gradient is deterministic and linear in dz, and the
    time-major variants of the last layer give the same numbers."""
    freq, cout = layer
    gen = torch.Generator(device=DEV).manual_seed(7)
    x = torch.randn(batch, frames, freq, 32, device=DEV, generator=gen)
    dz = torch.randn(batch, frames, freq // 2, cout, device=DEV, generator=gen)
    w = torch.randn(cout, 32, 11, 21, device=DEV, generator=gen) * 0.05
    packed = hip.conv_s12_pack_weights(w)
    y = hip.conv_s12_fwd(x, packed, cout)
    dx = hip.conv_s12_bwd_data(dz, packed)
    dw = hip.conv_s12_wrw(dz, x)
    form_y = float((dz.double() * y.double()).sum())
    form_x = float((x.double() * dx.double()).sum())
    form_w = float((w.double() * dw.double()).sum())
    scale = max(abs(form_y), 1.0)
    assert abs(form_x - form_y) < 1e-4 * scale and abs(form_w - form_y) < 1e-4 * scale
    assert torch.equal(hip.conv_s12_wrw(dz, x), dw)                       # deterministic
    dw2 = hip.conv_s12_wrw(2.0 * dz, x)
    assert float((dw2 - 2.0 * dw).abs().max()) <= 1e-3 * float(dw.abs().max())   # linear in dz
    dz_tm = dz.permute(1, 0, 2, 3).contiguous()
    assert torch.equal(hip.conv_s12_wrw(dz_tm, x, time_major=True), dw)
    assert torch.equal(hip.conv_s12_bwd_data(dz_tm, packed, time_major=True), dx)
    assert torch.equal(hip.conv_s12_fwd(x, packed, cout, time_major=True).permute(1, 0, 2, 3), y)


@pytest.mark.parametrize('batch', [32, 24])
def test_two_tile_recurrences_equal_the_single_barrier_kernels_at_full_size(hip, batch):
    """C3 shape (T' = 500, H = 1024, batch 32; and a ragged second tile, batch 24): the round-2
    kernels that run the two 16-row batch tiles as independent recurrences - forward and
    whole-chip backward as groups of workgroups per tile, half-chip backward as two chains per
    workgroup - against the kernels that walk both tiles behind one barrier (same arithmetic
    per tile: 1e-6 / 1e-5), and every one of them is deterministic."""
    num_steps, hidden = 500, 1024
    gen = torch.Generator(device=DEV).manual_seed(11)
    xw = torch.randn(num_steps, batch, 2, 4 * hidden, device=DEV, generator=gen) * 0.5
    w_hh = torch.randn(2, 4 * hidden, hidden, device=DEV, generator=gen) / 32
    bias = torch.randn(2 * 4 * hidden, device=DEV, generator=gen) * 0.1
    dy = torch.randn(num_steps, batch, 2 * hidden, device=DEV, generator=gen)
    w_hh_t = hip.transpose_batched(w_hh)
    y_ref, reserve, ws = hip.rnn_fwd('lstm', xw, w_hh, xw_bias=bias, flags=hip.RNN_ONE_BARRIER)
    y_new, reserve_new, _ = hip.rnn_fwd('lstm', xw, w_hh, xw_bias=bias, workspace=ws)
    y_again, _, _ = hip.rnn_fwd('lstm', xw, w_hh, xw_bias=bias, workspace=ws)
    assert torch.equal(y_new, y_again)
    assert float((y_new - y_ref).abs().max()) < 1e-6
    assert float((reserve_new.view(torch.float32) - reserve.view(torch.float32)).abs().max()) < 1e-5
    dxw_ref = hip.rnn_bwd('lstm', dy, y_ref, w_hh_t, reserve, workspace=ws,
                          flags=hip.RNN_ONE_BARRIER)
    for flags in (hip.RNN_DEFAULT, hip.RNN_WHOLE_CHIP, hip.RNN_WHOLE_CHIP | hip.RNN_ONE_BARRIER,
                  hip.RNN_REDUCE_SCATTER):
        dxw = hip.rnn_bwd('lstm', dy, y_ref, w_hh_t, reserve, workspace=ws, flags=flags)
        again = hip.rnn_bwd('lstm', dy, y_ref, w_hh_t, reserve, workspace=ws, flags=flags)
        assert torch.equal(dxw, again), flags
        assert float((dxw - dxw_ref).abs().max()) < 1e-5 * max(1.0, float(dxw_ref.abs().max()))
    hip.rnn_poll_error('lstm', ws, num_steps, batch, hidden)
