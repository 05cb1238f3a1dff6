"""GPU parity of the individual HIP kernels (through the C ABI) against the CPU oracle."""

import numpy as np
import pytest
import torch

from oracle import ctc as octc
from oracle import cref
from oracle import nn as onn
from tests.helpers import pack_labels

pytestmark = pytest.mark.gpu
DEV = 'cuda'


def _t(a, dtype=torch.float32):
    return torch.as_tensor(np.ascontiguousarray(a), dtype=dtype).to(DEV)


def test_log_softmax_fwd_bwd(hip):
    rng = np.random.default_rng(0)
    x = (rng.normal(size=(37, 5, 29)) * 3).astype(np.float32)
    y = hip.log_softmax_fwd(_t(x)).cpu().numpy()
    ref = octc.log_softmax(x)
    assert np.abs(y - ref).max() < 1e-5
    dy = rng.normal(size=x.shape).astype(np.float32)
    dx = hip.log_softmax_bwd(_t(ref), _t(dy)).cpu().numpy()
    ref_dx = dy - np.exp(ref) * dy.sum(-1, keepdims=True)
    assert np.abs(dx - ref_dx).max() < 1e-5


@pytest.mark.parametrize('shape', [(5, 2, 6), (50, 4, 29), (500, 16, 29), (850, 3, 29),
                                   (1699, 2, 29)])
def test_ctc_loss_matches_oracle(hip, shape):
    num_steps, batch, classes = shape
    rng = np.random.default_rng(num_steps)
    logits = (rng.normal(size=shape) * 2).astype(np.float32)
    max_len = max(1, min(num_steps // 3, 422))
    labels = [list(rng.integers(0, classes - 1, size=rng.integers(0, max_len + 1)))
              for _ in range(batch)]
    labels[0] = list(rng.integers(0, classes - 1, size=max_len))
    if batch > 1:
        labels[1] = []          # empty label row
    seq_len = np.array([num_steps] + [int(rng.integers(max(1, 2 * max_len), num_steps + 1))
                                      for _ in range(batch - 1)], dtype=np.int32)
    flat, offsets = pack_labels(labels)
    loss, grad, status = hip.ctc_loss_fwd_bwd(_t(logits), _t(flat, torch.int32),
                                              _t(offsets, torch.int32),
                                              _t(seq_len, torch.int32), max_len, grad_scale=0.5)
    ref_loss, ref_grad, ref_status = cref.ctc_loss(logits, labels, seq_len)
    assert (status.cpu().numpy() == ref_status).all() and (ref_status == 0).all()
    # bar: 1e-3 on the loss (fp32, north_star); the kernel keeps the lattice in double
    assert np.abs(loss.cpu().numpy() - ref_loss).max() < 1e-3
    assert np.abs(grad.cpu().numpy() - 0.5 * ref_grad).max() < 1e-4


def test_ctc_loss_known_answer_tf(hip):
    import json
    import os
    kat = json.load(open(os.path.join(os.path.dirname(__file__), 'golden', 'ctc_kat.json')))
    for case in kat['loss_cases']:
        probs = np.array(case['probs'], dtype=np.float64)
        logits = np.log(probs).astype(np.float32)[:, None, :]
        flat, offsets = pack_labels([case['targets']])
        loss, grad, status = hip.ctc_loss_fwd_bwd(
            _t(logits), _t(flat, torch.int32), _t(offsets, torch.int32),
            _t(np.array([probs.shape[0]]), torch.int32), len(case['targets']))
        assert int(status[0]) == 0
        assert abs(float(loss[0]) - case['loss']) < 1e-4
        assert np.abs(grad.cpu().numpy()[:, 0, :] - np.array(case['grad'])).max() < 1e-5


def test_ctc_infeasible_and_bad_label(hip):
    rng = np.random.default_rng(3)
    logits = rng.normal(size=(4, 3, 6)).astype(np.float32)
    labels = [[0, 0, 0], [1, 2], [5]]      # needs 5 frames; fine; blank as label
    flat, offsets = pack_labels(labels)
    loss, grad, status = hip.ctc_loss_fwd_bwd(_t(logits), _t(flat, torch.int32),
                                              _t(offsets, torch.int32),
                                              _t(np.array([4, 4, 4]), torch.int32), 3)
    assert status.cpu().tolist() == [1, 0, 2]
    assert torch.isinf(loss[0]) and torch.isinf(loss[2]) and torch.isfinite(loss[1])
    assert float(grad[:, 0].abs().max()) == 0.0 and float(grad[:, 2].abs().max()) == 0.0


@pytest.mark.parametrize('poison', [float('nan'), float('inf')])
def test_ctc_loss_of_non_finite_logits_is_nan(hip, poison):
    """A NaN / inf logit inside an utterance's frames makes THAT utterance's loss NaN - what
    tf.nn.ctc_loss's log-softmax yields and the reference's NanTensorHook stops on
    (asr/model.py:259, :368); the max-based log-sum-exp of the lattice alone would drop the NaN
    paths, report a finite loss and let the NaN gradient through (found by the Trainer test of
    round 5).  The other utterances and frames beyond the length are unaffected."""
    rng = np.random.default_rng(13)
    logits = rng.normal(size=(40, 3, 29)).astype(np.float32)
    labels = [[1, 2, 3], [4, 5], [6, 7, 8, 9]]
    flat, offsets = pack_labels(labels)
    lens = np.array([40, 25, 40], dtype=np.int32)
    args = (_t(flat, torch.int32), _t(offsets, torch.int32), _t(lens, torch.int32), 4)
    clean, _, _ = hip.ctc_loss_fwd_bwd(_t(logits), *args)
    bad = logits.copy()
    bad[17, 0, 5] = poison          # inside utterance 0
    bad[30, 1, 2] = poison          # beyond utterance 1's 25 frames: not looked at
    loss, grad, status = hip.ctc_loss_fwd_bwd(_t(bad), *args)
    assert status.cpu().tolist() == [0, 0, 0]
    assert torch.isnan(loss[0])
    assert torch.equal(loss[1:], clean[1:]) and torch.isfinite(grad[:, 1:]).all()


@pytest.mark.parametrize('shape', [(7, 3, 6), (300, 5, 29), (777, 2, 29)])
def test_greedy_decode(hip, shape):
    num_steps, batch, classes = shape
    rng = np.random.default_rng(11)
    logits = rng.normal(size=shape).astype(np.float32)
    logits[:, :, -1] += 1.0      # more blanks
    seq_len = rng.integers(1, num_steps + 1, size=batch).astype(np.int32)
    seq_len[0] = num_steps
    out, out_len = hip.ctc_greedy_decode(_t(logits), _t(seq_len, torch.int32))
    ref = octc.greedy_decode(logits, seq_len)
    out, out_len = out.cpu().numpy(), out_len.cpu().numpy()
    for b in range(batch):
        assert out[b, :out_len[b]].tolist() == ref[b]
        assert (out[b, out_len[b]:] == 0).all()


@pytest.mark.parametrize('cell', ['lstm', 'rnn_tanh', 'rnn_relu', 'gru'])
@pytest.mark.parametrize('use_len', [False, True])
@pytest.mark.parametrize('dims', [(9, 3, 64), (23, 18, 128), (12, 16, 1024), (7, 19, 1024),
                                  # B = 33..64: two persistent launches over blocks of <= 32 rows
                                  (5, 35, 1024), (6, 64, 1024), (4, 40, 2048),
                                  (5, 70, 1024),   # B > 64 -> streaming at H=1024
                                  (6, 16, 2048), (4, 21, 2048)])
def test_rnn_fwd_bwd(hip, cell, use_len, dims):
    # every cell at every shape: the shapes a cell has no persistent kernel for (gru / relu / tanh
    # at H=1024, B > 64) are exactly where the streaming kernels are the only path
    num_steps, batch, hidden = dims
    gates = onn.GATES[cell]
    rng = np.random.default_rng(5)
    xw = (rng.normal(size=(num_steps, batch, 2, gates * hidden)) * 0.5).astype(np.float32)
    w_hh = (rng.normal(size=(2, gates * hidden, hidden)) / np.sqrt(hidden)).astype(np.float32)
    seq_len = None
    if use_len:
        seq_len = rng.integers(1, num_steps + 1, size=batch).astype(np.int32)
        seq_len[0] = num_steps
    dy = rng.normal(size=(num_steps, batch, 2 * hidden)).astype(np.float32)

    # reference: torch CPU autograd over the same recurrence written with plain ops (float64)
    b_hh = (rng.normal(size=(2, gates * hidden)) * 0.3).astype(np.float32) if cell == 'gru' \
        else None
    xw_t = torch.tensor(xw, dtype=torch.float64, requires_grad=True)
    w_t = torch.tensor(w_hh, dtype=torch.float64, requires_grad=True)
    ys = torch.zeros(num_steps, batch, 2 * hidden, dtype=torch.float64)
    out_rows = []
    for d in (0, 1):
        for b in range(batch):
            steps = int(seq_len[b]) if use_len else num_steps
            h = torch.zeros(hidden, dtype=torch.float64)
            c = torch.zeros(hidden, dtype=torch.float64)
            for s in range(steps):
                t = s if d == 0 else steps - 1 - s
                pre = xw_t[t, b, d] + w_t[d] @ h
                if cell == 'gru':
                    rec = w_t[d] @ h
                    xr, xz, xn = xw_t[t, b, d].split(hidden)
                    rr, rz, rn = rec.split(hidden)
                    r = torch.sigmoid(xr + rr)
                    z = torch.sigmoid(xz + rz)
                    n = torch.tanh(xn + r * (rn + torch.tensor(b_hh[d, 2 * hidden:],
                                                               dtype=torch.float64)))
                    h = (1 - z) * n + z * h
                elif cell == 'lstm':
                    i, f, g, o = pre.split(hidden)
                    c = torch.sigmoid(f) * c + torch.sigmoid(i) * torch.tanh(g)
                    h = torch.sigmoid(o) * torch.tanh(c)
                elif cell == 'rnn_tanh':
                    h = torch.tanh(pre)
                else:
                    h = torch.relu(pre)
                out_rows.append((t, b, d, h))
    ys = torch.zeros(num_steps, batch, 2, hidden, dtype=torch.float64)
    idx_t = torch.tensor([r[0] for r in out_rows])
    idx_b = torch.tensor([r[1] for r in out_rows])
    idx_d = torch.tensor([r[2] for r in out_rows])
    ys = ys.index_put((idx_t, idx_b, idx_d), torch.stack([r[3] for r in out_rows]))
    ys = ys.reshape(num_steps, batch, 2 * hidden)
    (ys * torch.tensor(dy, dtype=torch.float64)).sum().backward()

    sl = _t(seq_len, torch.int32) if use_len else None
    y, reserve, ws = hip.rnn_fwd(cell, _t(xw), _t(w_hh), sl,
                                 b_hh_n=_t(b_hh) if cell == 'gru' else None)
    hip.rnn_poll_error(cell, ws, num_steps, batch, hidden)
    assert np.abs(y.cpu().numpy() - ys.detach().numpy()).max() < 2e-5
    # the same forward pass cut into three launches (ctcasr_rnn_fwd_steps) is bit-identical, and
    # so is the half-chip variant of the persistent LSTM kernel
    if num_steps >= 5:
        cuts = [0, 2, num_steps // 2 + 1, num_steps]
        y_cut = torch.full_like(y, float('nan'))
        reserve_cut, ws_cut = torch.zeros_like(reserve), torch.zeros_like(ws)
        for lo, hi in zip(cuts[:-1], cuts[1:]):
            hip.rnn_fwd(cell, _t(xw), _t(w_hh), sl, b_hh_n=_t(b_hh) if cell == 'gru' else None,
                        y=y_cut, reserve=reserve_cut, workspace=ws_cut, steps=(lo, hi))
        hip.rnn_poll_error(cell, ws_cut, num_steps, batch, hidden)
        assert torch.equal(y_cut, y)
        with pytest.raises(ValueError):
            hip.rnn_fwd(cell, _t(xw), _t(w_hh), sl, steps=(0, 2))
    if cell == 'lstm' and hidden == 1024 and batch <= 16:
        y_half, _, ws_half = hip.rnn_fwd(cell, _t(xw), _t(w_hh), sl, flags=hip.RNN_HALF_CHIP)
        hip.rnn_poll_error(cell, ws_half, num_steps, batch, hidden)
        assert np.abs((y_half - y).cpu().numpy()).max() < 1e-6
    w_hh_t = hip.transpose_batched(_t(w_hh))
    assert torch.equal(w_hh_t.cpu(), torch.tensor(w_hh).transpose(1, 2).contiguous())
    # ``dbias``: the bias gradients (column sums of dxw over time and batch; GRU: then those of
    # drec) accumulate inside the recurrence kernels / at the end of the streaming pass
    gh = gates * hidden
    dbias = torch.zeros(2 * gh * (2 if cell == 'gru' else 1), device=DEV)
    dxw = hip.rnn_bwd(cell, _t(dy), y, w_hh_t, reserve, sl, dbias=dbias, workspace=ws)
    hip.rnn_poll_error(cell, ws, num_steps, batch, hidden)
    assert np.abs(dxw.cpu().numpy() - xw_t.grad.numpy()).max() < 1e-4

    def bias_sums():
        want = [dxw.double().sum(dim=(0, 1)).reshape(-1)]
        if cell == 'gru':
            want.append(hip.rnn_gru_drec(reserve, num_steps, batch, hidden).double()
                        .sum(dim=(0, 1)).reshape(-1))
        return torch.cat(want)
    want_db = bias_sums()
    assert float((dbias.double() - want_db).abs().max()) < 1e-4 * max(1.0, float(want_db.abs().max()))
    # the same pass cut into three launches (ctcasr_rnn_bwd_steps) is bit-identical
    if num_steps >= 5:
        cuts = [num_steps, num_steps - 2, num_steps // 2, 0]
        dxw_cut = torch.full_like(dxw, float('nan'))
        dbias_cut = torch.zeros_like(dbias)
        for hi, lo in zip(cuts[:-1], cuts[1:]):
            hip.rnn_bwd(cell, _t(dy), y, w_hh_t, reserve, sl, dxw=dxw_cut, dbias=dbias_cut,
                        workspace=ws, steps=(lo, hi))
        hip.rnn_poll_error(cell, ws, num_steps, batch, hidden)
        assert torch.equal(dxw_cut, dxw)
        assert float((dbias_cut.double() - want_db).abs().max()) < \
            1e-4 * max(1.0, float(want_db.abs().max()))
        with pytest.raises(ValueError):
            hip.rnn_bwd(cell, _t(dy), y, w_hh_t, reserve, sl, steps=(0, 2))
        with pytest.raises(RuntimeError):
            hip.rnn_bwd(cell, _t(dy), y, w_hh_t, reserve, sl, dxw=dxw_cut, workspace=ws,
                        steps=(3, 3))
    if cell == 'gru' and not use_len:
        # dW_hh is a GEMM of drec (dxw with the candidate gate scaled by r) and h_{t-1}
        drec = hip.rnn_gru_drec(reserve, num_steps, batch, hidden).cpu().double()
        yd = ys.detach()
        dw0 = drec[1:, :, 0].reshape(-1, 3 * hidden).t() @ yd[:-1, :, :hidden].reshape(-1, hidden)
        assert np.abs(dw0.numpy() - w_t.grad[0].numpy()).max() < 1e-3


def _recurrence_float64(cell, xw, w_hh, b_hh, seq_len):
    """Forward recurrence in float64 with batched torch ops on the GPU: y [T, B, 2H]."""
    num_steps, batch, _, _ = xw.shape
    hidden = w_hh.shape[2]
    x64, w64 = xw.double(), w_hh.double()
    steps = torch.full((batch,), num_steps, device=xw.device, dtype=torch.long) \
        if seq_len is None else seq_len.long()
    rows = torch.arange(batch, device=xw.device)
    y = torch.zeros(num_steps, batch, 2 * hidden, dtype=torch.float64, device=xw.device)
    for d in (0, 1):
        h = torch.zeros(batch, hidden, dtype=torch.float64, device=xw.device)
        c = torch.zeros_like(h)
        for s in range(num_steps):
            active = s < steps
            t = torch.where(active, torch.full_like(steps, s) if d == 0 else steps - 1 - s,
                            torch.zeros_like(steps))
            x = x64[t, rows, d]
            rec = h @ w64[d].t()
            if cell == 'rnn_relu':
                h_new = torch.relu(x + rec)
            elif cell == 'lstm':
                i, f, g, o = (x + rec).split(hidden, dim=1)
                c_new = torch.sigmoid(f) * c + torch.sigmoid(i) * torch.tanh(g)
                h_new = torch.sigmoid(o) * torch.tanh(c_new)
                c = torch.where(active[:, None], c_new, c)
            else:
                xr, xz, xn = x.split(hidden, dim=1)
                rr, rz, rn = rec.split(hidden, dim=1)
                r, z = torch.sigmoid(xr + rr), torch.sigmoid(xz + rz)
                n = torch.tanh(xn + r * (rn + b_hh[d, 2 * hidden:].double()))
                h_new = (1 - z) * n + z * h
            h = torch.where(active[:, None], h_new, h)
            y[t[active], rows[active], d * hidden:(d + 1) * hidden] = h[active]
    return y


@pytest.mark.parametrize('cell', ['lstm', 'gru'])
@pytest.mark.parametrize('use_len', [False, True])
@pytest.mark.parametrize('dims', [(12, 16, 1024), (40, 7, 1024), (9, 19, 1024), (7, 32, 1024),
                                  (5, 35, 1024), (6, 16, 2048), (4, 21, 2048)])
def test_rnn_fwd_on_the_fp16_matrix_pipe(hip, cell, use_len, dims):
    """CTCASR_RNN_F16: h W_hh^T of the persistent LSTM / GRU forward kernels as two fp16 pieces
    per operand and three products.  y against a float64 recurrence inside the bar of the fp32
    kernel's own test (2e-5), next to the fp32 kernel on the same inputs; step ranges and the
    half-chip variant bit-identical / equal; the backward pass from ITS reserve."""
    num_steps, batch, hidden = dims
    gates = onn.GATES[cell]
    g = torch.Generator(device=DEV).manual_seed(31)
    xw = torch.randn(num_steps, batch, 2, gates * hidden, device=DEV, generator=g) * 0.5
    w_hh = torch.randn(2, gates * hidden, hidden, device=DEV, generator=g) / np.sqrt(hidden)
    b_hh = torch.randn(2, gates * hidden, device=DEV, generator=g) * 0.3 if cell == 'gru' else None
    bias = torch.randn(2 * gates * hidden, device=DEV, generator=g) * 0.1
    sl = None
    if use_len:
        sl = torch.randint(1, num_steps + 1, (batch,), device=DEV, generator=g).int()
        sl[0] = num_steps
    assert hip.rnn_persistent_supported(cell, num_steps, batch, hidden)
    ref = _recurrence_float64(cell, xw + bias.view(1, 1, 2, -1), w_hh, b_hh, sl)
    y32, _, ws32 = hip.rnn_fwd(cell, xw, w_hh, sl, b_hh_n=b_hh, xw_bias=bias)
    y16, reserve, ws = hip.rnn_fwd(cell, xw, w_hh, sl, b_hh_n=b_hh, xw_bias=bias,
                                   flags=hip.RNN_F16)
    y_f16 = y16
    hip.rnn_poll_error(cell, ws, num_steps, batch, hidden)
    hip.rnn_poll_error(cell, ws32, num_steps, batch, hidden)
    err16 = float((y16.double() - ref).abs().max())
    err32 = float((y32.double() - ref).abs().max())
    assert err16 < 2e-5, (err16, err32)
    assert err16 < 3 * err32 + 2e-6, (err16, err32)
    assert not torch.equal(y16, y32)             # (it IS another kernel)
    # step ranges: bit-identical to the single launch (h crosses launches as its fp16 pieces)
    if num_steps >= 5:
        cuts = [0, 2, num_steps // 2 + 1, num_steps]
        y_cut = torch.full_like(y16, float('nan'))
        reserve_cut, ws_cut = torch.zeros_like(reserve), torch.zeros_like(ws)
        for lo, hi in zip(cuts[:-1], cuts[1:]):
            hip.rnn_fwd(cell, xw, w_hh, sl, b_hh_n=b_hh, xw_bias=bias, y=y_cut,
                        reserve=reserve_cut, workspace=ws_cut, steps=(lo, hi), flags=hip.RNN_F16)
        hip.rnn_poll_error(cell, ws_cut, num_steps, batch, hidden)
        assert torch.equal(y_cut, y16)
        if cell == 'lstm' and not use_len:  # (parts of the reserve a forward pass does not write:
            assert torch.equal(reserve_cut, reserve)    # the GRU's drec, rows past their length)
    # the kernel's own fp16 pieces of y (what it publishes, written in the layout of split_f16):
    # bit for bit the split of the y it wrote; not offered where rows end early
    from ctc_asr_amd import split_gemm
    pieces16 = torch.zeros(num_steps * batch, 3, 2 * hidden, dtype=torch.float16, device=DEV)
    if use_len:
        with pytest.raises(hip.CtcAsrError):
            hip.rnn_fwd(cell, xw, w_hh, sl, b_hh_n=b_hh, xw_bias=bias, flags=hip.RNN_F16, y16=pieces16)
    else:
        y_again, _, ws_p = hip.rnn_fwd(cell, xw, w_hh, sl, b_hh_n=b_hh, xw_bias=bias,
                                       flags=hip.RNN_F16, y16=pieces16)
        hip.rnn_poll_error(cell, ws_p, num_steps, batch, hidden)
        assert torch.equal(y_again, y_f16)
        want = hip.split_f16(y_again.view(num_steps * batch, 2 * hidden),
                             split_gemm.RNN_F16_H_SCALE, split_gemm.H_A)
        assert torch.equal(pieces16.view(torch.int16), want.view(torch.int16))
        with pytest.raises(hip.CtcAsrError):        # the fp32 kernel has no pieces to give
            hip.rnn_fwd(cell, xw, w_hh, sl, b_hh_n=b_hh, xw_bias=bias, y16=pieces16)
    if cell == 'lstm' and hidden == 1024 and batch <= 16:
        # half of the chip: another split of the same sums (16 units per workgroup: other scales)
        y_half, _, ws_half = hip.rnn_fwd(cell, xw, w_hh, sl, xw_bias=bias,
                                         flags=hip.RNN_F16 | hip.RNN_HALF_CHIP)
        hip.rnn_poll_error(cell, ws_half, num_steps, batch, hidden)
        assert float((y_half - y16).abs().max()) < 2e-6
    # the backward pass (fp32 kernels) from this forward pass's reserve: dxw against autograd
    # through the float64 recurrence is covered by test_rnn_fwd_bwd for the fp32 reserve; here:
    # the two reserves differ by the forward kernels' rounding only
    dy = torch.randn(num_steps, batch, 2 * hidden, device=DEV, generator=g)
    w_hh_t = hip.transpose_batched(w_hh)
    _, reserve32, _ = hip.rnn_fwd(cell, xw, w_hh, sl, b_hh_n=b_hh, xw_bias=bias)
    dxw16 = hip.rnn_bwd(cell, dy, y16, w_hh_t, reserve, sl, workspace=ws)
    dxw32 = hip.rnn_bwd(cell, dy, y32, w_hh_t, reserve32, sl, workspace=ws)
    hip.rnn_poll_error(cell, ws, num_steps, batch, hidden)
    assert float((dxw16 - dxw32).abs().max()) < 1e-4


@pytest.mark.parametrize('use_len', [False, True])
@pytest.mark.parametrize('dims', [(12, 16, 1024), (30, 7, 1024), (9, 19, 1024), (11, 32, 1024),
                                  (5, 35, 1024), (6, 64, 1024),
                                  (12, 16, 2048), (9, 5, 2048), (7, 27, 2048), (5, 40, 2048)])
def test_rnn_bwd_on_the_fp16_matrix_pipe(hip, use_len, dims):
    """CTCASR_RNN_F16 on the backward LSTM-1024 kernel: dgates as two fp16 pieces scaled per
    (producer workgroup, row), W_hh per workgroup, three products.  dxw against autograd through
    the float64 recurrence next to the fp32 kernel's error; the column maxima the kernel
    accumulates are exactly max |dxw|; step ranges bit-identical; both chip arrangements."""
    num_steps, batch, hidden = dims
    g = torch.Generator(device=DEV).manual_seed(41)
    xw = torch.randn(num_steps, batch, 2, 4 * hidden, device=DEV, generator=g) * 0.5
    w_hh = torch.randn(2, 4 * hidden, hidden, device=DEV, generator=g) / np.sqrt(hidden)
    # gradients over many decades, rows (utterances) of very different size
    dy = torch.randn(num_steps, batch, 2 * hidden, device=DEV, generator=g) * \
        torch.logspace(-6, 0, batch, device=DEV).view(1, batch, 1)
    sl = None
    if use_len:
        sl = torch.randint(1, num_steps + 1, (batch,), device=DEV, generator=g).int()
        sl[0] = num_steps
    assert hip.rnn_bwd_f16_supported('lstm', num_steps, batch, hidden)
    xw64 = xw.double().requires_grad_(True)
    ref_y = _recurrence_float64('lstm', xw64, w_hh, None, sl)
    (ref_y * dy.double()).sum().backward()
    ref = xw64.grad
    y, reserve, ws = hip.rnn_fwd('lstm', xw, w_hh, sl)
    w_hh_t = hip.transpose_batched(w_hh)
    gh = 4 * hidden
    db32, db16 = torch.zeros(2 * gh, device=DEV), torch.zeros(2 * gh, device=DEV)
    colmax = torch.zeros(2 * gh, dtype=torch.int32, device=DEV)
    dxw32 = hip.rnn_bwd('lstm', dy, y, w_hh_t, reserve, sl, dbias=db32, workspace=ws)
    dxw16 = hip.rnn_bwd('lstm', dy, y, w_hh_t, reserve, sl, dbias=db16, workspace=ws,
                        flags=hip.RNN_F16, colmax=colmax)
    hip.rnn_poll_error('lstm', ws, num_steps, batch, hidden)
    assert not torch.equal(dxw16, dxw32)
    # per row (utterance): the error relative to that row's largest gradient
    def row_err(got):
        err = (got.double() - ref).abs().amax(dim=(0, 2, 3))
        return float((err / ref.abs().amax(dim=(0, 2, 3)).clamp_min(1e-30)).max())
    e16, e32 = row_err(dxw16), row_err(dxw32)
    assert e16 < 3 * e32 + 1e-6, (e16, e32)
    assert float((db16 - db32).abs().max()) < 1e-4 * max(1.0, float(db32.abs().max()))
    want = dxw16.abs().amax(dim=(0, 1)).reshape(-1)
    assert torch.equal(colmax.view(torch.float32), want)
    # whole chip: each 16-row tile as its own group of workgroups
    dxw_w = hip.rnn_bwd('lstm', dy, y, w_hh_t, reserve, sl, workspace=ws,
                        flags=hip.RNN_F16 | hip.RNN_WHOLE_CHIP)
    hip.rnn_poll_error('lstm', ws, num_steps, batch, hidden)
    assert row_err(dxw_w) < 3 * e32 + 1e-6
    if num_steps >= 5:
        cuts = [num_steps, num_steps - 2, num_steps // 2, 0]
        dxw_cut = torch.full_like(dxw16, float('nan'))
        db_cut = torch.zeros_like(db16)
        colmax_cut = torch.zeros_like(colmax)
        for hi, lo in zip(cuts[:-1], cuts[1:]):
            hip.rnn_bwd('lstm', dy, y, w_hh_t, reserve, sl, dxw=dxw_cut, dbias=db_cut,
                        workspace=ws, steps=(lo, hi), flags=hip.RNN_F16, colmax=colmax_cut)
        hip.rnn_poll_error('lstm', ws, num_steps, batch, hidden)
        assert torch.equal(dxw_cut, dxw16)
        assert torch.equal(colmax_cut, colmax)
    # no fp16 kernel for this call -> no column maxima from it
    with pytest.raises(hip.CtcAsrError):
        hip.rnn_bwd('lstm', dy, y, w_hh_t, reserve, sl, workspace=ws, colmax=colmax)


@pytest.mark.parametrize('xcd', [0, 1])
@pytest.mark.parametrize('dims', [(12, 32), (9, 19), (7, 17), (61, 27), (11, 24), (200, 32)])
def test_rnn_bwd_staggered_tiles_equal_one_barrier(hip, xcd, dims):
    """CTCASR_RNN_STAGGER (prnn_bwd16s_kernel): the two 16-row tiles of a 17..32-row batch half
    a step apart, each with its own arrival counters, next phase's operands requested under this
    phase's matrix work.  Every sum keeps the order of the one-barrier kernel: dxw, the bias
    gradients, the column maxima AND what the kernel publishes for the data-gradient kernel (the
    exchange blocks and inverse scales of every step) are bit-identical - a stale or early read
    of the exchange would show here; step ranges and repeated passes on one workspace (counters
    handed back clean) too.  Calls with per-row lengths keep the one-barrier kernel."""
    num_steps, batch = dims
    hidden, gh = 1024, 4096
    g = torch.Generator(device=DEV).manual_seed(43)
    xw = torch.randn(num_steps, batch, 2, gh, device=DEV, generator=g) * 0.5
    w_hh = torch.randn(2, gh, hidden, device=DEV, generator=g) / np.sqrt(hidden)
    dy = torch.randn(num_steps, batch, 2 * hidden, device=DEV, generator=g) * \
        torch.logspace(-5, 0, batch, device=DEV).view(1, batch, 1)
    y, reserve, ws = hip.rnn_fwd('lstm', xw, w_hh)
    w_hh_t = hip.transpose_batched(w_hh)
    base = hip.RNN_F16 | (hip.RNN_XCD_SPLIT if xcd else 0)
    x_off, s_off = hip.dgrad16_published_offsets(num_steps, batch, hidden)
    x_bytes = (num_steps + 1) * 2 * batch * gh * 4
    s_bytes = (num_steps + 1) * 2 * 64 * 32 * 4

    def run(flags, cuts=None):
        db = torch.zeros(2 * gh, device=DEV)
        colmax = torch.zeros(2 * gh, dtype=torch.int32, device=DEV)
        dxw = torch.full((num_steps, batch, 2, gh), float('nan'), device=DEV)
        cuts = cuts or [num_steps, 0]
        for hi, lo in zip(cuts[:-1], cuts[1:]):
            hip.rnn_bwd('lstm', dy, y, w_hh_t, reserve, None, dxw=dxw, dbias=db, workspace=ws,
                        steps=(lo, hi), flags=flags, colmax=colmax)
        hip.rnn_poll_error('lstm', ws, num_steps, batch, hidden)
        published = (ws[x_off:x_off + x_bytes].clone(), ws[s_off:s_off + s_bytes].clone())
        return dxw, db, colmax, published

    want = run(base)
    assert torch.isfinite(want[0]).all()
    for attempt in range(3):           # (repeated: the counters of the pass before are clean again)
        got = run(base | hip.RNN_STAGGER)
        assert torch.equal(got[0], want[0]), attempt
        assert torch.equal(got[1], want[1])
        assert torch.equal(got[2], want[2])
        assert torch.equal(got[3][0], want[3][0])
        assert torch.equal(got[3][1], want[3][1])
    if num_steps >= 7:
        cuts = [num_steps, num_steps - 1, num_steps - 3, num_steps // 2, 1, 0]
        got = run(base | hip.RNN_STAGGER, cuts)
        assert torch.equal(got[0], want[0])
        assert torch.equal(got[2], want[2])
        assert torch.equal(got[3][0], want[3][0])
        # (bias gradients: one atomic per launch and column - another order of the same sums)
        assert float((got[1] - want[1]).abs().max()) <= 1e-5 * max(1.0, float(want[1].abs().max()))
        # the two kernels may take turns inside one pass
        db = torch.zeros(2 * gh, device=DEV)
        dxw = torch.full_like(want[0], float('nan'))
        for k, (hi, lo) in enumerate(zip(cuts[:-1], cuts[1:])):
            hip.rnn_bwd('lstm', dy, y, w_hh_t, reserve, None, dxw=dxw, dbias=db, workspace=ws,
                        steps=(lo, hi), flags=base | (hip.RNN_STAGGER if k % 2 else 0))
        hip.rnn_poll_error('lstm', ws, num_steps, batch, hidden)
        assert torch.equal(dxw, want[0])
    # per-row lengths: the flag is ignored (one-barrier kernel), same results as without it
    sl = torch.randint(1, num_steps + 1, (batch,), device=DEV, generator=g).int()
    y_l, reserve_l, ws_l = hip.rnn_fwd('lstm', xw, w_hh, sl)
    a = hip.rnn_bwd('lstm', dy, y_l, w_hh_t, reserve_l, sl, workspace=ws_l, flags=base)
    b = hip.rnn_bwd('lstm', dy, y_l, w_hh_t, reserve_l, sl, workspace=ws_l,
                    flags=base | hip.RNN_STAGGER)
    hip.rnn_poll_error('lstm', ws_l, num_steps, batch, hidden)
    assert torch.equal(a, b)


@pytest.mark.parametrize('xcd', [0, 1])
@pytest.mark.parametrize('dims', [(12, 32), (7, 17), (61, 24)])
def test_rnn_bwd_k_pairs(hip, xcd, dims):
    """CTCASR_RNN_KPAIR (prnn_bwd16k_kernel, round 6): pairs of workgroups share 32 hidden units,
    each multiplies ONE K half of the published dgates and hands its partner a [16 x 16] partial
    tile through L2 (words tagged with the parity of the slot's write count).  dxw against autograd
    through the float64 recurrence within the staggered kernel's error (the sums only differ in
    their order); column maxima exact; step ranges of any cut - an odd number of steps flips the
    tag parity the next launch starts from - bit-identical to one launch, repeated passes on one
    workspace too; what it publishes is what the staggered kernel would publish from the same
    sums (the data-gradient kernel reads it); the kernels may take turns inside a pass."""
    num_steps, batch = dims
    hidden, gh = 1024, 4096
    g = torch.Generator(device=DEV).manual_seed(47)
    xw = torch.randn(num_steps, batch, 2, gh, device=DEV, generator=g) * 0.5
    w_hh = torch.randn(2, gh, hidden, device=DEV, generator=g) / np.sqrt(hidden)
    dy = torch.randn(num_steps, batch, 2 * hidden, device=DEV, generator=g) * \
        torch.logspace(-5, 0, batch, device=DEV).view(1, batch, 1)
    xw64 = xw.double().requires_grad_(True)
    ref_y = _recurrence_float64('lstm', xw64, w_hh, None, None)
    (ref_y * dy.double()).sum().backward()
    ref = xw64.grad
    y, reserve, ws = hip.rnn_fwd('lstm', xw, w_hh)
    w_hh_t = hip.transpose_batched(w_hh)
    base = hip.RNN_F16 | (hip.RNN_XCD_SPLIT if xcd else 0)
    x_off, s_off = hip.dgrad16_published_offsets(num_steps, batch, hidden)
    x_bytes = (num_steps + 1) * 2 * batch * gh * 4
    s_bytes = (num_steps + 1) * 2 * 64 * 32 * 4

    def run(flags, cuts=None, alternate=None):
        db = torch.zeros(2 * gh, device=DEV)
        colmax = torch.zeros(2 * gh, dtype=torch.int32, device=DEV)
        dxw = torch.full((num_steps, batch, 2, gh), float('nan'), device=DEV)
        cuts = cuts or [num_steps, 0]
        for k, (hi, lo) in enumerate(zip(cuts[:-1], cuts[1:])):
            hip.rnn_bwd('lstm', dy, y, w_hh_t, reserve, None, dxw=dxw, dbias=db, workspace=ws,
                        steps=(lo, hi), flags=(alternate if alternate and k % 2 else flags),
                        colmax=colmax)
        hip.rnn_poll_error('lstm', ws, num_steps, batch, hidden)
        published = (ws[x_off:x_off + x_bytes].clone(), ws[s_off:s_off + s_bytes].clone())
        return dxw, db, colmax, published

    def row_err(got):
        err = (got.double() - ref).abs().amax(dim=(0, 2, 3))
        return float((err / ref.abs().amax(dim=(0, 2, 3)).clamp_min(1e-30)).max())

    stag = run(base | hip.RNN_STAGGER)
    want = run(base | hip.RNN_KPAIR)
    assert torch.isfinite(want[0]).all()
    e_pair, e_stag = row_err(want[0]), row_err(stag[0])
    assert e_pair < 1.5 * e_stag + 2e-7, (e_pair, e_stag)
    assert torch.equal(want[2].view(torch.float32), want[0].abs().amax(dim=(0, 1)).reshape(-1))
    assert float((want[1] - stag[1]).abs().max()) < 1e-5 * max(1.0, float(stag[1].abs().max()))
    for attempt in range(3):
        got = run(base | hip.RNN_KPAIR)
        assert torch.equal(got[0], want[0]), attempt
        assert torch.equal(got[2], want[2])
        assert torch.equal(got[3][0], want[3][0])
        assert torch.equal(got[3][1], want[3][1])
    if num_steps >= 7:
        cuts = [num_steps, num_steps - 1, num_steps - 3, num_steps // 2, 1, 0]
        for attempt in range(2):
            got = run(base | hip.RNN_KPAIR, cuts)
            assert torch.equal(got[0], want[0]), attempt
            assert torch.equal(got[2], want[2])
            assert torch.equal(got[3][0], want[3][0])
            assert torch.equal(got[3][1], want[3][1])
        # taking turns with the staggered kernel inside one pass: each reads what the other
        # published
        mixed = run(base | hip.RNN_KPAIR, cuts, alternate=base | hip.RNN_STAGGER)
        assert row_err(mixed[0]) < 1.5 * e_stag + 2e-7
        again = run(base | hip.RNN_KPAIR)
        assert torch.equal(again[0], want[0])
    # per-row lengths: the flag is ignored (one-barrier kernel)
    sl = torch.randint(1, num_steps + 1, (batch,), device=DEV, generator=g).int()
    y_l, reserve_l, ws_l = hip.rnn_fwd('lstm', xw, w_hh, sl)
    a = hip.rnn_bwd('lstm', dy, y_l, w_hh_t, reserve_l, sl, workspace=ws_l, flags=base)
    b = hip.rnn_bwd('lstm', dy, y_l, w_hh_t, reserve_l, sl, workspace=ws_l,
                    flags=base | hip.RNN_KPAIR)
    hip.rnn_poll_error('lstm', ws_l, num_steps, batch, hidden)
    assert torch.equal(a, b)


@pytest.mark.parametrize('use_len', [False, True])
@pytest.mark.parametrize('dims', [(12, 16), (9, 5), (7, 27), (5, 40)])
def test_rnn_bwd_k_pairs_at_2048(hip, use_len, dims):
    """CTCASR_RNN_KPAIR on the LSTM-2048 backward kernel (prnn_bwd16w_kernel<.., true>, round 6):
    the workgroups {s, s ^ 8} share one FULL N tile of 16 units, each multiplies one K half and
    hands the partner a [16 x 8] partial (tagged words through L2).  dxw against autograd through
    the float64 recurrence within the plain fp16 kernel's error, column maxima exact, step ranges
    (odd counts flip the tag parity) and repeated passes bit-identical, per-row lengths, batches
    of two and three 16-row tiles (sequential launches share the slots), kernels taking turns."""
    num_steps, batch = dims
    hidden, gh = 2048, 8192
    g = torch.Generator(device=DEV).manual_seed(53)
    xw = torch.randn(num_steps, batch, 2, gh, device=DEV, generator=g) * 0.5
    w_hh = torch.randn(2, gh, hidden, device=DEV, generator=g) / np.sqrt(hidden)
    dy = torch.randn(num_steps, batch, 2 * hidden, device=DEV, generator=g) * \
        torch.logspace(-6, 0, batch, device=DEV).view(1, batch, 1)
    sl = None
    if use_len:
        sl = torch.randint(1, num_steps + 1, (batch,), device=DEV, generator=g).int()
        sl[0] = num_steps
    xw64 = xw.double().requires_grad_(True)
    ref_y = _recurrence_float64('lstm', xw64, w_hh, None, sl)
    (ref_y * dy.double()).sum().backward()
    ref = xw64.grad
    y, reserve, ws = hip.rnn_fwd('lstm', xw, w_hh, sl)
    w_hh_t = hip.transpose_batched(w_hh)

    def run(flags, cuts=None, alternate=None):
        db = torch.zeros(2 * gh, device=DEV)
        colmax = torch.zeros(2 * gh, dtype=torch.int32, device=DEV)
        dxw = torch.full((num_steps, batch, 2, gh), float('nan'), device=DEV)
        cuts = cuts or [num_steps, 0]
        for k, (hi, lo) in enumerate(zip(cuts[:-1], cuts[1:])):
            hip.rnn_bwd('lstm', dy, y, w_hh_t, reserve, sl, dxw=dxw, dbias=db, workspace=ws,
                        steps=(lo, hi), flags=(alternate if alternate and k % 2 else flags),
                        colmax=colmax)
        hip.rnn_poll_error('lstm', ws, num_steps, batch, hidden)
        return dxw, db, colmax

    def row_err(got):
        got = torch.nan_to_num(got.double())       # (steps past a row's length: zero-filled / nan)
        err = (got - ref).abs().amax(dim=(0, 2, 3))
        return float((err / ref.abs().amax(dim=(0, 2, 3)).clamp_min(1e-30)).max())

    plain = run(hip.RNN_F16)
    want = run(hip.RNN_F16 | hip.RNN_KPAIR)
    e_pair, e_plain = row_err(want[0]), row_err(plain[0])
    assert e_pair < 1.5 * e_plain + 2e-7, (e_pair, e_plain)
    assert not torch.equal(want[0], plain[0])
    assert torch.equal(want[2].view(torch.float32),
                       torch.nan_to_num(want[0]).abs().amax(dim=(0, 1)).reshape(-1))
    assert float((want[1] - plain[1]).abs().max()) < 1e-5 * max(1.0, float(plain[1].abs().max()))
    for attempt in range(2):
        got = run(hip.RNN_F16 | hip.RNN_KPAIR)
        assert torch.equal(got[0], want[0]), attempt
        assert torch.equal(got[2], want[2])
    if num_steps >= 7:
        cuts = [num_steps, num_steps - 1, num_steps - 3, num_steps // 2, 1, 0]
        for attempt in range(2):
            got = run(hip.RNN_F16 | hip.RNN_KPAIR, cuts)
            assert torch.equal(got[0], want[0]), attempt
            assert torch.equal(got[2], want[2])
        mixed = run(hip.RNN_F16 | hip.RNN_KPAIR, cuts, alternate=hip.RNN_F16)
        assert row_err(mixed[0]) < 1.5 * e_plain + 2e-7
        again = run(hip.RNN_F16 | hip.RNN_KPAIR)
        assert torch.equal(again[0], want[0])


@pytest.mark.parametrize('xcd', [0, 1])
@pytest.mark.parametrize('batch', [17, 20, 24, 27, 32])
def test_staggered_tiles_read_nothing_of_the_pass_before(hip, xcd, batch):
    """Round 6: the kernels that stagger the two 16-row tiles publish tile 1's rows of a step after
    tile 0's rows of that step have been read; where the two share a cache line the reader's L1 /
    L2 keeps tile 1's bytes of the PASS BEFORE (same addresses) - invisible when every pass
    computes the same thing, which is what every other test of these kernels does.  Here: a pass
    over data A, then a pass over data B on the same workspace, against data B on a fresh
    workspace through the one-barrier kernel.  Staggered: bit for bit (batches that are not a
    multiple of 8 keep the one-barrier kernel: `prnn_bwd`); K pairs: to rounding.  (The K-pair
    kernel at 17 .. 20 rows failed this before the dispatch kept it to multiples of 8:
    tools/r06_stale_probe.py.)"""
    num_steps, hidden, gh = 40, 1024, 4096
    base = hip.RNN_F16 | (hip.RNN_XCD_SPLIT if xcd else 0)

    def data(seed):
        g = torch.Generator(device=DEV).manual_seed(seed)
        xw = torch.randn(num_steps, batch, 2, gh, device=DEV, generator=g) * 0.5
        w = torch.randn(2, gh, hidden, device=DEV, generator=g) / np.sqrt(hidden)
        dy = torch.randn(num_steps, batch, 2 * hidden, device=DEV, generator=g)
        return xw, w, hip.transpose_batched(w), dy

    for extra, exact in ((hip.RNN_STAGGER, True), (hip.RNN_KPAIR, False)):
        for rep in range(3):
            xa, wa, wta, dya = data(100 + rep)
            xb, wb, wtb, dyb = data(200 + rep)
            ya, ra, ws = hip.rnn_fwd('lstm', xa, wa)
            hip.rnn_bwd('lstm', dya, ya, wta, ra, workspace=ws, flags=base | extra)
            yb, rb, _ = hip.rnn_fwd('lstm', xb, wb, workspace=ws)
            got = hip.rnn_bwd('lstm', dyb, yb, wtb, rb, workspace=ws, flags=base | extra)
            hip.rnn_poll_error('lstm', ws, num_steps, batch, hidden)
            yf, rf, wsf = hip.rnn_fwd('lstm', xb, wb)
            want = hip.rnn_bwd('lstm', dyb, yf, wtb, rf, workspace=wsf, flags=base)
            hip.rnn_poll_error('lstm', wsf, num_steps, batch, hidden)
            if exact:
                assert torch.equal(got, want), (rep, int((got != want).sum()))
            else:
                assert float((got - want).abs().max()) < 2e-6 * float(want.abs().max()), rep


@pytest.mark.parametrize('cell,hidden', [('lstm', 1024), ('gru', 1024), ('rnn_relu', 2048),
                                         ('lstm', 2048)])
def test_no_persistent_kernel_reads_the_pass_before(hip, cell, hidden):
    """Every persistent recurrence variant, forward and backward: a pass over data A, then a pass
    over data B on the SAME workspace, equals data B on a fresh workspace bit for bit (same
    kernel).  Exchange blocks, scales, counters and hand-off slots live at the same addresses in
    every pass; a kernel that meets a cached copy of the pass before - see
    test_staggered_tiles_read_nothing_of_the_pass_before - only shows when the data changes."""
    num_steps = 24
    gates = hip.CELL_GATES[cell]

    def data(batch, seed):
        g = torch.Generator(device=DEV).manual_seed(seed)
        xw = torch.randn(num_steps, batch, 2, gates * hidden, device=DEV, generator=g) * 0.5
        w = torch.randn(2, gates * hidden, hidden, device=DEV, generator=g) / np.sqrt(hidden)
        dy = torch.randn(num_steps, batch, 2 * hidden, device=DEV, generator=g)
        bh = torch.randn(2, gates * hidden, device=DEV, generator=g) * 0.3 if cell == 'gru' else None
        return xw, w, hip.transpose_batched(w), dy, bh

    def both(d, ws, fwd_flags, bwd_flags):
        xw, w, wt, dy, bh = d
        y, res, ws = hip.rnn_fwd(cell, xw, w, workspace=ws, flags=fwd_flags, b_hh_n=bh)
        dxw = hip.rnn_bwd(cell, dy, y, wt, res, workspace=ws, flags=bwd_flags)
        hip.rnn_poll_error(cell, ws, num_steps, d[0].shape[1], hidden)
        return y, dxw, ws

    f16 = hip.RNN_F16 | hip.RNN_XCD_SPLIT
    variants = ((0, 0), (0, hip.RNN_WHOLE_CHIP), (hip.RNN_ONE_BARRIER, hip.RNN_ONE_BARRIER),
                (f16, f16 | hip.RNN_STAGGER), (f16, f16 | hip.RNN_KPAIR),
                (hip.RNN_F16 | hip.RNN_HALF_CHIP, hip.RNN_F16))
    for batch in (9, 17, 27, 32):
        if not hip.rnn_persistent_supported(cell, num_steps, batch, hidden):
            continue
        for fwd_flags, bwd_flags in variants:
            a, b = data(batch, 10 + batch), data(batch, 20 + batch)
            _, _, ws = both(a, None, fwd_flags, bwd_flags)
            y2, d2, _ = both(b, ws, fwd_flags, bwd_flags)
            y1, d1, _ = both(b, None, fwd_flags, bwd_flags)
            assert torch.equal(y1, y2), (batch, fwd_flags)
            assert torch.equal(d1, d2), (batch, bwd_flags)


def test_staggered_launch_leaves_at_once_when_the_time_out_word_is_set(hip):
    """The sticky time-out word ends a staggered-tile launch like every other persistent launch
    (nothing written, no spinning), and the pass after the poll is whole again."""
    import time
    num_steps, batch, hidden = 300, 32, 1024
    g = torch.Generator(device=DEV).manual_seed(6)
    xw = torch.randn(num_steps, batch, 2, 4 * hidden, device=DEV, generator=g) * 0.5
    w_hh = torch.randn(2, 4 * hidden, hidden, device=DEV, generator=g) / np.sqrt(hidden)
    dy = torch.randn(num_steps, batch, 2 * hidden, device=DEV, generator=g)
    flags = hip.RNN_F16 | hip.RNN_XCD_SPLIT | hip.RNN_STAGGER
    y, reserve, ws = hip.rnn_fwd('lstm', xw, w_hh)
    w_hh_t = hip.transpose_batched(w_hh)
    want = hip.rnn_bwd('lstm', dy, y, w_hh_t, reserve, workspace=ws, flags=flags)
    hip.rnn_poll_error('lstm', ws, num_steps, batch, hidden)
    address = hip.rnn_timeout_words('lstm', ws, num_steps, batch, hidden)[0]
    offset = address - ws.data_ptr()
    ws[offset:offset + 4].view(torch.int32).fill_(1)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    dxw = torch.full_like(want, 7.0)
    hip.rnn_bwd('lstm', dy, y, w_hh_t, reserve, dxw=dxw, workspace=ws, flags=flags)
    torch.cuda.synchronize()
    assert time.perf_counter() - t0 < 0.05
    assert bool((dxw == 7.0).all())
    with pytest.raises(hip.CtcAsrError, match='time'):
        hip.rnn_poll_error('lstm', ws, num_steps, batch, hidden)
    again = hip.rnn_bwd('lstm', dy, y, w_hh_t, reserve, workspace=ws, flags=flags)
    hip.rnn_poll_error('lstm', ws, num_steps, batch, hidden)
    assert torch.equal(again, want)


@pytest.mark.parametrize('magnitude', [1e-6, 3.0, 40.0, 3000.0])
def test_rnn_fwd_f16_scales_its_weights_itself(hip, magnitude):
    """No assumption about the size of W_hh: every workgroup scales its slice by the power of
    two that puts the slice's largest magnitude below fp16's range.  One unit's weights are
    `magnitude` times livelier than the rest (a weight of 40 or 3000 would be inf in fp16 under
    any fixed scale that keeps the precision of the others)."""
    num_steps, batch, hidden = 6, 16, 1024
    g = torch.Generator(device=DEV).manual_seed(32)
    xw = torch.randn(num_steps, batch, 2, 4 * hidden, device=DEV, generator=g) * 0.5
    w_hh = torch.randn(2, 4 * hidden, hidden, device=DEV, generator=g) / np.sqrt(hidden)
    w_hh[:, 5] *= magnitude
    w_hh[:, 2 * hidden + 77, 3] = magnitude
    ref = _recurrence_float64('lstm', xw, w_hh, None, None)
    y16, _, ws = hip.rnn_fwd('lstm', xw, w_hh, flags=hip.RNN_F16)
    y32, _, _ = hip.rnn_fwd('lstm', xw, w_hh)
    hip.rnn_poll_error('lstm', ws, num_steps, batch, hidden)
    assert torch.isfinite(y16).all()
    err16 = float((y16.double() - ref).abs().max())
    err32 = float((y32.double() - ref).abs().max())
    assert err16 < 3 * err32 + 2e-6, (magnitude, err16, err32)


@pytest.mark.parametrize('cell,dims', [('lstm', (12, 16, 1024)), ('lstm', (9, 32, 1024)),
                                       ('rnn_relu', (8, 16, 2048))])
def test_streaming_kernels_on_persistent_shapes(hip, cell, dims, monkeypatch):
    """CTCASR_RNN_MODE=stream forces the per-step kernels for shapes the persistent kernels
    cover: both paths must agree (y to 1e-5, dxw to 1e-4), with and without the whole-chip /
    half-chip flags of the persistent backward pass."""
    num_steps, batch, hidden = dims
    gates = onn.GATES[cell]
    rng = np.random.default_rng(17)
    xw = _t((rng.normal(size=(num_steps, batch, 2, gates * hidden)) * 0.5).astype(np.float32))
    w_hh = _t((rng.normal(size=(2, gates * hidden, hidden)) / np.sqrt(hidden))
              .astype(np.float32))
    dy = _t(rng.normal(size=(num_steps, batch, 2 * hidden)).astype(np.float32))
    w_hh_t = hip.transpose_batched(w_hh)
    assert hip.rnn_persistent_supported(cell, num_steps, batch, hidden)
    y_p, reserve_p, ws_p = hip.rnn_fwd(cell, xw, w_hh)
    dxw_p = hip.rnn_bwd(cell, dy, y_p, w_hh_t, reserve_p, workspace=ws_p)
    dxw_w = hip.rnn_bwd(cell, dy, y_p, w_hh_t, reserve_p, workspace=ws_p,
                        flags=hip.RNN_WHOLE_CHIP)
    hip.rnn_poll_error(cell, ws_p, num_steps, batch, hidden)
    monkeypatch.setenv('CTCASR_RNN_MODE', 'stream')
    assert not hip.rnn_persistent_supported(cell, num_steps, batch, hidden)
    y_s, reserve_s, ws_s = hip.rnn_fwd(cell, xw, w_hh)
    # (the plain cells differentiate through y itself: hand both backward passes the same y, or
    # a ReLU output that is 0 on one path and 1e-7 on the other flips a whole gradient entry)
    dxw_s = hip.rnn_bwd(cell, dy, y_p if cell != 'lstm' else y_s, w_hh_t, reserve_s,
                        workspace=ws_s)
    torch.cuda.synchronize()
    monkeypatch.delenv('CTCASR_RNN_MODE')
    assert float((y_s - y_p).abs().max()) < 1e-5
    assert float((dxw_s - dxw_p).abs().max()) < 1e-4
    assert float((dxw_w - dxw_p).abs().max()) < 1e-4


@pytest.mark.parametrize('cell,dims', [('lstm', (7, 16, 1024)), ('lstm', (7, 27, 1024)),
                                       ('gru', (6, 5, 128)), ('rnn_tanh', (6, 20, 2048)),
                                       ('rnn_relu', (5, 3, 64))])
def test_rnn_fwd_adds_the_input_projection_bias_in_the_kernel(hip, cell, dims):
    """``xw_bias`` (b_W + b_R where it commutes) added inside the recurrence kernels - persistent
    and streaming - equals a forward pass over xw + bias."""
    num_steps, batch, hidden = dims
    gates = onn.GATES[cell]
    rng = np.random.default_rng(29)
    xw = (rng.normal(size=(num_steps, batch, 2, gates * hidden)) * 0.5).astype(np.float32)
    bias = (rng.normal(size=(2, gates * hidden)) * 0.5).astype(np.float32)
    w_hh = _t((rng.normal(size=(2, gates * hidden, hidden)) / np.sqrt(hidden)).astype(np.float32))
    b_hh = _t((rng.normal(size=(2, gates * hidden)) * 0.3).astype(np.float32)) \
        if cell == 'gru' else None
    y_ref, _, ws = hip.rnn_fwd(cell, _t(xw + bias[None, None]), w_hh, b_hh_n=b_hh)
    y_got, _, _ = hip.rnn_fwd(cell, _t(xw), w_hh, b_hh_n=b_hh, xw_bias=_t(bias.reshape(-1)),
                              workspace=ws)
    hip.rnn_poll_error(cell, ws, num_steps, batch, hidden)
    assert float((y_got - y_ref).abs().max()) < 1e-6


def test_rnn_timeout_word_is_sticky_until_polled(hip):
    """A time-out raised by ANY persistent launch must survive later launches on the same
    workspace (other layers, the other pass) until `rnn_poll_error` reads it - round-1 finding:
    every launch used to clear it.  The word is poked by hand here (a real time-out needs a
    starved GPU): later launches must leave it alone, the poll must raise once and clear it."""
    num_steps, batch, hidden = 6, 4, 1024
    rng = np.random.default_rng(23)
    xw = _t((rng.normal(size=(num_steps, batch, 2, 4 * hidden)) * 0.5).astype(np.float32))
    w_hh = _t((rng.normal(size=(2, 4 * hidden, hidden)) / 32).astype(np.float32))
    y, reserve, ws = hip.rnn_fwd('lstm', xw, w_hh)
    hip.rnn_poll_error('lstm', ws, num_steps, batch, hidden)
    state = (6 * batch * hidden * 4 + 255) // 256 * 256
    error_word = state + 2 * 2 * 8 * 64 * 4 + 2 * 2 * 64 * 4   # SyncWords: counters, done, `error`
    ws[error_word:error_word + 4] = torch.tensor([1, 0, 0, 0], dtype=torch.uint8, device=DEV)
    hip.rnn_fwd('lstm', xw, w_hh, y=y, reserve=reserve, workspace=ws)            # another "layer"
    dy = torch.ones_like(y)
    hip.rnn_bwd('lstm', dy, y, hip.transpose_batched(w_hh), reserve, workspace=ws)  # other pass
    with pytest.raises(hip.CtcAsrError, match='timed out'):
        hip.rnn_poll_error('lstm', ws, num_steps, batch, hidden)
    hip.rnn_poll_error('lstm', ws, num_steps, batch, hidden)        # cleared by the read


def test_bias_act_and_colsum(hip):
    rng = np.random.default_rng(9)
    for rows, cols in [(33, 29), (100, 64), (257, 96), (1001, 32), (77, 16), (50, 8), (9, 2)]:
        z = (rng.normal(size=(rows, cols)) * 15).astype(np.float32)
        bias = rng.normal(size=cols).astype(np.float32)
        y = hip.bias_act_fwd(_t(z), _t(bias), 20.0)
        ref = np.minimum(np.maximum(z + bias, 0), 20.0)
        assert np.abs(y.cpu().numpy() - ref).max() < 1e-6
        dy = rng.normal(size=(rows, cols)).astype(np.float32)
        dbias = torch.zeros(cols, device=DEV)
        dz = hip.bias_act_bwd(y, _t(dy), 20.0, 0.0, dbias)
        ref_dz = dy * ((ref > 0) & (ref < 20.0))
        assert np.abs(dz.cpu().numpy() - ref_dz).max() < 1e-6
        assert np.abs(dbias.cpu().numpy() - ref_dz.sum(0)).max() < 1e-3
        # plain bias add (logits layer) + standalone column sum
        y2 = hip.bias_act_fwd(_t(z), _t(bias), 0.0)
        assert np.abs(y2.cpu().numpy() - (z + bias)).max() < 1e-6
        db2 = torch.zeros(cols, device=DEV)
        hip.colsum_accumulate(_t(dy), db2)
        assert np.abs(db2.cpu().numpy() - dy.sum(0)).max() < 1e-3


def test_bias_act_dropout_statistics(hip):
    rows, cols, rate = 512, 256, 0.1
    z = torch.full((rows, cols), 5.0, device=DEV)
    y = hip.bias_act_fwd(z.clone(), torch.zeros(cols, device=DEV), 20.0, rate, seed=1234)
    kept = (y > 0).float().mean().item()
    assert abs(kept - (1 - rate)) < 0.01
    assert torch.allclose(y[y > 0], torch.tensor(5.0 / (1 - rate), device=DEV))
    y2 = hip.bias_act_fwd(z.clone(), torch.zeros(cols, device=DEV), 20.0, rate, seed=1234)
    assert torch.equal(y, y2)                       # same seed -> same mask
    dz = hip.bias_act_bwd(y, torch.ones_like(y), 20.0, rate)
    assert torch.allclose(dz, (y > 0).float() / (1 - rate))


def test_adam_tf_form(hip):
    rng = np.random.default_rng(2)
    n = 1003
    p = rng.normal(size=n).astype(np.float32)
    m = np.zeros(n, np.float32)
    v = np.zeros(n, np.float32)
    dp, dm, dv = _t(p), _t(m), _t(v)
    ref_p, ref_m, ref_v = p.astype(np.float64), m.astype(np.float64), v.astype(np.float64)
    for step in range(1, 4):
        g = rng.normal(size=n).astype(np.float32)
        hip.adam_step(dp, _t(g), dm, dv, step, lr=1e-3, grad_scale=0.5)
        ref_p, ref_m, ref_v = onn.adam_step(ref_p, 0.5 * g.astype(np.float64), ref_m, ref_v,
                                            step, lr=1e-3)
    assert np.abs(dp.cpu().numpy() - ref_p).max() < 1e-6
    assert np.abs(dv.cpu().numpy() - ref_v).max() < 1e-6


@pytest.mark.parametrize('feature_type', ['mel', 'mfcc'])
@pytest.mark.parametrize('norm', ['none', 'local', 'local_scalar'])
def test_features_match_oracle(hip, feature_type, norm):
    from oracle import features as ofeat
    rng = np.random.default_rng(21)
    lengths = np.array([16000, 9000, 401, 12345], dtype=np.int32)
    pcm = np.zeros((4, 16000), dtype=np.int16)
    for b, n in enumerate(lengths):
        t = np.arange(n) / 16000.0
        tone = 4000 * np.sin(2 * np.pi * (200 + 150 * b) * t) * np.exp(-2 * t)
        pcm[b, :n] = np.clip(rng.normal(size=n) * 1500 + tone, -32768, 32767).astype(np.int16)
    for drop in (False, True):
        out, out_len = hip.features(_t(pcm, torch.int16), _t(lengths, torch.int32), feature_type,
                                    norm, drop)
        out, out_len = out.cpu().numpy(), out_len.cpu().numpy()
        for b, n in enumerate(lengths):
            if n == 401 and norm != 'none':
                continue     # 2 frames: constant-column / tiny-sample statistics, not a parity case
            ref, ref_len = ofeat.load_sample_from_pcm(pcm[b, :n], 16000, feature_type, norm, drop)
            assert out_len[b] == int(ref_len)
            got = out[b, :out_len[b]]
            # bar: 1e-3 (north_star, fp32); the kernel keeps float64 until the final cast
            assert np.abs(got - ref).max() < 1e-3, (b, np.abs(got - ref).max())
            assert np.abs(out[b, out_len[b]:]).max(initial=0.0) == 0.0


@pytest.mark.parametrize('norm', ['max', 'log_softmax'])
def test_beam_search_matches_tensorflow_style_oracle(hip, norm):
    rng = np.random.default_rng(31)
    logits = (rng.normal(size=(40, 6, 29)) * 2).astype(np.float32)
    logits[:, :, -1] += 2.0
    seq_len = np.array([40, 17, 1, 33, 40, 25], dtype=np.int32)
    for width in (1, 2, 3, 8, 64, 200):
        out, out_len, logp = hip.ctc_beam_decode(_t(logits), _t(seq_len, torch.int32), width,
                                                 normalization=norm)
        ref_paths, ref_logp = cref.beam_search_decode(logits, seq_len, width, normalization=norm)
        out, out_len, logp = out.cpu().numpy(), out_len.cpu().numpy(), logp.cpu().numpy()
        for b in range(len(seq_len)):
            assert out[b, :out_len[b]].tolist() == ref_paths[b], (width, b)
            assert (out[b, out_len[b]:] == 0).all()
        assert np.allclose(logp, ref_logp, atol=1e-4)


def test_beam_search_full_size_default_width(hip):
    """T'=500, 29 classes, beam 1024 (reference default) and 64 (BASELINE config 5)."""
    rng = np.random.default_rng(32)
    logits = (rng.normal(size=(500, 3, 29)) * 3).astype(np.float32)
    logits[:, :, -1] += 3.0
    seq_len = np.array([500, 431, 500], dtype=np.int32)
    for width in (64, 1024):
        out, out_len, logp = hip.ctc_beam_decode(_t(logits), _t(seq_len, torch.int32), width)
        ref_paths, ref_logp = cref.beam_search_decode(logits, seq_len, width)
        out, out_len = out.cpu().numpy(), out_len.cpu().numpy()
        for b in range(3):
            assert out[b, :out_len[b]].tolist() == ref_paths[b], (width, b)
        assert np.allclose(logp.cpu().numpy(), ref_logp, rtol=1e-5, atol=1e-2)


def test_beam_search_known_answer(hip):
    import json
    import os
    case = json.load(open(os.path.join(os.path.dirname(__file__), 'golden',
                                       'ctc_kat.json')))['decode_case']
    logits = np.log(np.array(case['probs'])).astype(np.float32)[:, None, :]
    sl = _t(np.array([5]), torch.int32)
    out, n, _ = hip.ctc_beam_decode(_t(logits), sl, 64)
    assert out[0, :int(n[0])].tolist() == case['beam_wide']
    out, n, _ = hip.ctc_beam_decode(_t(logits), sl, 2)
    assert out[0, :int(n[0])].tolist() == case['beam_width_2']


def test_dropout_kernel(hip):
    x = torch.ones(100000, device=DEV) * 2.0
    y = hip.dropout(x, 0.25, seed=77)
    kept = (y != 0)
    assert abs(kept.float().mean().item() - 0.75) < 0.01
    assert torch.allclose(y[kept], torch.tensor(2.0 / 0.75, device=DEV))
    g = hip.dropout(torch.ones_like(x), 0.25, seed=77)       # backward: same mask
    assert torch.equal(g != 0, kept)
    assert not torch.equal(hip.dropout(x, 0.25, seed=78) != 0, kept)


@pytest.mark.parametrize('layer', [(40, 32), (20, 96)])
@pytest.mark.parametrize('shape', [(1, 1), (2, 7), (3, 37), (1, 500)])
@pytest.mark.parametrize('weight_size', [0.05, 30.0, 1e-5])
def test_conv_s12_forward_on_the_fp16_matrix_pipe(hip, shape, layer, weight_size):
    """`ctcasr_conv_s12_fwd16` (csrc/conv16.hip): the forward convolution with input and weights as
    two fp16 pieces each.  Input in [0, 20] like the clipped ReLU in front of it (scale 2^11), any
    weight magnitude (the scale comes from the kernel's own maximum, found on the device).
    Against fp64 `conv2d` next to the fp32-MFMA kernel on the same operands: not further away
    than 3 x its error (measured: closer); fused epilogue and time-major output as the fp32
    kernel's; an input beyond the bound saturates to a finite result."""
    from ctc_asr_amd import split_gemm
    batch, frames = shape
    freq, cout = layer
    rng = np.random.default_rng(frames + cout)
    x_np = np.clip(rng.normal(size=(batch, frames, freq, 32)) * 6.0, 0.0, 20.0).astype(np.float32)
    x_np[rng.random(x_np.shape) < 0.3] *= 1e-4            # small values beside the clipped ones
    weight = (rng.normal(size=(cout, 32, 11, 21)) * weight_size).astype(np.float32)
    bias = (rng.normal(size=cout) * weight_size).astype(np.float32)
    scale = split_gemm.f16_scale(20.0)
    packed16 = hip.conv_s12_pack_weights16(_t(weight))
    packed = hip.conv_s12_pack_weights(_t(weight))
    y16 = hip.conv_s12_fwd16(_t(x_np), scale, packed16, cout, _t(bias))
    y32 = hip.conv_s12_fwd(_t(x_np), packed, cout, _t(bias))
    x = torch.zeros(batch, 32, frames + 10, freq + 19, dtype=torch.float64)
    x[:, :, 5:5 + frames, 9:9 + freq] = torch.tensor(x_np, dtype=torch.float64).permute(0, 3, 1, 2)
    ref = torch.nn.functional.conv2d(x, torch.tensor(weight, dtype=torch.float64),
                                     torch.tensor(bias, dtype=torch.float64), stride=(1, 2)) \
        .permute(0, 2, 3, 1).numpy()
    top = max(np.abs(ref).max(), 1e-30)
    e16 = np.abs(y16.cpu().numpy() - ref).max() / top
    e32 = np.abs(y32.cpu().numpy() - ref).max() / top
    assert e16 < 3 * e32 + 1e-7, (e16, e32)
    assert e16 < 2e-5, e16
    assert not torch.equal(y16, y32)
    cut = float(0.3 * np.abs(ref).max())
    y_act = hip.conv_s12_fwd16(_t(x_np), scale, packed16, cout, _t(bias), relu_cutoff=cut)
    assert torch.equal(y_act, torch.clamp(y16, 0.0, cut))
    y_tm = hip.conv_s12_fwd16(_t(x_np), scale, packed16, cout, _t(bias), time_major=True)
    assert torch.equal(y_tm.permute(1, 0, 2, 3), y16)
    # re-packing into the same buffer after the weights changed (what a training step does)
    again = hip.conv_s12_pack_weights16(_t(weight * 2), packed16)
    assert again.data_ptr() == packed16.data_ptr()
    y_twice = hip.conv_s12_fwd16(_t(x_np), scale, again, cout)
    assert float((y_twice - 2 * (y16 - _t(bias))).abs().max()) <= 4e-6 * top
    big = hip.conv_s12_fwd16(_t(x_np * 100.0), scale, again, cout)
    assert torch.isfinite(big).all()


@pytest.mark.parametrize('layer', [(40, 32), (20, 96)])
@pytest.mark.parametrize('shape', [(1, 1), (2, 7), (3, 37), (1, 500)])
def test_conv_s12_data_gradient_on_the_fp16_matrix_pipe(hip, shape, layer):
    """`ctcasr_conv_s12_bwd_data16`: dz over eight decades across frames and utterances (a
    power-of-two scale per dz frame of a workgroup's patch, found while it is staged), zero cells,
    the fused epilogue mask and the time-major layout; against fp64 autograd next to the fp32-MFMA
    kernel: per (utterance, frame) the error relative to that frame's largest gradient is not
    above 3 x the fp32 kernel's."""
    batch, frames = shape
    freq, cout = layer
    rng = np.random.default_rng(3 * frames + cout)
    x_np = rng.normal(size=(batch, frames, freq, 32)).astype(np.float32)
    dz = rng.normal(size=(batch, frames, freq // 2, cout))
    dz *= 10.0 ** rng.uniform(-8, 0, size=(batch, frames, 1, 1))
    dz[rng.random(dz.shape[:3]) < 0.1] = 0.0
    dz = dz.astype(np.float32)
    weight = (rng.normal(size=(cout, 32, 11, 21)) * 0.05).astype(np.float32)
    packed = hip.conv_s12_pack_weights(_t(weight))
    packed16 = hip.conv_s12_pack_weights16(_t(weight))
    dx32 = hip.conv_s12_bwd_data(_t(dz), packed)
    dx16 = hip.conv_s12_bwd_data16(_t(dz), packed16)
    x = torch.zeros(batch, 32, frames + 10, freq + 19, dtype=torch.float64, requires_grad=True)
    y = torch.nn.functional.conv2d(x, torch.tensor(weight, dtype=torch.float64), stride=(1, 2))
    y.backward(torch.tensor(dz, dtype=torch.float64).permute(0, 3, 1, 2))
    ref = x.grad[:, :, 5:5 + frames, 9:9 + freq].permute(0, 2, 3, 1)
    top = ref.abs().amax(dim=(2, 3)).clamp_min(1e-300)

    def frame_err(got):
        return float(((got.double().cpu() - ref).abs().amax(dim=(2, 3)) / top).max())
    e16, e32 = frame_err(dx16), frame_err(dx32)
    assert e16 < 3 * e32 + 1e-6, (e16, e32)
    assert not torch.equal(dx16, dx32)
    dz_tm = _t(dz).permute(1, 0, 2, 3).contiguous()
    assert torch.equal(hip.conv_s12_bwd_data16(dz_tm, packed16, time_major=True), dx16)
    act = _t(rng.uniform(-1.0, 2.5, size=dz.shape).astype(np.float32))
    masked = _t(dz) * ((act > 0) & (act < 1.5)).float()
    assert torch.equal(hip.conv_s12_bwd_data16(_t(dz), packed16, act=act, relu_cutoff=1.5),
                       hip.conv_s12_bwd_data16(masked, packed16))


@pytest.mark.parametrize('layer', [(40, 32), (20, 96)])
@pytest.mark.parametrize('shape', [(1, 1), (2, 7), (9, 37), (16, 12), (3, 500)])
def test_conv_s12_kernel_gradient_on_the_fp16_matrix_pipe(hip, shape, layer):
    """`ctcasr_conv_s12_wrw16`: x in [0, 20] behind the clipped ReLU (many exact zeros), dz over
    six decades across output channels and three across frames (one power-of-two scale per output
    channel, found on the device), batches that do not fill a block of 8 utterances; against fp64
    autograd next to the fp32-MFMA kernel: per output channel the error relative to the channel's
    largest gradient is not above 3 x the fp32 kernel's; bias gradient, mask and time-major
    layout as in `ctcasr_conv_s12_wrw`."""
    batch, frames = shape
    freq, cout = layer
    rng = np.random.default_rng(5 * frames + cout + batch)
    x_np = np.clip(rng.normal(size=(batch, frames, freq, 32)) * 6.0, 0.0, 20.0).astype(np.float32)
    dz = rng.normal(size=(batch, frames, freq // 2, cout))
    dz *= 10.0 ** rng.uniform(-6, 0, size=(1, 1, 1, cout))
    dz *= 10.0 ** rng.uniform(-3, 0, size=(batch, frames, 1, 1))
    dz[rng.random(dz.shape[:3]) < 0.1] = 0.0
    dz = dz.astype(np.float32)
    x_scale = 2.0 ** 11                       # 20 x 2^11 < 65504
    dw32 = hip.conv_s12_wrw(_t(dz), _t(x_np))
    db16 = torch.zeros(cout, device='cuda')
    dw16 = hip.conv_s12_wrw16(_t(dz), _t(x_np), x_scale, dbias=db16)
    x = torch.zeros(batch, 32, frames + 10, freq + 19, dtype=torch.float64)
    x[:, :, 5:5 + frames, 9:9 + freq] = torch.tensor(x_np, dtype=torch.float64).permute(0, 3, 1, 2)
    w64 = torch.zeros(cout, 32, 11, 21, dtype=torch.float64, requires_grad=True)
    y = torch.nn.functional.conv2d(x, w64, stride=(1, 2))
    y.backward(torch.tensor(dz, dtype=torch.float64).permute(0, 3, 1, 2))
    ref = w64.grad
    top = ref.abs().amax(dim=(1, 2, 3)).clamp_min(1e-300)

    def channel_err(got):
        return float(((got.double().cpu() - ref).abs().amax(dim=(1, 2, 3)) / top).max())
    e16, e32 = channel_err(dw16), channel_err(dw32)
    assert e16 < 3 * e32 + 1e-6, (e16, e32)
    assert not torch.equal(dw16, dw32)
    want_db = dz.astype(np.float64).sum(axis=(0, 1, 2))
    assert np.abs(db16.cpu().numpy() - want_db).max() < 1e-4 * max(1e-30, np.abs(want_db).max())
    dz_tm = _t(dz).permute(1, 0, 2, 3).contiguous()
    assert torch.equal(hip.conv_s12_wrw16(dz_tm, _t(x_np), x_scale, time_major=True), dw16)
    act = _t(rng.uniform(-1.0, 2.5, size=dz.shape).astype(np.float32))
    masked = _t(dz) * ((act > 0) & (act < 1.5)).float()
    db_mask, db_plain = torch.zeros(cout, device='cuda'), torch.zeros(cout, device='cuda')
    got = hip.conv_s12_wrw16(_t(dz), _t(x_np), x_scale, act=act, relu_cutoff=1.5, dbias=db_mask)
    assert torch.equal(got, hip.conv_s12_wrw16(masked, _t(x_np), x_scale, dbias=db_plain))
    assert float((db_mask - db_plain).abs().max()) <= 1e-5 * float(db_plain.abs().max() + 1e-30)


@pytest.mark.parametrize('layer', [(40, 32), (20, 96)])
@pytest.mark.parametrize('shape', [(1, 1), (2, 7), (3, 16), (2, 37), (3, 65), (1, 500)])
def test_conv_s12_kernels_match_the_library_convolution(hip, shape, layer):
    """Implicit-GEMM forward, data gradient and kernel gradient of the 11x21 / stride (1,2) layers (32 -> 32 channels
    on 40 frequencies, 32 -> 96 on 20) against torch's convolution on the explicitly SAME-padded
    input (pad 5/5 in time, 9/10 in frequency), fp64 on the CPU."""
    batch, frames = shape
    freq, cout = layer
    assert hip.conv_s12_supported(freq, cout) and not hip.conv_s12_supported(freq, 48)
    rng = np.random.default_rng(frames + cout)
    x_np = rng.normal(size=(batch, frames, freq, 32)).astype(np.float32)
    dz = rng.normal(size=(batch, frames, freq // 2, cout)).astype(np.float32)
    weight = (rng.normal(size=(cout, 32, 11, 21)) * 0.05).astype(np.float32)
    bias = rng.normal(size=cout).astype(np.float32)
    packed = hip.conv_s12_pack_weights(_t(weight))
    y_gpu = hip.conv_s12_fwd(_t(x_np), packed, cout, _t(bias)).cpu().numpy()
    y_nobias = hip.conv_s12_fwd(_t(x_np), packed, cout).cpu().numpy()
    dx = hip.conv_s12_bwd_data(_t(dz), packed).cpu().numpy()
    x = torch.zeros(batch, 32, frames + 10, freq + 19, dtype=torch.float64)
    x[:, :, 5:5 + frames, 9:9 + freq] = torch.tensor(x_np, dtype=torch.float64).permute(0, 3, 1, 2)
    x.requires_grad_(True)
    y = torch.nn.functional.conv2d(x, torch.tensor(weight, dtype=torch.float64),
                                   torch.tensor(bias, dtype=torch.float64), stride=(1, 2))
    assert tuple(y.shape) == (batch, cout, frames, freq // 2)
    ref_y = y.detach().permute(0, 2, 3, 1).numpy()
    assert np.abs(y_gpu - ref_y).max() < 2e-4 * max(1.0, np.abs(ref_y).max())
    assert np.abs(y_nobias + bias - ref_y).max() < 2e-4 * max(1.0, np.abs(ref_y).max())
    w64 = torch.tensor(weight, dtype=torch.float64, requires_grad=True)
    y = torch.nn.functional.conv2d(x, w64, torch.tensor(bias, dtype=torch.float64), stride=(1, 2))
    y.backward(torch.tensor(dz, dtype=torch.float64).permute(0, 3, 1, 2))
    ref = x.grad[:, :, 5:5 + frames, 9:9 + freq].permute(0, 2, 3, 1).numpy()
    assert dx.shape == ref.shape
    assert np.abs(dx - ref).max() < 2e-4 * max(1.0, np.abs(ref).max())
    # fused epilogue (ReLU + min(., cutoff)) and the time-major variants the last layer of the
    # stack uses: same numbers, other layout
    y_act = hip.conv_s12_fwd(_t(x_np), packed, cout, _t(bias), relu_cutoff=1.5).cpu().numpy()
    assert np.abs(y_act - np.minimum(np.maximum(y_gpu, 0.0), 1.5)).max() < 1e-6
    y_tm = hip.conv_s12_fwd(_t(x_np), packed, cout, _t(bias), time_major=True)
    assert tuple(y_tm.shape) == (frames, batch, freq // 2, cout)
    assert np.array_equal(y_tm.permute(1, 0, 2, 3).cpu().numpy(), y_gpu)
    dz_tm = _t(dz).permute(1, 0, 2, 3).contiguous()
    assert np.array_equal(hip.conv_s12_bwd_data(dz_tm, packed, time_major=True).cpu().numpy(), dx)
    assert torch.equal(hip.conv_s12_wrw(dz_tm, _t(x_np), time_major=True),
                       hip.conv_s12_wrw(_t(dz), _t(x_np)))
    # kernel gradient (split over (b, t) tiles, two-stage reduction): deterministic
    dw = hip.conv_s12_wrw(_t(dz), _t(x_np))
    dw_again = hip.conv_s12_wrw(_t(dz), _t(x_np))
    assert torch.equal(dw, dw_again)
    ref_w = w64.grad.numpy()
    assert tuple(dw.shape) == ref_w.shape
    assert np.abs(dw.cpu().numpy() - ref_w).max() < 2e-4 * max(1.0, np.abs(ref_w).max())
    # backward of the fused epilogue inside the gradient kernels (`act` = the stored output of
    # min(max(., 0), cutoff), dz = the gradient w.r.t. that output): identical to masking dz
    # first (`bias_act_bwd`), and the bias gradient comes out of the kernel-gradient kernel
    act = _t(y_act)
    dbias_ref = torch.zeros(cout, device=DEV)
    masked = hip.bias_act_bwd(act, _t(dz), 1.5, 0.0, dbias_ref)
    assert 0.05 < float((masked != 0).float().mean()) < 0.95
    dbias = torch.zeros(cout, device=DEV)
    assert torch.equal(hip.conv_s12_wrw(_t(dz), _t(x_np), act=act, relu_cutoff=1.5, dbias=dbias),
                       hip.conv_s12_wrw(masked, _t(x_np)))
    assert float((dbias - dbias_ref).abs().max()) < 1e-4 * max(1.0, float(dbias_ref.abs().max()))
    assert torch.equal(hip.conv_s12_bwd_data(_t(dz), packed, act=act, relu_cutoff=1.5),
                       hip.conv_s12_bwd_data(masked, packed))
    act_tm = act.permute(1, 0, 2, 3).contiguous()
    dbias_tm = torch.zeros(cout, device=DEV)
    assert torch.equal(hip.conv_s12_wrw(dz_tm, _t(x_np), time_major=True, act=act_tm,
                                        relu_cutoff=1.5, dbias=dbias_tm),
                       hip.conv_s12_wrw(masked, _t(x_np)))
    assert float((dbias_tm - dbias_ref).abs().max()) < 1e-4 * max(1.0, float(dbias_ref.abs().max()))
    assert torch.equal(hip.conv_s12_bwd_data(dz_tm, packed, time_major=True, act=act_tm,
                                             relu_cutoff=1.5),
                       hip.conv_s12_bwd_data(masked, packed))


@pytest.mark.parametrize('shape', [(1, 1), (2, 9), (2, 10), (3, 64), (2, 131), (1, 999), (1, 1000)])
def test_conv0_fwd_matches_the_library_convolution(hip, shape):
    """First DS2 convolution (1 -> 32 channels, 11x41, stride (2,2), TensorFlow SAME padding: 5/5
    or 4/5 frames depending on the parity of T, 19/20 frequencies) against torch's conv2d, fp64."""
    from ctc_asr_amd.model import same_padding
    batch, frames = shape
    rng = np.random.default_rng(frames)
    x_np = rng.normal(size=(batch, frames, 80)).astype(np.float32)
    weight = (rng.normal(size=(32, 1, 11, 41)) * 0.1).astype(np.float32)
    bias = rng.normal(size=32).astype(np.float32)
    y_gpu = hip.conv0_fwd(_t(x_np), _t(weight), _t(bias)).cpu().numpy()
    y_act = hip.conv0_fwd(_t(x_np), _t(weight), _t(bias), relu_cutoff=0.7).cpu().numpy()
    assert np.abs(y_act - np.minimum(np.maximum(y_gpu, 0.0), 0.7)).max() < 1e-6
    t_out, pt0, pt1 = same_padding(frames, 11, 2)
    f_out, pf0, pf1 = same_padding(80, 41, 2)
    assert (f_out, pf0, pf1) == (40, 19, 20)
    x = torch.nn.functional.pad(torch.tensor(x_np, dtype=torch.float64).unsqueeze(1),
                                (pf0, pf1, pt0, pt1))
    y = torch.nn.functional.conv2d(x, torch.tensor(weight, dtype=torch.float64),
                                   torch.tensor(bias, dtype=torch.float64), stride=(2, 2))
    ref = y.permute(0, 2, 3, 1).numpy()
    assert y_gpu.shape == ref.shape == (batch, t_out, 40, 32)
    assert np.abs(y_gpu - ref).max() < 2e-4 * max(1.0, np.abs(ref).max())
    # kernel gradient of the same layer
    dz = rng.normal(size=(batch, t_out, 40, 32)).astype(np.float32)
    dw = hip.conv0_wrw(_t(dz), _t(x_np)).cpu().numpy()
    w64 = torch.tensor(weight, dtype=torch.float64, requires_grad=True)
    torch.nn.functional.conv2d(x, w64, stride=(2, 2)).backward(
        torch.tensor(dz, dtype=torch.float64).permute(0, 3, 1, 2))
    ref_dw = w64.grad.numpy()
    assert dw.shape == ref_dw.shape
    assert np.abs(dw - ref_dw).max() < 2e-4 * max(1.0, np.abs(ref_dw).max())
    # backward of the fused epilogue inside the kernel (mask from the stored output + bias sums)
    act = _t(y_act)
    dbias_ref = torch.zeros(32, device=DEV)
    masked = hip.bias_act_bwd(act, _t(dz), 0.7, 0.0, dbias_ref)
    dbias = torch.zeros(32, device=DEV)
    assert torch.equal(hip.conv0_wrw(_t(dz), _t(x_np), act=act, relu_cutoff=0.7, dbias=dbias),
                       hip.conv0_wrw(masked, _t(x_np)))
    assert float((dbias - dbias_ref).abs().max()) < 1e-4 * max(1.0, float(dbias_ref.abs().max()))


@pytest.mark.parametrize('shape', [(1, 1), (2, 6), (3, 33), (9, 64), (16, 47), (2, 999)])
def test_conv0_on_the_fp16_matrix_pipe(hip, shape):
    """`ctcasr_conv0_fwd16` / `ctcasr_conv0_wrw16` against fp64 next to the fp32-MFMA kernels:
    features whose magnitude varies over four decades between utterances and frames (the forward
    kernel scales every workgroup's patch on its own), dz over six decades across channels.
    Forward: per (utterance, frame) the error relative to the frame's largest output is not above
    3 x the fp32 kernel's; kernel gradient: per output channel likewise."""
    from ctc_asr_amd.model import same_padding
    batch, frames = shape
    rng = np.random.default_rng(11 * frames + batch)
    x_np = rng.normal(size=(batch, frames, 80))
    x_np *= 10.0 ** rng.uniform(-4, 0.5, size=(batch, 1, 1))
    x_np *= 10.0 ** rng.uniform(-1, 0, size=(batch, frames, 1))
    x_np = x_np.astype(np.float32)
    weight = (rng.normal(size=(32, 1, 11, 41)) * 0.1).astype(np.float32)
    bias = (rng.normal(size=32) * 1e-3).astype(np.float32)
    packed16 = hip.conv0_pack_weights16(_t(weight))
    y32 = hip.conv0_fwd(_t(x_np), _t(weight), _t(bias))
    y16 = hip.conv0_fwd16(_t(x_np), packed16, _t(bias))
    y_act = hip.conv0_fwd16(_t(x_np), packed16, _t(bias), relu_cutoff=0.7)
    assert float((y_act - y16.clamp(0.0, 0.7)).abs().max()) < 1e-6
    t_out, pt0, pt1 = same_padding(frames, 11, 2)
    x = torch.nn.functional.pad(torch.tensor(x_np, dtype=torch.float64).unsqueeze(1),
                                (19, 20, pt0, pt1))
    ref = torch.nn.functional.conv2d(x, torch.tensor(weight, dtype=torch.float64),
                                     torch.tensor(bias, dtype=torch.float64),
                                     stride=(2, 2)).permute(0, 2, 3, 1)
    assert tuple(y16.shape) == tuple(ref.shape) == (batch, t_out, 40, 32)
    top = ref.abs().amax(dim=(2, 3)).clamp_min(1e-30)

    def frame_err(got):
        return float(((got.double().cpu() - ref).abs().amax(dim=(2, 3)) / top).max())
    e16, e32 = frame_err(y16), frame_err(y32)
    assert e16 < 3 * e32 + 1e-6, (e16, e32)
    assert not torch.equal(y16, y32)

    dz = rng.normal(size=(batch, t_out, 40, 32))
    dz *= 10.0 ** rng.uniform(-6, 0, size=(1, 1, 1, 32))
    dz[rng.random(dz.shape[:3]) < 0.1] = 0.0
    dz = dz.astype(np.float32)
    dw32 = hip.conv0_wrw(_t(dz), _t(x_np))
    db16 = torch.zeros(32, device=DEV)
    dw16 = hip.conv0_wrw16(_t(dz), _t(x_np), dbias=db16)
    w64 = torch.tensor(weight, dtype=torch.float64, requires_grad=True)
    torch.nn.functional.conv2d(x, w64, stride=(2, 2)).backward(
        torch.tensor(dz, dtype=torch.float64).permute(0, 3, 1, 2))
    ref_dw = w64.grad
    top_c = ref_dw.abs().amax(dim=(1, 2, 3)).clamp_min(1e-300)

    def channel_err(got):
        return float(((got.double().cpu() - ref_dw).abs().amax(dim=(1, 2, 3)) / top_c).max())
    c16, c32 = channel_err(dw16), channel_err(dw32)
    assert c16 < 3 * c32 + 1e-6, (c16, c32)
    want_db = dz.astype(np.float64).sum(axis=(0, 1, 2))
    assert np.abs(db16.cpu().numpy() - want_db).max() < 1e-4 * max(1e-30, np.abs(want_db).max())
    act = _t(rng.uniform(-0.5, 1.0, size=dz.shape).astype(np.float32))
    masked = _t(dz) * ((act > 0) & (act < 0.7)).float()
    assert torch.equal(hip.conv0_wrw16(_t(dz), _t(x_np), act=act, relu_cutoff=0.7),
                       hip.conv0_wrw16(masked, _t(x_np)))


def test_beam_search_random_sweep_against_the_oracle(hip):
    """The register-resident beam set of round 2 against the C oracle over a sweep of shapes:
    few and many classes, widths around the 64-lane and 256 / 1024 register-tile boundaries,
    flat and peaked logits, ragged lengths.  (Continuous random logits: no exact ties.)"""
    rng = np.random.default_rng(2024)
    cases = 0
    for classes in (5, 29, 40):
        for width in (1, 7, 63, 64, 65, 255, 256, 257, 700):
            steps = int(rng.integers(8, 40))
            batch = int(rng.integers(1, 5))
            scale = float(rng.choice([0.3, 1.0, 3.0]))
            logits = (rng.normal(size=(steps, batch, classes)) * scale).astype(np.float32)
            logits[:, :, -1] += float(rng.choice([0.0, 2.0, 4.0]))
            seq_len = rng.integers(1, steps + 1, size=batch).astype(np.int32)
            seq_len[0] = steps
            norm = 'max' if (cases % 2 == 0) else 'log_softmax'
            out, out_len, logp = hip.ctc_beam_decode(_t(logits), _t(seq_len, torch.int32), width,
                                                     normalization=norm)
            ref_paths, ref_logp = cref.beam_search_decode(logits, seq_len, width,
                                                          normalization=norm)
            out, out_len = out.cpu().numpy(), out_len.cpu().numpy()
            for b in range(batch):
                assert out[b, :out_len[b]].tolist() == ref_paths[b], (classes, width, steps, b)
            assert np.allclose(logp.cpu().numpy(), ref_logp, rtol=1e-5, atol=1e-3)
            cases += 1
    assert cases == 27


@pytest.mark.parametrize('batch,lengths,num_steps', [
    (3, False, 37), (16, True, 37), (19, True, 37), (32, False, 37), (32, True, 37),
    (45, True, 37),            # two blocks of rows (32 + 13)
    (19, False, 1), (19, True, 2), (33, False, 3)])    # degenerate passes; a block of ONE row
def test_reduce_scatter_backward_equals_the_all_gather_kernels(hip, batch, lengths, num_steps):
    """`RNN_REDUCE_SCATTER`: the LSTM-1024 backward recurrence with the product dgates x R cut
    along K (every workgroup multiplies the dgates of its OWN units into a partial dh for all
    units, consumers sum 64 partial tiles) against the default kernels (all-gather of dgates,
    themselves checked against float64 autograd in `test_rnn_fwd_bwd`): same results to fp32
    summation order, deterministic, bit-identical when cut into step ranges; one batch tile
    (one chain) and two (two chains per workgroup), with and without per-row lengths."""
    hidden = 1024
    gen = torch.Generator(device=DEV).manual_seed(100 + batch)
    xw = torch.randn(num_steps, batch, 2, 4 * hidden, device=DEV, generator=gen) * 0.5
    w_hh = torch.randn(2, 4 * hidden, hidden, device=DEV, generator=gen) / 32
    dy = torch.randn(num_steps, batch, 2 * hidden, device=DEV, generator=gen)
    seq_len = None
    if lengths:
        seq_len = torch.randint(1, num_steps + 1, (batch,), device=DEV, generator=gen,
                                dtype=torch.int32)
        seq_len[0] = num_steps
        seq_len[batch - 1] = 1
    assert hip.rnn_persistent_supported('lstm', num_steps, batch, hidden)
    w_hh_t = hip.transpose_batched(w_hh)
    y, reserve, ws = hip.rnn_fwd('lstm', xw, w_hh, seq_len)
    ref = hip.rnn_bwd('lstm', dy, y, w_hh_t, reserve, seq_len, workspace=ws)
    dbias = torch.zeros(2 * 4 * hidden, device=DEV)
    got = hip.rnn_bwd('lstm', dy, y, w_hh_t, reserve, seq_len, workspace=ws, dbias=dbias,
                      flags=hip.RNN_REDUCE_SCATTER)
    want_db = got.double().sum(dim=(0, 1)).reshape(-1)
    assert float((dbias.double() - want_db).abs().max()) < 1e-4 * max(1.0, float(want_db.abs().max()))
    again = hip.rnn_bwd('lstm', dy, y, w_hh_t, reserve, seq_len, workspace=ws,
                        flags=hip.RNN_REDUCE_SCATTER)
    hip.rnn_poll_error('lstm', ws, num_steps, batch, hidden)
    assert torch.equal(got, again)
    scale = max(1.0, float(ref.abs().max()))
    assert float((got - ref).abs().max()) < 2e-5 * scale
    if seq_len is not None:       # steps past a row's length carry no gradient
        assert float(got[1:, batch - 1].abs().max()) == 0.0
    cut = torch.empty_like(got)
    marks = sorted({num_steps, num_steps * 20 // 37, num_steps * 19 // 37, num_steps * 7 // 37, 0},
                   reverse=True)
    for hi, lo in zip(marks[:-1], marks[1:]):
        hip.rnn_bwd('lstm', dy, y, w_hh_t, reserve, seq_len, dxw=cut, workspace=ws,
                    steps=(lo, hi), flags=hip.RNN_REDUCE_SCATTER)
    hip.rnn_poll_error('lstm', ws, num_steps, batch, hidden)
    assert torch.equal(cut, got)


@pytest.mark.parametrize('flags', ['RNN_DEFAULT', 'RNN_F16'])
def test_a_set_time_out_word_ends_every_later_launch_at_once(hip, flags):
    """The persistent kernels' time-out word is sticky until the host polls it.  A launch that
    finds it set returns immediately (round 5): whatever it would compute is invalid anyway, and a
    wedged barrier at N > 1 must not turn the rest of a benchmark leg into minutes of bounded
    spinning.  The poll reports the time-out, clears the word, and the next launch is whole."""
    import time
    num_steps, batch, hidden = 300, 16, 1024
    flags = getattr(hip, flags)
    g = torch.Generator(device=DEV).manual_seed(5)
    xw = torch.randn(num_steps, batch, 2, 4 * hidden, device=DEV, generator=g) * 0.5
    w_hh = torch.randn(2, 4 * hidden, hidden, device=DEV, generator=g) / np.sqrt(hidden)
    dy = torch.randn(num_steps, batch, 2 * hidden, device=DEV, generator=g)
    y, reserve, ws = hip.rnn_fwd('lstm', xw, w_hh, flags=flags)
    hip.rnn_poll_error('lstm', ws, num_steps, batch, hidden)
    address = hip.rnn_timeout_words('lstm', ws, num_steps, batch, hidden)[0]
    offset = address - ws.data_ptr()
    ws[offset:offset + 4].view(torch.int32).fill_(1)
    w_hh_t = hip.transpose_batched(w_hh)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    y2 = torch.full_like(y, 7.0)
    hip.rnn_fwd('lstm', xw, w_hh, y=y2, reserve=reserve.clone(), workspace=ws, flags=flags)
    dxw = torch.full((num_steps, batch, 2, 4 * hidden), 7.0, device=DEV)
    hip.rnn_bwd('lstm', dy, y, w_hh_t, reserve, dxw=dxw, workspace=ws, flags=flags)
    torch.cuda.synchronize()
    assert time.perf_counter() - t0 < 0.05          # (a whole pass takes ~1 ms per 300 steps)
    assert bool((y2 == 7.0).all()) and bool((dxw == 7.0).all())      # nothing ran
    with pytest.raises(hip.CtcAsrError, match='time'):
        hip.rnn_poll_error('lstm', ws, num_steps, batch, hidden)
    y3, _, _ = hip.rnn_fwd('lstm', xw, w_hh, workspace=ws, flags=flags)
    hip.rnn_poll_error('lstm', ws, num_steps, batch, hidden)
    assert torch.equal(y3, y)


@pytest.mark.parametrize('dims', [(12, 16, 2048), (40, 5, 2048), (9, 11, 2048)])
def test_relu_recurrence_on_the_fp16_matrix_pipe(hip, dims):
    """CTCASR_RNN_F16 for the ReLU cell at H = 2048 - the reference's default model
    (asr/params.py:43-50) - forward and backward (`prnn_relu16_kernel`): h and dpre have no bound, so
    both are published as two fp16 pieces under a power of two per (producer workgroup, row).
    Against the float64 recurrence / autograd next to the fp32 kernels' errors, with a recurrent
    matrix that lets h grow over the steps and gradients over five decades of rows; bias in the
    kernel; bias gradient; column maxima exact; step ranges bit-identical; per-row lengths and
    bigger batches fall back to the fp32 kernels."""
    num_steps, batch, hidden = dims
    g = torch.Generator(device=DEV).manual_seed(51)
    xw = torch.randn(num_steps, batch, 2, hidden, device=DEV, generator=g)
    w_hh = torch.randn(2, hidden, hidden, device=DEV, generator=g) * (1.3 / np.sqrt(hidden))
    bias = torch.randn(2 * hidden, device=DEV, generator=g) * 0.3
    dy = torch.randn(num_steps, batch, 2 * hidden, device=DEV, generator=g) * \
        torch.logspace(-5, 0, batch, device=DEV).view(1, batch, 1)
    assert hip.rnn_f16_recurrence('rnn_relu', num_steps, batch, hidden, hip.RNN_F16)
    assert not hip.rnn_f16_recurrence('rnn_relu', num_steps, batch, hidden, hip.RNN_F16, ragged=True)
    assert not hip.rnn_f16_recurrence('rnn_relu', num_steps, 20, hidden, hip.RNN_F16)
    assert not hip.rnn_fwd_f16_supported('rnn_relu', num_steps, batch, hidden)     # (no y_pieces)
    xb = (xw + bias.view(1, 1, 2, hidden)).double().requires_grad_(True)
    ref_y = _recurrence_float64('rnn_relu', xb, w_hh, None, None)
    (ref_y * dy.double()).sum().backward()
    ref_dxw = xb.grad
    y16, reserve, ws = hip.rnn_fwd('rnn_relu', xw, w_hh, xw_bias=bias, flags=hip.RNN_F16)
    y32, reserve32, _ = hip.rnn_fwd('rnn_relu', xw, w_hh, xw_bias=bias, workspace=ws)
    hip.rnn_poll_error('rnn_relu', ws, num_steps, batch, hidden)
    assert not torch.equal(y16, y32)
    scale = float(ref_y.abs().max())
    e16, e32 = float((y16.double() - ref_y).abs().max()), float((y32.double() - ref_y).abs().max())
    assert e16 < 3 * e32 + 1e-6 * scale, (e16, e32, scale)
    # backward from the SAME y (the fp32 kernel's) so that the masks y > 0 agree
    w_hh_t = hip.transpose_batched(w_hh)
    db16, db32 = torch.zeros(2 * hidden, device=DEV), torch.zeros(2 * hidden, device=DEV)
    colmax = torch.zeros(2 * hidden, dtype=torch.int32, device=DEV)
    dxw32 = hip.rnn_bwd('rnn_relu', dy, y32, w_hh_t, reserve32, dbias=db32, workspace=ws)
    dxw16 = hip.rnn_bwd('rnn_relu', dy, y32, w_hh_t, reserve32, dbias=db16, workspace=ws,
                        flags=hip.RNN_F16, colmax=colmax)
    hip.rnn_poll_error('rnn_relu', ws, num_steps, batch, hidden)
    assert not torch.equal(dxw16, dxw32)
    # (the float64 reference's mask comes from its own y: compare where the masks agree)
    same = ((ref_y > 0) == (y32.double() > 0)).view(num_steps, batch, 2, hidden)

    def row_err(got):
        err = ((got.double() - ref_dxw).abs() * same).amax(dim=(0, 2, 3))
        return float((err / ref_dxw.abs().amax(dim=(0, 2, 3)).clamp_min(1e-30)).max())
    assert row_err(dxw16) < 3 * row_err(dxw32) + 1e-6, (row_err(dxw16), row_err(dxw32))
    assert float((db16 - db32).abs().max()) < 1e-4 * max(1.0, float(db32.abs().max()))
    assert torch.equal(colmax.view(torch.float32), dxw16.abs().amax(dim=(0, 1)).reshape(-1))
    if num_steps >= 9:
        cuts = [0, 3, num_steps // 2, num_steps]
        y_cut = torch.full_like(y16, float('nan'))
        for lo, hi in zip(cuts[:-1], cuts[1:]):
            hip.rnn_fwd('rnn_relu', xw, w_hh, y=y_cut, reserve=reserve, workspace=ws,
                        steps=(lo, hi), xw_bias=bias, flags=hip.RNN_F16)
        assert torch.equal(y_cut, y16)
        dxw_cut = torch.full_like(dxw16, float('nan'))
        db_cut, colmax_cut = torch.zeros_like(db16), torch.zeros_like(colmax)
        for hi, lo in zip(cuts[::-1][:-1], cuts[::-1][1:]):
            hip.rnn_bwd('rnn_relu', dy, y32, w_hh_t, reserve32, dxw=dxw_cut, dbias=db_cut,
                        workspace=ws, steps=(lo, hi), flags=hip.RNN_F16, colmax=colmax_cut)
        hip.rnn_poll_error('rnn_relu', ws, num_steps, batch, hidden)
        assert torch.equal(dxw_cut, dxw16) and torch.equal(colmax_cut, colmax)
    # per-row lengths: the fp32 kernel runs, whatever the flag says
    sl = torch.full((batch,), num_steps, dtype=torch.int32, device=DEV)
    sl[-1] = max(1, num_steps - 2)
    y_len, _, _ = hip.rnn_fwd('rnn_relu', xw, w_hh, sl, xw_bias=bias, workspace=ws,
                               flags=hip.RNN_F16)
    y_len32, _, _ = hip.rnn_fwd('rnn_relu', xw, w_hh, sl, xw_bias=bias, workspace=ws)
    assert torch.equal(y_len, y_len32)
