"""The block-scaled data-gradient kernel (csrc/dgrad16.hip, ABI v6): dx = dxw . W_ih read from what
the fp16-pipe backward recurrence published - the gradient of the cuDNN input projection
(asr/model.py:194-215).  Checked (a) against float64 next to the library's fp32 GEMM on operands
published by a torch restatement of the producer's arithmetic (any shape, gradients over many
decades), at the C3 layer shape 16000 x 8192 -> 2048; (b) behind the REAL recurrence kernel against
float64 `dxw @ W_ih` of the dxw that kernel wrote; (c) step ranges / directions / accumulate."""

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = 'cuda'
H = 1024


@pytest.fixture(scope='module')
def hip():
    if not torch.cuda.is_available():
        pytest.skip('needs the MI355X')
    from ctc_asr_amd import hip as hip_mod
    hip_mod.load()
    return hip_mod


def publish(hip, dxw, workspace=None):
    """Write dxw [T, B, 2, 4H] into a recurrence workspace the way prnn_bwd16_kernel publishes its
    dgates (rnn_persistent.hip): per (step, dir, producer = 16 units x 4 gates) and row the power
    of two that puts the row's largest of the 64 values into [2^13, 2^14), two fp16 pieces,
    [step][dir][P][half m][piece][k group q][b][e = 4 (unit & 1) + gate]; inverse scales
    [step][dir][P][32 rows].  Step s = time s (dir 0) / T - 1 - s (dir 1)."""
    steps, batch = dxw.shape[:2]
    if workspace is None:
        workspace = hip.rnn_workspace('lstm', steps, batch, H, dxw.device)
    x_off, s_off = hip.dgrad16_published_offsets(steps, batch, H)
    # (t, b, dir, gate, P, m, q, u1): unit = 16 P + 8 m + 2 q + u1
    d = dxw.view(steps, batch, 2, 4, 64, 2, 4, 2)
    top = d.abs().amax(dim=(3, 5, 6, 7))                          # (t, b, dir, P)
    expo = (torch.frexp(top)[1] - 1).float()                     # floor(log2(top)), exactly
    expo = torch.where(top > 0, (13 - expo).clamp(-100, 100), torch.zeros_like(expo))
    scale = torch.exp2(expo)
    scaled = d * scale.view(steps, batch, 2, 1, 64, 1, 1, 1)
    h1 = scaled.half()
    h2 = (scaled - h1.float()).half()
    pieces = torch.stack([h1, h2], dim=0)                         # (piece, t, b, dir, gate, P, m, q, u1)
    # -> (t, dir, P, m, piece, q, b, u1, gate)
    pieces = pieces.permute(1, 3, 5, 6, 0, 7, 2, 8, 4).contiguous()
    pieces[:, 1] = pieces[:, 1].flip(0)                           # dir 1: step s = T - 1 - t
    block = 2 * batch * 4 * H * 4                                 # bytes per step
    workspace[x_off + block:x_off + block * (steps + 1)] = pieces.view(torch.uint8).view(-1)
    inv = torch.zeros(steps, 2, 64, 32, device=dxw.device)
    inv[..., :batch] = (1.0 / scale).permute(0, 2, 3, 1)
    inv[:, 1] = inv[:, 1].flip(0)
    workspace[s_off:s_off + inv.numel() * 4] = inv.view(torch.uint8).view(-1)
    return workspace


def rel_errors(got, ref):
    """(rms relative error, largest row error relative to the row's largest |ref|)."""
    err = got.double() - ref
    rms = float(err.norm() / ref.norm())
    row = float((err.abs().amax(dim=1) / ref.abs().amax(dim=1).clamp_min(1e-300)).max())
    return rms, row


def decades(steps, batch, gen):
    """Gradients like a real step's: frames over eight decades, gate units over three, utterances
    over two."""
    dxw = torch.randn(steps, batch, 2, 4 * H, device=DEV, generator=gen)
    dxw *= torch.logspace(-8, 0, steps, device=DEV)[torch.randperm(steps, device=DEV,
                                                                    generator=gen)].view(-1, 1, 1, 1)
    dxw *= torch.logspace(-3, 0, 4 * H, device=DEV)[torch.randperm(4 * H, device=DEV,
                                                                   generator=gen)].view(1, 1, 1, -1)
    dxw *= torch.logspace(-2, 0, batch, device=DEV).view(1, -1, 1, 1)
    return dxw


@pytest.mark.parametrize('steps,batch,n', [(500, 32, 2048), (37, 32, 640), (50, 16, 2048),
                                           (21, 20, 304), (9, 5, 2048), (3, 1, 256)])
def test_against_float64_next_to_the_fp32_gemm(hip, steps, batch, n):
    gen = torch.Generator(device=DEV).manual_seed(7 + steps)
    dxw = decades(steps, batch, gen)
    w = torch.randn(8 * H, n, device=DEV, generator=gen) / np.sqrt(n)
    w[5, 3] = 11.0                                               # (in range: |w| <= 14.6)
    ws = publish(hip, dxw)
    packed = hip.dgrad16_pack_weights(w, H, 2048.0)
    got = hip.dgrad16_blockscaled(ws, steps, batch, H, packed, 2048.0, n)
    d2 = dxw.view(steps * batch, 8 * H)
    ref = d2.double() @ w.double()
    lib = torch.mm(d2, w)
    rms, row = rel_errors(got, ref)
    rms32, row32 = rel_errors(lib, ref)
    assert torch.isfinite(got).all()
    # fp32-grade: two pieces are 22 bits relative to a (row, 64-column block) maximum
    assert rms < 2.0 * rms32 + 1e-7 and row < 3.0 * row32 + 1e-7, (rms, rms32, row, row32)


def test_step_ranges_directions_and_accumulate(hip):
    steps, batch, n = 45, 32, 512
    gen = torch.Generator(device=DEV).manual_seed(3)
    dxw = decades(steps, batch, gen)
    w = torch.randn(8 * H, n, device=DEV, generator=gen) / np.sqrt(n)
    ws = publish(hip, dxw)
    packed = hip.dgrad16_pack_weights(w, H, 2048.0)
    whole = hip.dgrad16_blockscaled(ws, steps, batch, H, packed, 2048.0, n)
    # rows [lo, hi) only: the other rows are not touched
    part = torch.full((steps * batch, n), 7.0, device=DEV)
    hip.dgrad16_blockscaled(ws, steps, batch, H, packed, 2048.0, n, out=part, steps=(11, 30))
    assert torch.equal(part[11 * batch:30 * batch], whole[11 * batch:30 * batch])
    assert bool((part[:11 * batch] == 7.0).all()) and bool((part[30 * batch:] == 7.0).all())
    # one direction's gate columns, then the other's added: the two halves of the sum
    halves = hip.dgrad16_blockscaled(ws, steps, batch, H, packed, 2048.0, n, dirs=(0, 1))
    ref0 = dxw[:, :, 0].reshape(steps * batch, 4 * H).double() @ w[:4 * H].double()
    assert rel_errors(halves, ref0)[0] < 2e-6
    hip.dgrad16_blockscaled(ws, steps, batch, H, packed, 2048.0, n, out=halves, dirs=(1, 2),
                            accumulate=True)
    assert float((halves - whole).abs().max()) <= 1e-5 * float(whole.abs().max())
    with pytest.raises(hip.CtcAsrError):
        hip.dgrad16_blockscaled(ws, steps, batch, H, packed, 2048.0, n, steps=(5, 5))


@pytest.mark.parametrize('steps,batch', [(40, 32), (25, 16), (12, 20)])
def test_behind_the_recurrence_kernel(hip, steps, batch):
    """The operands as prnn_bwd16_kernel itself leaves them (whole pass and in step ranges): dx
    against float64 `dxw @ W_ih` of the dxw the same launch wrote; and the torch restatement of
    the producer used above reproduces the kernel's exchange blocks value for value."""
    gen = torch.Generator(device=DEV).manual_seed(11)
    n = 2 * H
    xw = torch.randn(steps, batch, 2, 4 * H, device=DEV, generator=gen) * 0.5
    w_hh = torch.randn(2, 4 * H, H, device=DEV, generator=gen) / np.sqrt(H)
    dy = torch.randn(steps, batch, 2 * H, device=DEV, generator=gen) * \
        torch.logspace(-5, 0, batch, device=DEV).view(1, batch, 1)
    w_ih = torch.randn(8 * H, n, device=DEV, generator=gen) / np.sqrt(n)
    y, reserve, ws = hip.rnn_fwd('lstm', xw, w_hh, flags=hip.RNN_F16)
    w_hh_t = hip.transpose_batched(w_hh)
    packed = hip.dgrad16_pack_weights(w_ih, H, 2048.0)
    for cuts in ([steps, 0], [steps, steps // 2, 0]):
        dxw = torch.zeros(steps, batch, 2, 4 * H, device=DEV)
        for hi, lo in zip(cuts[:-1], cuts[1:]):
            hip.rnn_bwd('lstm', dy, y, w_hh_t, reserve, dxw=dxw, workspace=ws, steps=(lo, hi),
                        flags=hip.RNN_F16)
        hip.rnn_poll_error('lstm', ws, steps, batch, H)
        got = hip.dgrad16_blockscaled(ws, steps, batch, H, packed, 2048.0, n)
        ref = dxw.view(steps * batch, 8 * H).double() @ w_ih.double()
        lib = torch.mm(dxw.view(steps * batch, 8 * H), w_ih)
        rms, row = rel_errors(got, ref)
        rms32, row32 = rel_errors(lib, ref)
        assert rms < 2.0 * rms32 + 1e-7 and row < 3.0 * row32 + 1e-7, (rms, rms32, row, row32)
    x_off, s_off = hip.dgrad16_published_offsets(steps, batch, H)
    block = 2 * batch * 4 * H * 4
    mine = publish(hip, dxw)
    # (as values: where a residual is zero torch leaves -0.0, the kernel +0.0)
    assert torch.equal(mine[x_off + block:x_off + block * (steps + 1)].view(torch.float16),
                       ws[x_off + block:x_off + block * (steps + 1)].view(torch.float16))
    # (the kernel also stores the inverse scales - 1.0 - of the rows a 16-row tile has past the batch)
    count = steps * 2 * 64 * 32
    scales = [t[s_off:s_off + count * 4].view(torch.float32).view(-1, 32)[:, :batch]
              for t in (mine, ws)]
    assert torch.equal(scales[0], scales[1])


def test_weight_pieces_saturate_and_cover_ragged_column_counts(hip):
    n = 300
    w = torch.zeros(8 * H, n, device=DEV)
    w[0, 0], w[1, 299], w[8 * H - 1, 17] = 1e9, -1e9, 3.0
    packed = hip.dgrad16_pack_weights(w, H, 2048.0)
    assert packed.numel() == 2 * 128 * 19 * 2048
    assert torch.isfinite(packed.view(torch.float16).float()).all()
    assert float(packed.view(torch.float16).float().abs().max()) <= 60000.0


@pytest.mark.parametrize('batch,frames', [(32, 399), (16, 329)])
def test_the_model_takes_the_kernel_and_its_schedules_agree(hip, batch, frames):
    """`CTCModel.backward` with the own data-gradient kernel (the default for LSTM-1024) against
    the library form (CTCASR_OWN_DGRAD=0) on the same weights and batch - every gradient slice -
    and the kernel's optional schedules (`dgrad_early`: a finished direction's share of a step
    range multiplied beside the next recurrence launch, on the side stream / a stream of its own)
    against the plain one; T' odd in the second case (the two directions' ranges overlap by a row)."""
    from ctc_asr_amd.model import CTCModel, ModelConfig
    cfg = ModelConfig(used_model='ds2', conv_filters=(32, 32), num_units_dense=2048,
                      num_layers_rnn=2, num_units_rnn=H, rnn_cell='lstm', cudnn=True,
                      dense_dropout_rate=0.0, conv_dropout_rate=0.0)
    rng = np.random.default_rng(3)
    feats = torch.tensor(rng.normal(size=(batch, frames, 80)).astype(np.float32), device=DEV)
    flen = torch.full((batch,), frames, dtype=torch.int32)
    labels = [list(rng.integers(1, 28, size=20)) for _ in range(batch)]

    def grads(own, early):
        model = CTCModel(cfg, DEV, seed=5)
        model.own_dgrad, model.dgrad_early = own, early
        loss = model.forward_backward(feats, flen, labels)
        torch.cuda.synchronize()
        model.check_rnn_error()
        return float(loss), model.arena.grad.clone(), model

    loss_lib, grad_lib, _ = grads(False, '0')
    loss_own, grad_own, model = grads(True, '0')
    arith = model.arithmetic()
    assert arith['rnn1/data_gradient'].startswith('fp16x3 block-scaled')
    assert arith['rnn0/data_gradient'].startswith('fp16x3 block-scaled')
    assert loss_lib == loss_own                      # (the forward pass is the same)
    for name, a, b in model.arena.layer_slices:
        ref = grad_lib[a:b].double()
        rel = float((ref - grad_own[a:b].double()).norm() / ref.norm().clamp_min(1e-30))
        assert rel < 2e-5, (name, rel)               # both are fp32-grade forms of one product
    for early in ('side', 'own'):
        _, grad_e, _ = grads(True, early)
        for name, a, b in model.arena.layer_slices:
            ref = grad_own[a:b].double()
            rel = float((ref - grad_e[a:b].double()).norm() / ref.norm().clamp_min(1e-30))
            assert rel < 2e-6, (early, name, rel)    # the same products, summed in another order
