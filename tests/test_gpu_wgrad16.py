"""The own weight-gradient kernel (csrc/wgrad16.hip, ABI v6): dW_x += D^T X, dW_y += D^T Y over
the rows of a step range - the gradients of a cuDNN layer's two matrices (asr/model.py:194-215) -
from operands packed transposed as fp16 pieces.  Against float64 next to the library's fp32 GEMM."""

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = 'cuda'


@pytest.fixture(scope='module')
def hip():
    if not torch.cuda.is_available():
        pytest.skip('needs the MI355X')
    from ctc_asr_amd import hip as hip_mod
    hip_mod.load()
    return hip_mod


def own_wgrad(hip, d, x, x_scale, y=None, y_shift=0, y_scale=1.0, row_lo=0, row_hi=None, parts=1):
    """dW_x, dW_y of rows [row_lo, row_hi) of d against the same rows of x and rows shifted by
    ``y_shift`` of y (zeros outside), through ctcasr_wgrad16_pack / _gemm."""
    row_hi = d.shape[0] if row_hi is None else row_hi
    n_rows = row_hi - row_lo
    stages = (n_rows + 31) // 32
    scale, inv = hip.colmax_scale(d[row_lo:row_hi])
    d_pk = hip.wgrad16_pack(d[row_lo:row_hi], n_rows, 0, stages, 1.0, col_scale=scale)
    x_pk = hip.wgrad16_pack(x[row_lo:], min(n_rows, x.shape[0] - row_lo), 0, stages, x_scale)
    dw_x = torch.zeros(d.shape[1], x.shape[1], device=DEV)
    dw_y = y_pk = None
    if y is not None:
        # rows [row_lo + shift, ...) of y; rows outside [0, len(y)) read as zeros
        y_pk = hip.wgrad16_pack(y, y.shape[0], row_lo + y_shift, stages, y_scale)
        dw_y = torch.zeros(d.shape[1], y.shape[1], device=DEV)
    hip.wgrad16_gemm(d_pk, d.shape[1], stages, inv, x_pk, 0, x_scale, dw_x,
                     y_packed=y_pk, y_stage0=0, y_scale=y_scale, dw_y=dw_y, parts=parts)
    return dw_x, dw_y


def errors(got, ref):
    err = got.double() - ref
    return float(err.norm() / ref.norm()), \
        float((err.abs().amax(dim=1) / ref.abs().amax(dim=1).clamp_min(1e-300)).max())


@pytest.mark.parametrize('rows,m,nx,ny,shift', [(8000, 4096, 2048, 1024, -32), (1456, 512, 640, 256, 16),
                                                (100, 300, 48, 272, -16), (32, 16, 16, 16, 16)])
def test_against_float64_next_to_the_fp32_gemm(hip, rows, m, nx, ny, shift):
    g = torch.Generator(device=DEV).manual_seed(rows)
    d = torch.randn(rows, m, device=DEV, generator=g)
    d *= torch.logspace(-9, -3, m, device=DEV)[torch.randperm(m, device=DEV, generator=g)]
    d *= torch.logspace(-3, 0, rows, device=DEV).view(-1, 1)
    x = torch.rand(rows, nx, device=DEV, generator=g) * 2 - 1          # |x| <= 1: scale 2^15
    y = torch.rand(rows, ny, device=DEV, generator=g) * 2 - 1
    dw_x, dw_y = own_wgrad(hip, d, x, 32768.0, y, shift, 32768.0)
    ref_x = d.double().t() @ x.double()
    y_sh = torch.zeros_like(y)
    if shift < 0:
        y_sh[-shift:] = y[:shift]
    else:
        y_sh[:rows - shift] = y[shift:]
    ref_y = d.double().t() @ y_sh.double()
    for got, ref, lib in ((dw_x, ref_x, torch.mm(d.t(), x)), (dw_y, ref_y, torch.mm(d.t(), y_sh))):
        rms, row = errors(got, ref)
        rms32, row32 = errors(lib, ref)
        assert torch.isfinite(got).all()
        assert rms < 2.0 * rms32 + 1e-7 and row < 3.0 * row32 + 1e-7, (rms, rms32, row, row32)


def test_ranges_accumulate_and_only_one_operand(hip):
    g = torch.Generator(device=DEV).manual_seed(4)
    rows, m, nx = 640, 256, 512
    d = torch.randn(rows, m, device=DEV, generator=g) * 1e-4
    x = torch.rand(rows, nx, device=DEV, generator=g) * 40 - 20        # clipped-ReLU range: 2^11
    whole, none = own_wgrad(hip, d, x, 2048.0)
    assert none is None
    a, _ = own_wgrad(hip, d, x, 2048.0, row_lo=0, row_hi=352)
    b, _ = own_wgrad(hip, d, x, 2048.0, row_lo=352, row_hi=640)
    ref = d.double().t() @ x.double()
    assert errors(a + b, ref)[0] < 2e-6 and errors(whole, ref)[0] < 2e-6
    # the kernel accumulates into what is there
    scale, inv = hip.colmax_scale(d)
    d_pk = hip.wgrad16_pack(d, rows, 0, 20, 1.0, col_scale=scale)
    x_pk = hip.wgrad16_pack(x, rows, 0, 20, 2048.0)
    out = torch.full((m, nx), 3.0, device=DEV)
    hip.wgrad16_gemm(d_pk, m, 20, inv, x_pk, 0, 2048.0, out)
    assert errors(out - 3.0, ref)[0] < 1e-5
    # a stage offset into a second operand packed for all rows
    part = torch.zeros(m, nx, device=DEV)
    d2_pk = hip.wgrad16_pack(d[320:], 320, 0, 10, 1.0, col_scale=scale)
    hip.wgrad16_gemm(d2_pk, m, 10, inv, x_pk, 10, 2048.0, part)
    assert errors(part, d[320:].double().t() @ x[320:].double())[0] < 2e-6


def test_parts_add_in_order(hip):
    """A tile's row sum cut into workgroups: the same bits on every run, the float64 answer, and
    the sync words back at zero."""
    g = torch.Generator(device=DEV).manual_seed(9)
    rows, m, nx, ny = 4000, 1024, 768, 512
    d = torch.randn(rows, m, device=DEV, generator=g) * 1e-3
    x = torch.rand(rows, nx, device=DEV, generator=g) * 2 - 1
    y = torch.rand(rows, ny, device=DEV, generator=g) * 2 - 1
    ref = d.double().t() @ x.double()
    one = own_wgrad(hip, d, x, 32768.0, y, 0, 32768.0)
    for parts in (2, 5, 8, 1000):
        runs = [own_wgrad(hip, d, x, 32768.0, y, 0, 32768.0, parts=parts) for _ in range(3)]
        for dw_x, dw_y in runs[1:]:
            assert torch.equal(dw_x, runs[0][0]) and torch.equal(dw_y, runs[0][1])
        assert errors(runs[0][0], ref)[0] < 2e-6
        assert errors(runs[0][1], one[1].double())[0] < 1e-6
    assert not hip.wgrad16_gave_up_waiting(DEV)
    from ctc_asr_amd.hip import _WGRAD16_SYNC
    assert int(_WGRAD16_SYNC[torch.cuda.current_device()].abs().sum()) == 0


@pytest.mark.parametrize('batch,frames', [(32, 399), (16, 329), (20, 261)])
def test_the_model_with_the_own_kernel_agrees_with_the_library_form(hip, batch, frames):
    """`CTCModel.backward` with `own_wgrad` against the library's TN GEMMs on the same weights and
    batch, every gradient slice; T' odd / rows of a range not a multiple of 32 in the later cases
    (the last stage of a range is zero-filled, the shifted output rows cross the sequence ends)."""
    from ctc_asr_amd.model import CTCModel, ModelConfig
    cfg = ModelConfig(used_model='ds2', conv_filters=(32, 32), num_units_dense=2048,
                      num_layers_rnn=2, num_units_rnn=1024, rnn_cell='lstm', cudnn=True,
                      dense_dropout_rate=0.0, conv_dropout_rate=0.0)
    rng = np.random.default_rng(3)
    feats = torch.tensor(rng.normal(size=(batch, frames, 80)).astype(np.float32), device=DEV)
    flen = torch.full((batch,), frames, dtype=torch.int32)
    labels = [list(rng.integers(1, 28, size=20)) for _ in range(batch)]

    def grads(own):
        model = CTCModel(cfg, DEV, seed=5)
        model.own_wgrad = own
        loss = model.forward_backward(feats, flen, labels)
        torch.cuda.synchronize()
        model.check_rnn_error()
        return float(loss), model.arena.grad.clone(), model

    loss_lib, grad_lib, _ = grads(False)
    loss_own, grad_own, model = grads(True)
    assert loss_lib == loss_own
    assert not hip.wgrad16_gave_up_waiting(DEV)
    for name, a, b in model.arena.layer_slices:
        ref = grad_lib[a:b].double()
        rel = float((ref - grad_own[a:b].double()).norm() / ref.norm().clamp_min(1e-30))
        assert rel < 2e-5, (name, rel)


def test_a_part_that_gives_up_adds_nothing_and_the_step_is_dropped(hip):
    """VERDICT r05 item 4 / ADVICE r05: a part of a tile that stops waiting for its turn raises the
    sticky word and leaves WITHOUT adding (dW stays a sum of whole parts, in order); every later
    launch leaves at once while the word is set; `step_guard` folds the word into the Adam skip
    flag; `hip.wgrad16_check` (from `CTCModel.check_rnn_error`) raises and hands the words back
    zeroed - through the Trainer: the update is not applied, the deferred check raises, the next
    step trains again."""
    from ctc_asr_amd.hip import _WGRAD16_SYNC
    g = torch.Generator(device=DEV).manual_seed(11)
    rows, m, nx = 1024, 512, 512
    d = torch.randn(rows, m, device=DEV, generator=g) * 1e-3
    x = torch.rand(rows, nx, device=DEV, generator=g) * 2 - 1
    good = own_wgrad(hip, d, x, 32768.0, parts=2)[0]
    sync = _WGRAD16_SYNC[torch.cuda.current_device()]
    hip.set_option('wgrad16_spin_limit', 1 << 16)     # (~20 ms instead of ~5 s)
    try:
        sync[1] = 99                    # tile 0's word: nobody's turn
        torch.cuda.synchronize()
        bad = own_wgrad(hip, d, x, 32768.0, parts=2)[0]
        torch.cuda.synchronize()
        assert hip.wgrad16_gave_up_waiting(DEV)
        assert float(bad[:256, :256].abs().max()) == 0.0        # tile 0: no part added
        status = torch.zeros(3, dtype=torch.int32, device=DEV)
        assert hip.step_guard(status, torch.ones(3, device=DEV)).tolist() == [1, 1 << 30]
        assert hip.step_guard(status, torch.ones(3, device=DEV), wgrad_word=False).tolist() == [0, 0]
        # the word is sticky: a later launch adds nothing at all
        later = own_wgrad(hip, d, x, 32768.0, parts=2)[0]
        assert float(later.abs().max()) == 0.0
        with pytest.raises(hip.CtcAsrError, match='turn'):
            hip.wgrad16_check(DEV)
        hip.wgrad16_check(DEV)
        assert int(sync.abs().sum()) == 0
        assert torch.equal(own_wgrad(hip, d, x, 32768.0, parts=2)[0], good)

        # through the Trainer (a model whose weight gradients take the own kernel in 2 parts)
        from ctc_asr_amd.engine import Trainer
        from ctc_asr_amd.model import ModelConfig
        cfg = ModelConfig(used_model='ds2', conv_filters=(32, 32), num_units_dense=2048,
                          num_layers_rnn=1, num_units_rnn=1024, rnn_cell='lstm', cudnn=True,
                          dense_dropout_rate=0.0, conv_dropout_rate=0.0)
        trainer = Trainer(cfg, device=DEV, seed=3)
        rng = np.random.default_rng(5)
        feats = torch.tensor(rng.normal(size=(20, 261, 80)).astype(np.float32))
        flen = torch.full((20,), 261, dtype=torch.int32)
        labels = [list(rng.integers(1, 28, size=12)) for _ in range(20)]
        trainer.train_step(feats, flen, labels)
        trainer.drain_checks()
        sync = _WGRAD16_SYNC[torch.cuda.current_device()]
        before = trainer.model.arena.param.clone()
        sync[1] = 99
        trainer.train_step(feats, flen, labels)
        torch.cuda.synchronize()
        assert torch.equal(trainer.model.arena.param, before)
        with pytest.raises(hip.CtcAsrError, match='turn'):
            trainer.drain_checks()
        trainer.drain_checks()
        trainer.train_step(feats, flen, labels)
        trainer.drain_checks()
        assert not torch.equal(trainer.model.arena.param, before)
        assert torch.isfinite(trainer.model.arena.param).all()
    finally:
        hip.set_option('wgrad16_spin_limit', 0)
        _WGRAD16_SYNC[torch.cuda.current_device()].zero_()
