"""fp32 GEMMs on the bf16 matrix pipe (ctcasr_split_bf16 + ctc_asr_amd/split_gemm.py): the split is
exact to 24 bits, the six-product GEMMs are at least as close to fp64 as the library's fp32 GEMM,
and a training step through them equals the fp32-GEMM step to fp32 round-off."""

import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _torch_split(x):
    a1 = x.to(torch.bfloat16)
    r = x - a1.float()
    a2 = r.to(torch.bfloat16)
    r = r - a2.float()
    return a1, a2, r.to(torch.bfloat16)


def test_split_kernel_is_the_three_piece_rne_split(hip):
    g = torch.Generator(device='cuda').manual_seed(0)
    # magnitudes from 1e-30 to 1e+20 in one matrix; exact zeros, negative zero, a denormal
    x = torch.randn(300, 64, device='cuda', generator=g) * \
        torch.logspace(-30, 20, 300, device='cuda')[:, None]
    x[5, :8] = torch.tensor([0.0, -0.0, 1e-40, -1e-40, 1.0, -1.0, 65504.0, 3.0e38], device='cuda')
    got = hip.split_bf16(x, (0, 1, 2))
    for piece, want in enumerate(_torch_split(x)):
        assert torch.equal(got[:, piece].view(torch.int16), want.view(torch.int16)), piece
    # the pieces carry every one of the 24 mantissa bits of a normal number: the sum IS x
    normal = x.abs() > 1e-30
    total = got[:, 0].double() + got[:, 1].double() + got[:, 2].double()
    assert torch.equal(total[normal].float(), x[normal])
    # block orders and strided outputs: a row / column range of a bigger [rows, blocks, cols]
    six = hip.split_bf16(x, (0, 1, 2, 0, 1, 0))
    for block, piece in enumerate((0, 1, 2, 0, 1, 0)):
        assert torch.equal(six[:, block], got[:, piece])
    big = torch.zeros(300, 3, 128, dtype=torch.bfloat16, device='cuda')
    hip.split_bf16(x[10:50, 32:64], (2, 1, 0), out=big[10:50, :, 64:96])
    assert torch.equal(big[10:50, 0, 64:96], got[10:50, 2, 32:64])
    assert torch.equal(big[10:50, 2, 64:96], got[10:50, 0, 32:64])
    assert float(big[:, :, :64].abs().sum()) == 0 and float(big[50:].abs().sum()) == 0
    # blocks stacked along the rows (an operand whose K axis is its row axis)
    stacked = torch.empty(3, 300, 64, dtype=torch.bfloat16, device='cuda')
    hip.split_bf16(x, (0, 1, 2), out=stacked.permute(1, 0, 2))
    assert torch.equal(stacked[1], got[:, 1])
    # argument errors come back as errors, not as launches
    with pytest.raises(hip.CtcAsrError):
        hip.split_bf16(x[:, :12], (0, 1, 2))                 # cols % 8
    with pytest.raises(hip.CtcAsrError):
        hip.split_bf16(x, (0, 1, 3))                         # no such piece
    with pytest.raises(hip.CtcAsrError):
        hip.split_bf16(x, (0,) * 7)                          # too many blocks
    with pytest.raises(hip.CtcAsrError):
        hip.split_bf16(x.cpu(), (0, 1, 2))


def _errors(got, ref):
    d = got.double() - ref
    return float(d.pow(2).mean().sqrt() / ref.pow(2).mean().sqrt()), float(d.abs().max())


def test_split_products_are_at_least_as_close_to_fp64_as_the_fp32_gemm(hip):
    """The three product layouts of split_gemm.py on a layer-sized problem (rows scaled down):
    forward projection, data gradient, weight gradient (with a row range and a shift, as the
    recurrent weight gradient uses them) against fp64; the plain fp32 GEMM beside them."""
    from ctc_asr_amd import split_gemm as sg
    g = torch.Generator(device='cuda').manual_seed(1)
    rows, feat, gates = 2048, 1024, 4096
    # ReLU-like activations with exact zeros, weights ~ 1/sqrt(K), gradients spanning 6 decades
    x = torch.randn(rows, feat, device='cuda', generator=g).clamp_(0, 20)
    w = torch.randn(gates, feat, device='cuda', generator=g) / feat ** 0.5
    d = torch.randn(rows, gates, device='cuda', generator=g) * \
        torch.logspace(-7, -1, rows, device='cuda')[torch.randperm(rows, device='cuda')][:, None]
    xs, ws = sg.split(x, sg.A_ORDER), sg.split(w, sg.B_ORDER)
    ds = sg.split(d, sg.B_ORDER)
    wt = sg.split(w.t().contiguous(), sg.A_ORDER)
    checks = []
    ref = x.double() @ w.double().t()
    checks.append(('forward', sg.mm_nt(xs, ws), torch.mm(x, w.t()), ref))
    ref = d.double() @ w.double()
    dx = torch.empty(rows, feat, device='cuda')
    checks.append(('data gradient', sg.mm_nt_by_order(dx, ds, wt).clone(), torch.mm(d, w), ref))
    sg.mm_pieces(dx, ds.piece, ws.piece)
    checks.append(('data gradient, six calls', dx.clone(), torch.mm(d, w), ref))
    lo, hi, shift = 64, 1984, -32
    ref = d[lo:hi, 1024:3072].double().t() @ x[lo + shift:hi + shift, 256:768].double()
    dw = torch.zeros(2048, 512, device='cuda')
    sg.mm_tn_rows(dw, ds, xs, lo, hi, a_cols=slice(1024, 3072), b_cols=slice(256, 768),
                  b_shift=shift)
    checks.append(('weight gradient', dw,
                   torch.mm(d[lo:hi, 1024:3072].t(), x[lo + shift:hi + shift, 256:768]), ref))
    stacked = sg.split_rows_stacked(w.t().contiguous(), sg.B_ORDER)       # [6 feat, gates]
    checks.append(('forward, stacked rows', sg.mm_nn_stacked(xs, stacked), torch.mm(x, w.t()),
                   x.double() @ w.double().t()))
    for name, split, plain, ref in checks:
        s_rms, s_max = _errors(split, ref)
        p_rms, p_max = _errors(plain, ref)
        # fp32-grade: rms relative error of a few 1e-7 (fp32 epsilon 6e-8 times sqrt(K) growth),
        # and not worse than the library's fp32 GEMM on the same operands
        assert s_rms < 1.5e-6, (name, s_rms)
        assert s_rms <= 1.25 * p_rms + 1e-8, (name, s_rms, p_rms)
        assert s_max <= 2.0 * p_max + 1e-12, (name, s_max, p_max)


def _train_steps(split, batch, frames, cell, steps=1):
    from ctc_asr_amd.model import CTCModel, ModelConfig
    os.environ['CTCASR_SPLIT_GEMM'] = split
    try:
        cfg = ModelConfig(used_model='ds2', conv_filters=(32, 32), num_units_dense=2048,
                          num_layers_rnn=2, num_units_rnn=1024, rnn_cell=cell, cudnn=True,
                          dense_dropout_rate=0.0, conv_dropout_rate=0.0)
        model = CTCModel(cfg, 'cuda', seed=5)
    finally:
        os.environ.pop('CTCASR_SPLIT_GEMM', None)
    assert model.split_gemm == (split == '1')
    rng = np.random.default_rng(1)
    feats = torch.tensor(rng.normal(size=(batch, frames, 80)).astype(np.float32), device='cuda')
    flen = torch.full((batch,), frames, dtype=torch.int32)
    labels = [list(rng.integers(1, 28, size=20)) for _ in range(batch)]
    for _ in range(steps):
        loss = model.forward_backward(feats, flen, labels)
        model.apply_gradients(1e-4)
    torch.cuda.synchronize()
    model.check_rnn_error()
    return float(loss), model.last_logits.clone(), model.arena.grad.clone(), model


@pytest.mark.parametrize('cell,batch', [('lstm', 32), ('lstm', 16), ('gru', 24)])
def test_training_step_equals_the_fp32_gemm_step(hip, cell, batch):
    """Same weights, same batch: loss, logits and every gradient of the split-GEMM step against the
    step with the library's fp32 GEMMs (CTCASR_SPLIT_GEMM=0).  Both carry fp32 round-off through
    200 recurrent steps; the two agree far inside the 1e-3 parity bar."""
    loss0, logits0, grad0, model0 = _train_steps('0', batch, 400, cell)
    loss1, logits1, grad1, model1 = _train_steps('1', batch, 400, cell)
    assert model1._w_split and not model0._w_split          # the split path really ran
    assert abs(loss0 - loss1) <= 2e-6 * abs(loss0)
    assert float((logits0 - logits1).abs().max()) < 5e-6
    for name, a, b in model0.arena.layer_slices:
        ref = grad0[a:b].double()
        rel = float((ref - grad1[a:b].double()).pow(2).sum().sqrt() /
                    ref.pow(2).sum().sqrt().clamp_min(1e-30))
        assert rel < 5e-4, (name, rel)


@pytest.mark.timeout(300)
def test_c2_shaped_steps_with_one_library_gemm_in_flight(hip):
    """Regression: at the C2 shape (16 x 10 s, two layers) the bf16 data gradient of the bottom
    layer [8000 x 640] on the main stream and a [4096 x 640] weight gradient on the side stream
    - two stream-K GEMMs of the library in flight at once - waited for each other forever in the
    second training step.  `backward` now lets the side stream's GEMMs finish before a main-stream
    GEMM starts; six steps run and the loss moves."""
    loss, _, _, model = _train_steps('1', 16, 1000, 'lstm', steps=6)
    assert np.isfinite(loss) and model.step_count == 6


@pytest.mark.parametrize('m,n,k', [(300, 270, 48), (256, 256, 16), (1, 5, 32), (700, 513, 2048),
                                   (2048, 1024, 640)])
def test_own_split_gemm_kernel(hip, m, n, k):
    """`ctcasr_gemm_split_nt` - fp32 tiles split in registers, six bf16 MFMAs per fragment pair,
    no library call - against fp64: partial tiles in both directions, K from one step to many,
    strided operands and output, accumulation.  fp32-grade: its error stays within 1.5 x of the
    library fp32 GEMM's on the same operands (it adds all six products into ONE fp32 accumulator,
    where the K-concatenated library form sums the small terms first)."""
    g = torch.Generator(device='cuda').manual_seed(m + n + k)
    big_a = torch.randn(m, k + 8, device='cuda', generator=g)
    big_b = torch.randn(n, k + 4, device='cuda', generator=g) / k ** 0.5
    a, b = big_a[:, 4:4 + k], big_b[:, :k]                 # row strides k + 8 / k + 4
    ref = a.double() @ b.double().t()
    got = hip.gemm_split_nt(a, b)
    plain = torch.mm(a, b.t())
    scale = float(ref.abs().max()) + 1e-30
    assert float((got.double() - ref).abs().max()) <= \
        1.5 * float((plain.double() - ref).abs().max()) + 2e-7 * scale
    # into a column range of a wider matrix, accumulating
    wide = torch.ones(m, n + 6, device='cuda')
    hip.gemm_split_nt(a, b, out=wide[:, 3:3 + n], accumulate=True)
    assert torch.equal(wide[:, :3], torch.ones(m, 3, device='cuda'))
    assert torch.equal(wide[:, 3 + n:], torch.ones(m, 3, device='cuda'))
    assert float((wide[:, 3:3 + n].double() - 1.0 - ref).abs().max()) <= \
        1.5 * float((plain.double() - ref).abs().max()) + 4e-7 * (scale + 1.0)
    with pytest.raises(hip.CtcAsrError):
        hip.gemm_split_nt(a[:, :k - 4], b[:, :k - 4])        # K % 16
    with pytest.raises(hip.CtcAsrError):
        hip.gemm_split_nt(a, b[:, :k - 16])                  # K mismatch


@pytest.mark.parametrize('m,n,k', [(300, 270, 48), (256, 256, 16), (5, 1, 7), (513, 700, 1000),
                                   (4096, 640, 5344)])
def test_own_split_gemm_kernel_over_the_row_axis(hip, m, n, k):
    """`ctcasr_gemm_split_tn`: out (+)= a[K, M]^T b[K, N] - the weight-gradient form - for any K
    (rows past K count as zeros), column ranges of wider operands, accumulation into a zeroed or
    a filled out."""
    g = torch.Generator(device='cuda').manual_seed(m * 7 + n + k)
    wide_a = torch.randn(k, m + 5, device='cuda', generator=g) * \
        torch.logspace(-4, 0, k, device='cuda')[:, None]
    wide_b = torch.randn(k, n + 3, device='cuda', generator=g).clamp_(0, 20)
    a, b = wide_a[:, 2:2 + m], wide_b[:, 1:1 + n]
    ref = a.double().t() @ b.double()
    plain = torch.mm(a.t(), b)
    scale = float(ref.abs().max()) + 1e-30
    out = torch.full((m, n), 3.0, device='cuda')
    hip.gemm_split_tn(a, b, out, accumulate=False)
    bound = 1.5 * float((plain.double() - ref).abs().max()) + 2e-7 * scale
    assert float((out.double() - ref).abs().max()) <= bound
    hip.gemm_split_tn(a, b, out, accumulate=True)
    assert float((out.double() - 2.0 * ref).abs().max()) <= 2.5 * bound
    with pytest.raises(hip.CtcAsrError):
        hip.gemm_split_tn(a, b[:-1], out)


def test_fp16_two_piece_split_and_the_forward_product(hip):
    """`ctcasr_split_f16`: h1 = rne_f16(x s), h2 = rne_f16(x s - h1) bit for bit, and the three
    products h1 k1 + h1 k2 + h2 k1 with 1 / (s_x s_w) as the GEMM's alpha against fp64 - for the
    bounded operands the model feeds it (|h| <= 1 outputs of gated cells, activations behind the
    clipped ReLU): as close as the library's fp32 GEMM, by assertion."""
    from ctc_asr_amd import split_gemm as sg
    g = torch.Generator(device='cuda').manual_seed(3)
    x = torch.randn(300, 64, device='cuda', generator=g).clamp_(-1, 1) * \
        torch.logspace(-9, 0, 300, device='cuda')[:, None]
    x[7, :4] = torch.tensor([1.0, -1.0, 0.0, 2.0 ** -20], device='cuda')
    scale = sg.f16_scale(1.0)
    assert scale == 2.0 ** 15 and sg.f16_scale(20.0) == 2.0 ** 11 and sg.f16_scale(22.3) == 2.0 ** 11
    assert sg.f16_scale(65.0) is None and sg.f16_scale(None) is None
    got = hip.split_f16(x, scale, (0, 1, 0))
    s = x * scale
    h1 = s.to(torch.float16)
    h2 = (s - h1.float()).to(torch.float16)
    assert torch.equal(got[:, 0].view(torch.int16), h1.view(torch.int16))
    assert torch.equal(got[:, 1].view(torch.int16), h2.view(torch.int16))
    assert torch.equal(got[:, 2], got[:, 0]) and bool(torch.isfinite(got.float()).all())
    with pytest.raises(hip.CtcAsrError):
        hip.split_f16(x, 0.0, (0, 1))
    with pytest.raises(hip.CtcAsrError):
        hip.split_f16(x, scale, (0, 2))
    rows, k, n = 2048, 2048, 4096
    w = torch.randn(n, k, device='cuda', generator=g) / k ** 0.5
    ws = sg.split16(w, sg.W_SCALE, sg.H_B)
    cases = {
        'gated-cell outputs': (torch.sigmoid(torch.randn(rows, k, device='cuda', generator=g) * 2) *
                               torch.tanh(torch.randn(rows, k, device='cuda', generator=g) * 1.5),
                               1.0),
        'clipped ReLU with dropout scale': ((torch.randn(rows, k, device='cuda', generator=g) * 6)
                                            .clamp_(0, 20) / 0.9, 20.0 / 0.9)}
    for name, (a, bound) in cases.items():
        sa = sg.f16_scale(bound)
        ref = a.double() @ w.double().t()
        got = sg.mm_nt16(sg.split16(a, sa, sg.H_A), ws, sa * sg.W_SCALE)
        s_rms, s_max = _errors(got, ref)
        p_rms, p_max = _errors(torch.mm(a, w.t()), ref)
        assert s_rms <= 1.3 * p_rms + 1e-8, (name, s_rms, p_rms)
        assert s_max <= 2.0 * p_max + 1e-12, (name, s_max, p_max)


def test_scaled_fp16_splits_and_the_gradient_products(hip):
    """The gradient GEMMs in the fp16 form: `ctcasr_colmax_scale` (power of two per column, the
    largest magnitude lands in [2^13, 2^14), all-zero columns get 1), `ctcasr_split_f16_cols` /
    `_rows` (bit for bit against torch), `ctcasr_rescale_rows`, and the two products built from
    them - weight gradient with per-column scales of the gradient operand, data gradient with
    per-row scales - against fp64 next to the library's fp32 GEMM, on gradient-like operands:
    frames spanning eight decades, gate units three, a tenth exact zeros, one all-zero column."""
    from ctc_asr_amd import split_gemm as sg
    g = torch.Generator(device='cuda').manual_seed(11)
    rows, gates, feat = 4096, 2048, 1024
    d = torch.randn(rows, gates, device='cuda', generator=g) * \
        torch.logspace(-10, -2, rows, device='cuda')[torch.randperm(rows, device='cuda')][:, None] * \
        torch.logspace(-2, 1, gates, device='cuda')[None, :] * \
        (torch.rand(rows, gates, device='cuda', generator=g) < 0.9)
    d[:, 5] = 0.0
    d[17] = 0.0
    x = torch.sigmoid(torch.randn(rows, feat, device='cuda', generator=g) * 2) * \
        torch.tanh(torch.randn(rows, feat, device='cuda', generator=g))
    w = torch.randn(gates, feat, device='cuda', generator=g) / feat ** 0.5
    # column scales
    scale, inv = hip.colmax_scale(d)
    top = d.abs().amax(dim=0) * scale
    live = d.abs().amax(dim=0) > 0
    assert bool(((top[live] >= 2.0 ** 13) & (top[live] < 2.0 ** 14)).all())
    assert float(scale[5]) == 1.0 and torch.equal(scale * inv, torch.ones_like(scale))
    assert bool((torch.log2(scale) == torch.log2(scale).round()).all())       # powers of two
    got = hip.split_f16_cols(d, scale, 1.0, sg.H_B)
    s = d * scale
    h1 = s.to(torch.float16)
    h2 = (s - h1.float()).to(torch.float16)
    assert torch.equal(got[:, 0].view(torch.int16), h1.view(torch.int16))
    assert torch.equal(got[:, 1].view(torch.int16), h2.view(torch.int16))
    assert torch.equal(got[:, 2], got[:, 0])
    # row scales found on the fly
    rgot, rinv = hip.split_f16_rows(d, sg.H_A)
    rtop = d.abs().amax(dim=1) / rinv
    rlive = d.abs().amax(dim=1) > 0
    assert bool(((rtop[rlive] >= 2.0 ** 13) & (rtop[rlive] < 2.0 ** 14)).all())
    assert float(rinv[17]) == 1.0
    rs = d / rinv[:, None]
    assert torch.equal(rgot[:, 0].view(torch.int16), rs.to(torch.float16).view(torch.int16))
    assert torch.equal(rgot[:, 2].view(torch.int16),
                       (rs - rs.to(torch.float16).float()).to(torch.float16).view(torch.int16))
    # rescale
    t = torch.randn(300, 64, device='cuda', generator=g)
    f = torch.rand(300, device='cuda', generator=g) + 0.5
    out = torch.ones(300, 64, device='cuda')
    hip.rescale_rows(t, f, 0.25, out, accumulate=True)
    assert torch.allclose(out, 1.0 + t * f[:, None] * 0.25, rtol=1e-6, atol=1e-7)
    hip.rescale_rows(t, f, 2.0, t, accumulate=False)                  # in place
    # the weight gradient of a range of rows, with a shift and a column range on the bounded side
    sx = sg.f16_scale(1.0)
    x16 = sg.split16(x, sx, sg.H_A)
    lo, hi, shift = 256, 3840, -32
    d16, dinv = sg.wgrad16_operand(d[lo:hi, 512:1536])
    dw = torch.zeros(1024, 512, device='cuda')
    sg.wgrad16(dw, d16, dinv, x16, sx, lo + 64 + shift, x_cols=slice(256, 768),
               d_rows=slice(64, hi - lo))
    ref = d[lo + 64:hi, 512:1536].double().t() @ x[lo + 64 + shift:hi + shift, 256:768].double()
    plain = torch.mm(d[lo + 64:hi, 512:1536].t(), x[lo + 64 + shift:hi + shift, 256:768])
    s_rms, s_max = _errors(dw, ref)
    p_rms, p_max = _errors(plain, ref)
    assert s_rms <= 1.3 * p_rms + 1e-8 and s_max <= 2.5 * p_max + 1e-12, (s_rms, p_rms, s_max, p_max)
    # the data gradient
    wt16 = sg.split16(w.t().contiguous(), sg.W_SCALE, sg.H_B)
    ref = d.double() @ w.double()
    dx = sg.dgrad16(d, wt16, sg.W_SCALE)
    plain = torch.mm(d, w)
    # (row by row: every frame has its own scale, and its own magnitude)
    row_ref = ref.abs().amax(dim=1).clamp_min(1e-300)
    s_err = ((dx.double() - ref).abs().amax(dim=1) / row_ref)[rlive]
    p_err = ((plain.double() - ref).abs().amax(dim=1) / row_ref)[rlive]
    # fp32-grade, not fp32-equal: two pieces carry 22 bits and h2 k2 is dropped - over K = 2048
    # gate units the rows come out at ~2.4 x the fp32 GEMM's (tiny) error
    assert float(s_err.max()) < 2e-5 and float(s_err.mean()) <= 3.0 * float(p_err.mean()) + 1e-8
    assert float(dx[17].abs().max()) == 0.0


def test_fp16_forms_of_an_unbounded_operand(hip):
    """Layers behind a ReLU cell (the reference's default model, asr/params.py:43-50) have inputs
    without a bound: the projection x W^T takes x with a scale per ROW found on the device
    (`split_gemm.mm_rows16`), the weight gradients d^T x take x with a scale per COLUMN
    (`split_gemm.ColScaled`) - both against float64 next to the library's fp32 GEMM, on an x whose
    rows span six decades and whose columns span four, with outliers in the thousands."""
    from ctc_asr_amd import split_gemm
    g = torch.Generator(device='cuda').manual_seed(9)
    rows, k, n = 4096, 2048, 1536
    x = torch.randn(rows, k, device='cuda', generator=g).abs()
    x *= torch.logspace(-3, 3, rows, device='cuda').view(-1, 1)
    x *= torch.logspace(-2, 2, k, device='cuda')[torch.randperm(k, device='cuda', generator=g)]
    x[7, 5], x[100, 2000] = 9000.0, 3.0e4
    w = torch.randn(n, k, device='cuda', generator=g) / np.sqrt(k)
    w16 = split_gemm.split16(w, split_gemm.W_SCALE, split_gemm.H_B)
    got = split_gemm.mm_rows16(x, w16, split_gemm.W_SCALE)
    ref = x.double() @ w.double().t()
    lib = torch.mm(x, w.t())
    err = lambda a: float(((a.double() - ref).abs().amax(dim=1) / ref.abs().amax(dim=1)).max())
    assert torch.isfinite(got).all() and err(got) < 3.0 * err(lib) + 1e-7, (err(got), err(lib))
    # the same product with W's pieces stacked along its rows (dense4: x K)
    stacked = split_gemm.split16_rows_stacked(w.t().contiguous(), split_gemm.W_SCALE,
                                              split_gemm.H_B)
    got_s = split_gemm.mm_rows16(x, stacked, split_gemm.W_SCALE, stacked=True)
    assert err(got_s) < 3.0 * err(lib) + 1e-7
    # weight gradient: d [rows, m] (scaled per column like a layer's dxw) against the unbounded x
    m = 1024
    d = torch.randn(rows, m, device='cuda', generator=g) * \
        torch.logspace(-9, -3, m, device='cuda').view(1, -1)
    d16, inv = split_gemm.wgrad16_operand(d)
    cols = split_gemm.ColScaled(x)
    out = torch.zeros(m, k, device='cuda')
    split_gemm.wgrad16(out, d16, inv, cols[0], cols[1], 0, x_col_inv=cols.col_inv)
    ref_w = d.double().t() @ x.double()
    lib_w = torch.mm(d.t(), x)
    rel = lambda a: float((a.double() - ref_w).norm() / ref_w.norm())
    col = lambda a: float(((a.double() - ref_w).abs().amax(dim=0) / ref_w.abs().amax(dim=0)).max())
    assert rel(out) < 2.0 * rel(lib_w) + 1e-7 and col(out) < 3.0 * col(lib_w) + 1e-7, \
        (rel(out), rel(lib_w), col(out), col(lib_w))
    # a shifted row range and a column slice (W_hh's product reads the layer output one step off)
    out2 = torch.zeros(m, 512, device='cuda')
    split_gemm.wgrad16(out2, d16, inv, cols[0], cols[1], 64, x_cols=slice(256, 768),
                       d_rows=slice(0, rows - 64), x_col_inv=cols.col_inv)
    ref2 = d[:rows - 64].double().t() @ x[64:, 256:768].double()
    assert float((out2.double() - ref2).norm() / ref2.norm()) < 3e-6
