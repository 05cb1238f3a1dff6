"""The big fp32 GEMMs of the stack (RNN input projections, their data and weight gradients) on the
bf16 matrix pipe, at fp32 accuracy.

gfx950 has no xf32 and its fp32 MFMA runs at 1/16 of the bf16 rate (MI355X_MICROARCH.md), and the
library's fp32 GEMM sits at 88 % of that peak already: the only way to make
these products faster is to leave the fp32 pipe.  Each fp32 operand is split into three bfloat16
pieces a = a1 + a2 + a3 (`hip.split_bf16`, 24 mantissa bits in all, fp32's exponent range) and the
product is the six partial products of order <= 2,

    a b  ~  a1 b1 + (a1 b2 + a2 b1) + (a1 b3 + a2 b2 + a3 b1),        dropped terms <= 2^-24 |a b|,

accumulated in fp32 by the library's bf16 GEMM (6/16 of the fp32-MFMA time on paper).  Measured
against fp64 on the C3 layer shapes the result is CLOSER than the fp32 GEMM's (rms relative error
1.5e-7 .. 6e-7 against 4e-7 .. 1.6e-6: the pieces' products are exact and the accumulation is the
same fp32 one; `tools/gemm_split_probe.py`, `profiles/r03_gemm_bf16_split.md`).

Layouts.  Every split operand is stored [rows, 6, cols] with the piece order `A_ORDER` or
`B_ORDER`; block k of one order pairs with block k of the other.
  - K = the column axis of both operands (forward projections): the buffers read as
    [rows, 6 cols] ARE the K-concatenated operands - one GEMM over 6 K (`mm_nt`);
  - K = the row axis of both operands (weight gradients): the buffers read as [6 rows, cols]
    pair block k of a row with block k of the same row - again one GEMM (`mm_tn_rows`);
  - K = columns of one, rows of the other (data gradient): six accumulating calls on piece views
    (`mm_pieces`); the output is the small matrix there, the extra passes over it cost little, and
    the library's kernels for one K = 49152 call measured slower than six K = 8192 calls.
"""

import torch

from . import hip

A_ORDER = (0, 1, 2, 0, 1, 0)
B_ORDER = (2, 1, 0, 1, 0, 0)
THREE = (0, 1, 2)
# (piece of the first operand, piece of the second), smallest terms first
PAIRS = ((2, 0), (1, 1), (0, 2), (1, 0), (0, 1), (0, 0))
_BLOCK_OF_PIECE = {A_ORDER: (0, 1, 2), B_ORDER: (2, 1, 0), THREE: (0, 1, 2)}
F32 = torch.float32


class Split:
    """bf16 pieces of an fp32 matrix [rows, cols]: `buf` [rows, blocks, cols]."""

    __slots__ = ('buf', 'order')

    def __init__(self, buf, order):
        self.buf, self.order = buf, tuple(order)

    @property
    def rows(self):
        return self.buf.shape[0]

    @property
    def cols(self):
        return self.buf.shape[2]

    def piece(self, p):
        """Piece p as a [rows, cols] view (row stride blocks * cols)."""
        return self.buf[:, _BLOCK_OF_PIECE[self.order][p]]

    def concat(self):
        """[rows, blocks * cols]: the K-concatenated operand."""
        rows, blocks, cols = self.buf.shape
        return self.buf.view(rows, blocks * cols)


def split(x2d, order, out=None):
    return Split(hip.split_bf16(x2d, order, out=None if out is None else out.buf), order)


def empty(rows, cols, order, device):
    return Split(torch.empty((rows, len(order), cols), dtype=torch.bfloat16, device=device), order)


def mm_nt(a, b, out=None, rows=None, accumulate=False):
    """out[M, N] (+)= A[M, K] . B[N, K]^T from a = Split(A, A_ORDER), b = Split(B, B_ORDER) (or the
    orders the other way round) in ONE call; ``rows`` = a slice of A's rows (then out is that
    slice's product)."""
    assert {a.order, b.order} == {A_ORDER, B_ORDER} and a.cols == b.cols
    lhs = a.concat() if rows is None else a.concat()[rows]
    if out is None:
        return torch.mm(lhs, b.concat().t(), out_dtype=F32)
    if accumulate:
        return torch.addmm(out, lhs, b.concat().t(), out_dtype=F32, out=out)
    return torch.mm(lhs, b.concat().t(), out_dtype=F32, out=out)


def mm_nt_by_order(out, a, b):
    """out[M, N] = A . B^T like `mm_nt`, as three accumulating calls over the block ranges
    [0, 3), [3, 5), [5, 6) - the terms of order 2, 1, 0.  For a SMALL out beside a long K (the data
    gradient: N = 2048, K = 6 x 8192) the library's kernels run three K <= 24576 calls 1.4 x
    faster than one K = 49152 call (tools/gemm_split_probe.py)."""
    assert {a.order, b.order} == {A_ORDER, B_ORDER} and a.cols == b.cols
    k = a.cols
    lhs, rhs = a.concat(), b.concat()
    torch.mm(lhs[:, :3 * k], rhs[:, :3 * k].t(), out_dtype=F32, out=out)
    torch.addmm(out, lhs[:, 3 * k:5 * k], rhs[:, 3 * k:5 * k].t(), out_dtype=F32, out=out)
    return torch.addmm(out, lhs[:, 5 * k:], rhs[:, 5 * k:].t(), out_dtype=F32, out=out)


def mm_nn_stacked(a, b_stacked, out=None):
    """out[M, N] = A[M, K] . B[K, N] from a = Split(A, order) and the pieces of B stacked along its
    rows, [6 K, N] in the opposite order (`split_rows_stacked`): one call."""
    return torch.mm(a.concat(), b_stacked, out_dtype=F32) if out is None else \
        torch.mm(a.concat(), b_stacked, out_dtype=F32, out=out)


def split_rows_stacked(x2d, order, out=None):
    """Pieces of x [K, N] stacked along the rows: bf16 [6 K, N], block k = rows [k K, (k + 1) K)."""
    rows, cols = x2d.shape
    if out is None:
        out = torch.empty((len(order), rows, cols), dtype=torch.bfloat16, device=x2d.device)
    hip.split_bf16(x2d, order, out=out.permute(1, 0, 2))
    return out.view(len(order) * rows, cols)


def mm_pieces(out, lhs, rhs, accumulate=False):
    """out (+)= sum over `PAIRS` of lhs(i) . rhs(j): ``lhs`` / ``rhs`` map a piece index to the
    bf16 matrix view to multiply ([M, K] and [K, N]; transposed / sliced views are fine)."""
    for i, j in PAIRS:
        if accumulate:
            torch.addmm(out, lhs(i), rhs(j), out_dtype=F32, out=out)
        else:
            torch.mm(lhs(i), rhs(j), out_dtype=F32, out=out)
            accumulate = True
    return out


def mm_tn_rows(out, a, b, lo, hi, a_cols=slice(None), b_cols=slice(None), b_shift=0,
               accumulate=True):
    """out[Ma, Nb] (+)= A[lo:hi, a_cols]^T . B[lo + b_shift:hi + b_shift, b_cols] - a product over
    the ROW axis of both operands (weight gradients) - in ONE call: with a = Split(A, B_ORDER) and
    b = Split(B, A_ORDER) (or the other way round) the buffers read as [rows * 6, cols] matrices
    pair block k of a row of A with block k of the matching row of B: exactly the six products."""
    assert {a.order, b.order} == {A_ORDER, B_ORDER}
    lhs = a.buf.view(a.rows * 6, a.cols)[6 * lo:6 * hi, a_cols].t()
    rhs = b.buf.view(b.rows * 6, b.cols)[6 * (lo + b_shift):6 * (hi + b_shift), b_cols]
    if accumulate:
        return torch.addmm(out, lhs, rhs, out_dtype=F32, out=out)
    return torch.mm(lhs, rhs, out_dtype=F32, out=out)


# ---- forward projections of bounded operands: two fp16 pieces, three products ------------------
H_A, H_B = (0, 0, 1), (0, 1, 0)            # h1 k1 + h1 k2 + h2 k1
F16_MAX = 60000.0                          # (fp16's largest finite value is 65504)
W_SCALE = 2.0 ** 11                        # weights: |w| < 29
RNN_F16_H_SCALE = 2.0 ** 15                # scale of the pieces of h the fp16-pipe recurrence
                                           # kernels publish / write (PRNN_F16_H_SCALE)


def f16_scale(bound):
    """The power of two that maps [-bound, bound] into fp16's range, or None if there is none
    worth using (the split then keeps too few bits of the small elements)."""
    if bound is None or not bound > 0.0 or bound > 64.0:
        return None
    scale = 2.0 ** 15
    while bound * scale > F16_MAX:
        scale /= 2.0
    return scale


def split16(x2d, scale, order, out=None):
    return Split(hip.split_f16(x2d, scale, order, out=None if out is None else out.buf), order)


def empty16(rows, cols, order, device):
    return Split(torch.empty((rows, len(order), cols), dtype=torch.float16, device=device), order)


def mm_nt16(a, b, scale_product, out=None):
    """out[M, N] = A . B^T from a = split16(A, sa, H_A), b = split16(B, sb, H_B), scale_product =
    sa * sb: one fp16 GEMM over 3 K with 1 / (sa sb) as its alpha."""
    assert {a.order, b.order} == {H_A, H_B} and a.cols == b.cols
    if out is None:
        out = torch.empty((a.rows, b.rows), dtype=F32, device=a.buf.device)
    return torch.addmm(out, a.concat(), b.concat().t(), out_dtype=F32, beta=0.0,
                       alpha=1.0 / scale_product, out=out)


def split16_rows_stacked(x2d, scale, order, out=None):
    """fp16 pieces of x [K, N] * scale stacked along the rows: [3 K, N]."""
    rows, cols = x2d.shape
    if out is None:
        out = torch.empty((len(order), rows, cols), dtype=torch.float16, device=x2d.device)
    hip.split_f16(x2d, scale, order, out=out.permute(1, 0, 2))
    return out.view(len(order) * rows, cols)


def mm_nn16_stacked(a, b_stacked, scale_product, out=None):
    """out[M, N] = A[M, K] . B[K, N] from a = split16(A, sa, H_A) and the fp16 pieces of B stacked
    along its rows in the order H_B: one call, 1 / (sa sb) as its alpha."""
    if out is None:
        out = torch.empty((a.rows, b_stacked.shape[1]), dtype=F32, device=a.buf.device)
    return torch.addmm(out, a.concat(), b_stacked, out_dtype=F32, beta=0.0,
                       alpha=1.0 / scale_product, out=out)


# ---- gradient GEMMs in the fp16 form: scales per column (weight gradients) / per row (data
# gradient) of dxw, found on the device; the bounded operand keeps its fixed scale ----------------
def wgrad16_operand(d2d, colmax=None):
    """fp16 pieces [rows, 3, cols] (order H_B) of a block of gradient rows, scaled per column, and
    the inverse scales f32[cols].  ``colmax``: int32[cols] bit patterns of the column maxima when
    the kernel that wrote the block has found them already (`hip.rnn_bwd(..., colmax=)`), else a
    pass over the block finds them."""
    scale, inv = hip.colmax_scale(d2d) if colmax is None else hip.colscale_from_max(colmax)
    return hip.split_f16_cols(d2d, scale, 1.0, H_B), inv


class ColScaled:
    """fp16 pieces of an UNBOUNDED matrix x [rows, cols] (the output of a ReLU-cell layer) for the
    weight-gradient products over its ROW axis: a power of two per COLUMN - x's column is the
    product's output column, the scale comes back out of it - found on the device and applied on
    first use, on whatever stream is current then (the weight gradients' side stream).  Quacks like
    the (pieces, scale) pair of a bounded input: [0] the pieces, [1] their common scale (1),
    `col_inv` the inverse column scales."""

    def __init__(self, x2d):
        self.x2d, self._pieces, self.col_inv, self._made_on = x2d, None, None, None

    def __getitem__(self, index):
        if index == 1:
            return 1.0
        current = torch.cuda.current_stream(self.x2d.device)
        if self._pieces is None:
            scale, self.col_inv = hip.colmax_scale(self.x2d)
            self._pieces = Split(hip.split_f16_cols(self.x2d, scale, 1.0, H_A), H_A)
            self._made_on = current
        elif current != self._made_on:
            # read on another stream than the one whose allocator made them (the whole-chip path
            # runs the upper layers' weight gradients on the main stream, the bottom layer's on the
            # side stream): the block must not be handed out again before that stream is through
            self._pieces.buf.record_stream(current)
            self.col_inv.record_stream(current)
        return self._pieces


def mm_rows16(x2d, w16, w_scale, stacked=False):
    """x [R, K] . W^T for an UNBOUNDED x: x split with one scale per ROW found on the fly
    (`hip.split_f16_rows`, like a layer's dxw), one fp16 GEMM over 3 K against the fixed-scale
    pieces of W ([N, 3, K] `split16(.., H_B)`, or ``stacked`` [3 K, N] for a product x K), the row
    scales and 1 / w_scale taken back out in place."""
    d16, inv = hip.split_f16_rows(x2d, H_A)
    rows, _, k = d16.shape
    rhs = w16 if stacked else w16.concat().t()
    tmp = torch.mm(d16.view(rows, 3 * k), rhs, out_dtype=F32)
    return hip.rescale_rows(tmp, inv, 1.0 / w_scale, tmp, accumulate=False)


def wgrad16(out, d16, inv, x16, x_scale, x_lo, x_cols=slice(None), d_rows=slice(None),
            x_col_inv=None):
    """out[M, N] += D^T X for D = the rows ``d_rows`` of the block whose pieces are ``d16`` (from
    `wgrad16_operand`) and X = rows [x_lo, x_lo + len) x ``x_cols`` of the operand whose forward
    pieces are ``x16`` (split16(..., x_scale, H_A)): one fp16 GEMM over 3 x rows, then the
    per-column scales of D - per row of out - and 1 / x_scale come back out."""
    d16 = d16[d_rows]
    rows, _, m = d16.shape
    lhs = d16.reshape(rows * 3, m).t()
    rhs = x16.buf.view(x16.rows * 3, x16.cols)[3 * x_lo:3 * (x_lo + rows), x_cols]
    tmp = torch.mm(lhs, rhs, out_dtype=F32)
    if x_col_inv is not None:           # (x scaled per column: `ColScaled`)
        tmp.mul_(x_col_inv[x_cols].view(1, -1))
    return hip.rescale_rows(tmp, inv, 1.0 / x_scale, out, accumulate=True)


def dgrad16(d2d, wt16, w_scale, out=None, before_gemm=None):
    """out[R, N] = D[R, K] . W[K, N] from the fp16 pieces of W^T [N, 3, K] (order H_B, scale
    w_scale): D is split with a scale per ROW found on the fly; the row scales and 1 / w_scale
    come back out of the product in place.  ``before_gemm()`` runs between the split pass and the
    library GEMM (the caller's "one library GEMM in flight" wait: the split - an HBM-bound own
    kernel - may run beside whatever that wait is for)."""
    d16, inv = hip.split_f16_rows(d2d, H_A)
    if before_gemm is not None:
        before_gemm()
    rows, _, k = d16.shape
    tmp = torch.mm(d16.view(rows, 3 * k), wt16.concat().t(), out_dtype=F32) if out is None else \
        torch.mm(d16.view(rows, 3 * k), wt16.concat().t(), out_dtype=F32, out=out)
    return hip.rescale_rows(tmp, inv, 1.0 / w_scale, tmp, accumulate=False)


def worthwhile(m, k, n):
    """Shapes the split pays for: a big product (the splits are HBM passes over the operands)
    whose dimensions suit the 8-element vectors of the split kernel."""
    return k % 8 == 0 and n % 8 == 0 and m >= 256 and k >= 256 and n >= 256 and \
        m * k * n >= (1 << 31)
