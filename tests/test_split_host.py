"""Host-side facts the bf16-split GEMMs rest on (no GPU): the block orders pair up as the six
products of order <= 2, and those six products of exact three-piece splits reproduce an fp32
product to ~2^-24 - restated in numpy with bit-level bfloat16 rounding."""

import numpy as np

from ctc_asr_amd import split_gemm


def _bf16_rne(x):
    """float32 array -> float32 array holding round-to-nearest-even bfloat16 values."""
    u = np.ascontiguousarray(x, dtype=np.float32).view(np.uint32).astype(np.uint64)
    u = (u + 0x7FFF + ((u >> 16) & 1)) >> 16 << 16
    return (u & 0xFFFFFFFF).astype(np.uint32).view(np.float32)


def _split3(x):
    pieces, r = [], np.asarray(x, dtype=np.float32)
    for _ in range(3):
        p = _bf16_rne(r)
        pieces.append(p)
        r = (r - p).astype(np.float32)
    return pieces


def test_block_orders_pair_up_as_the_six_products():
    pairs = sorted(zip(split_gemm.A_ORDER, split_gemm.B_ORDER))
    assert pairs == sorted(split_gemm.PAIRS)
    assert sorted(i + j for i, j in pairs) == [0, 1, 1, 2, 2, 2]       # every term of order <= 2
    # the ranges `mm_nt_by_order` multiplies: blocks [0,3) order 2, [3,5) order 1, [5,6) order 0
    orders = [i + j for i, j in zip(split_gemm.A_ORDER, split_gemm.B_ORDER)]
    assert orders == [2, 2, 2, 1, 1, 0]
    # `Split.piece` finds piece p of either order in the block that holds it
    for order in (split_gemm.A_ORDER, split_gemm.B_ORDER):
        for p in range(3):
            assert order[split_gemm._BLOCK_OF_PIECE[order][p]] == p


def test_three_pieces_are_exact_and_six_products_are_fp32_grade():
    rng = np.random.default_rng(0)
    a = (rng.normal(size=(64, 512)) * np.exp(rng.uniform(-20, 10, size=(64, 1)))).astype(np.float32)
    b = (rng.normal(size=(512, 48)) / 512 ** 0.5).astype(np.float32)
    ap, bp = _split3(a), _split3(b)
    assert np.array_equal((ap[0].astype(np.float64) + ap[1] + ap[2]).astype(np.float32), a)
    assert np.array_equal((bp[0].astype(np.float64) + bp[1] + bp[2]).astype(np.float32), b)
    ref = a.astype(np.float64) @ b.astype(np.float64)
    six = sum(ap[i].astype(np.float64) @ bp[j].astype(np.float64) for i, j in split_gemm.PAIRS)
    scale = np.abs(a).astype(np.float64) @ np.abs(b).astype(np.float64)
    # dropped terms: a2 b3, a3 b2, a3 b3 - each below 2^-24 of |a| |b|
    assert float((np.abs(six - ref) / scale).max()) < 3 * 2.0 ** -24
    # three products (order <= 1) would not do: 2^-16
    three = sum(ap[i].astype(np.float64) @ bp[j].astype(np.float64)
                for i, j in split_gemm.PAIRS if i + j <= 1)
    assert float((np.abs(three - ref) / scale).max()) > 2.0 ** -20


def test_worthwhile_shapes():
    assert split_gemm.worthwhile(16000, 2048, 8192) and split_gemm.worthwhile(8000, 640, 8192)
    assert not split_gemm.worthwhile(3200, 80, 128)       # the small models of the tests: fp32
    assert not split_gemm.worthwhile(16000, 2044, 8192)   # K not a multiple of the 8-wide vectors
