// fp32 operands for the bf16 matrix pipe: x = x1 + x2 + x3 with three bfloat16 pieces (8 + 8 + 8
// mantissa bits; x1 = rne(x), x2 = rne(x - x1), x3 = rne(x - x1 - x2), both differences exact in
// fp32), written as K-concatenated blocks [rows][blocks][cols] so that ONE bf16 GEMM with fp32
// accumulation over blocks * cols computes the six products of order <= 2 of an fp32 GEMM
// (x1w1; x1w2 + x2w1; x1w3 + x2w2 + x3w1: everything down to 2^-24 relative, like an fp32 FMA
// chain).  gfx950's fp32 MFMA runs at 1/16 of the bf16 rate (no xf32), so six bf16 passes cost
// 6/16 of one fp32 pass.  HBM-bound: 4 bytes read, 2 * blocks written per element.
#include "common.h"

namespace {

struct SplitOrder {
    int piece[CTCASR_SPLIT_MAX_BLOCKS];
};

__device__ __forceinline__ unsigned bf16_rne(float f) {
    unsigned u = __float_as_uint(f);
    u += 0x7FFFu + ((u >> 16) & 1u);
    return u >> 16;
}

// thread -> 8 consecutive columns of one row: two 16-byte loads, one 16-byte store per block
__global__ void __launch_bounds__(256)
split_bf16_kernel(const float *__restrict__ x, int64_t ld_x, int64_t vecs, int vec_per_row,
                  int blocks, SplitOrder order, uint4 *__restrict__ out, int64_t ld_out8,
                  int64_t block_stride8) {
    const int64_t v = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (v >= vecs) return;
    const int64_t row = v / vec_per_row;
    const int c8 = (int)(v - row * vec_per_row);
    const float4 *src = reinterpret_cast<const float4 *>(x + row * ld_x) + 2 * c8;
    const float4 lo = src[0], hi = src[1];
    const float f[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
    unsigned h[3][8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        float r = f[i];
#pragma unroll
        for (int p = 0; p < 3; ++p) {
            h[p][i] = bf16_rne(r);
            r -= __uint_as_float(h[p][i] << 16);
        }
    }
    uint4 packed[3];
#pragma unroll
    for (int p = 0; p < 3; ++p)
        packed[p] = make_uint4(h[p][0] | (h[p][1] << 16), h[p][2] | (h[p][3] << 16),
                               h[p][4] | (h[p][5] << 16), h[p][6] | (h[p][7] << 16));
    uint4 *dst = out + row * ld_out8 + c8;
    for (int b = 0; b < blocks; ++b) {
        const int p = order.piece[b];
        dst[b * block_stride8] = p == 0 ? packed[0] : (p == 1 ? packed[1] : packed[2]);
    }
}

// Two fp16 pieces of x * scale: h1 = rne_f16(x scale), h2 = rne_f16(x scale - h1) - 11 + 11 mantissa
// bits, for operands with a known bound (scale a power of two that keeps |x| scale < 65504)
__global__ void __launch_bounds__(256)
split_f16_kernel(const float *__restrict__ x, int64_t ld_x, int64_t vecs, int vec_per_row,
                 float scale, int blocks, SplitOrder order, uint4 *__restrict__ out,
                 int64_t ld_out8, int64_t block_stride8) {
    const int64_t v = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (v >= vecs) return;
    const int64_t row = v / vec_per_row;
    const int c8 = (int)(v - row * vec_per_row);
    const float4 *src = reinterpret_cast<const float4 *>(x + row * ld_x) + 2 * c8;
    const float4 lo = src[0], hi = src[1];
    const float f[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
    unsigned h[2][8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const float s = f[i] * scale;
        const _Float16 h1 = (_Float16)s;
        const _Float16 h2 = (_Float16)(s - (float)h1);
        h[0][i] = __builtin_bit_cast(unsigned short, h1);
        h[1][i] = __builtin_bit_cast(unsigned short, h2);
    }
    uint4 packed[2];
#pragma unroll
    for (int p = 0; p < 2; ++p)
        packed[p] = make_uint4(h[p][0] | (h[p][1] << 16), h[p][2] | (h[p][3] << 16),
                               h[p][4] | (h[p][5] << 16), h[p][6] | (h[p][7] << 16));
    uint4 *dst = out + row * ld_out8 + c8;
    for (int b = 0; b < blocks; ++b) dst[b * block_stride8] = order.piece[b] == 0 ? packed[0] : packed[1];
}

}  // namespace

extern "C" int ctcasr_split_f16(const float *x, int64_t rows, int cols, int64_t ld_x, float scale,
                                const int *order, int blocks, void *out, int64_t ld_out,
                                int64_t block_stride, ctcasr_stream_t stream) {
    if (!x || !out || !order || rows < 0 || cols <= 0 || cols % 8 != 0 || ld_x < cols ||
        ld_x % 4 != 0 || blocks < 1 || blocks > CTCASR_SPLIT_MAX_BLOCKS || ld_out < cols ||
        ld_out % 8 != 0 || block_stride < cols || block_stride % 8 != 0 || !(scale > 0.f))
        return CTCASR_ERR_BAD_ARGUMENT;
    if ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(out)) % 16 != 0)
        return CTCASR_ERR_BAD_ARGUMENT;
    SplitOrder o;
    for (int b = 0; b < CTCASR_SPLIT_MAX_BLOCKS; ++b) {
        o.piece[b] = b < blocks ? order[b] : 0;
        if (o.piece[b] < 0 || o.piece[b] > 1) return CTCASR_ERR_BAD_ARGUMENT;
    }
    if (rows == 0) return CTCASR_OK;
    const int vec_per_row = cols / 8;
    const int64_t vecs = rows * vec_per_row;
    split_f16_kernel<<<(unsigned)((vecs + 255) / 256), 256, 0, (hipStream_t)stream>>>(
        x, ld_x, vecs, vec_per_row, scale, blocks, o, reinterpret_cast<uint4 *>(out), ld_out / 8,
        block_stride / 8);
    return ctcasr_launch_status();
}

extern "C" int ctcasr_split_bf16(const float *x, int64_t rows, int cols, int64_t ld_x,
                                 const int *order, int blocks, void *out, int64_t ld_out,
                                 int64_t block_stride, ctcasr_stream_t stream) {
    if (!x || !out || !order || rows < 0 || cols <= 0 || cols % 8 != 0 || ld_x < cols ||
        ld_x % 4 != 0 || blocks < 1 || blocks > CTCASR_SPLIT_MAX_BLOCKS || ld_out < cols ||
        ld_out % 8 != 0 || block_stride < cols || block_stride % 8 != 0)
        return CTCASR_ERR_BAD_ARGUMENT;
    if ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(out)) % 16 != 0)
        return CTCASR_ERR_BAD_ARGUMENT;
    SplitOrder o;
    for (int b = 0; b < CTCASR_SPLIT_MAX_BLOCKS; ++b) {
        o.piece[b] = b < blocks ? order[b] : 0;
        if (o.piece[b] < 0 || o.piece[b] > 2) return CTCASR_ERR_BAD_ARGUMENT;
    }
    if (rows == 0) return CTCASR_OK;
    const int vec_per_row = cols / 8;
    const int64_t vecs = rows * vec_per_row;
    split_bf16_kernel<<<(unsigned)((vecs + 255) / 256), 256, 0, (hipStream_t)stream>>>(
        x, ld_x, vecs, vec_per_row, blocks, o, reinterpret_cast<uint4 *>(out), ld_out / 8,
        block_stride / 8);
    return ctcasr_launch_status();
}
