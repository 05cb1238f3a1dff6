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
        // (saturating: an operand outside the range its scale was chosen for becomes the largest
        // finite fp16, not inf - wrong, but it cannot poison a whole product with NaNs; callers
        // that cannot bound the operand check it, ctcasr_absmax)
        const float s = fminf(fmaxf(f[i] * scale, -65504.f), 65504.f);
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

// ---- per-column / per-row power-of-two scales for operands without a bound (gradients) ---------
// scale = 2^(13 - e) with e = exponent of the column's (row's) largest magnitude: the largest
// element lands in [2^13, 2^14); an all-zero column gets scale 1.
__device__ __forceinline__ float scale_for_max(unsigned max_bits) {
    const int e = (int)((max_bits >> 23) & 0xFF) - 127;
    if (max_bits == 0u) return 1.0f;
    const int se = min(max(13 - e, -126), 127);
    return __uint_as_float((unsigned)(se + 127) << 23);
}

__global__ void __launch_bounds__(256)
colmax_kernel(const float *__restrict__ x, int64_t rows, int cols, int64_t ld_x, int rows_per_block,
              unsigned *__restrict__ max_bits) {
    const int c4 = blockIdx.x * 256 + threadIdx.x;
    if (c4 * 4 >= cols) return;
    const int64_t r0 = (int64_t)blockIdx.y * rows_per_block;
    const int64_t r1 = min(rows, r0 + rows_per_block);
    float m[4] = {0.f, 0.f, 0.f, 0.f};
    for (int64_t r = r0; r < r1; ++r) {
        const float4 v = *reinterpret_cast<const float4 *>(x + r * ld_x + 4 * c4);
        m[0] = fmaxf(m[0], fabsf(v.x));
        m[1] = fmaxf(m[1], fabsf(v.y));
        m[2] = fmaxf(m[2], fabsf(v.z));
        m[3] = fmaxf(m[3], fabsf(v.w));
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) atomicMax(max_bits + 4 * c4 + i, __float_as_uint(m[i]));
}

__global__ void __launch_bounds__(256)
colscale_kernel(const unsigned *__restrict__ max_bits, int cols, float *__restrict__ scale,
                float *__restrict__ inv) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= cols) return;
    const float s = scale_for_max(max_bits[c]);
    scale[c] = s;
    inv[c] = 1.0f / s;
}

// two fp16 pieces of x[r][c] * col_scale[c] (col_scale may be null) * scale
__global__ void __launch_bounds__(256)
split_f16_cols_kernel(const float *__restrict__ x, int64_t ld_x, int64_t vecs, int vec_per_row,
                      const float *__restrict__ col_scale, float scale, int blocks,
                      SplitOrder order, uint4 *__restrict__ out, int64_t ld_out8,
                      int64_t block_stride8) {
    const int64_t v = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (v >= vecs) return;
    const int64_t row = v / vec_per_row;
    const int c8 = (int)(v - row * vec_per_row);
    const float4 *src = reinterpret_cast<const float4 *>(x + row * ld_x) + 2 * c8;
    const float4 lo = src[0], hi = src[1];
    const float4 s0 = reinterpret_cast<const float4 *>(col_scale)[2 * c8];
    const float4 s1 = reinterpret_cast<const float4 *>(col_scale)[2 * c8 + 1];
    const float f[8] = {lo.x * s0.x, lo.y * s0.y, lo.z * s0.z, lo.w * s0.w,
                        hi.x * s1.x, hi.y * s1.y, hi.z * s1.z, hi.w * s1.w};
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

// one workgroup per row: largest magnitude of the row -> its scale (inverse written out), then the
// two fp16 pieces of the scaled row.  cols <= 256 * 8 * SPLIT_ROW_VECS.
constexpr int SPLIT_ROW_VECS = 8;
__global__ void __launch_bounds__(256)
split_f16_rows_kernel(const float *__restrict__ x, int64_t ld_x, int cols, int blocks,
                      SplitOrder order, uint4 *__restrict__ out, int64_t ld_out8,
                      int64_t block_stride8, float *__restrict__ inv_scale) {
    __shared__ float wave_top[4];
    const int64_t row = blockIdx.x;
    const int vec_per_row = cols / 8;
    float f[SPLIT_ROW_VECS][8];
    float m = 0.f;
#pragma unroll
    for (int u = 0; u < SPLIT_ROW_VECS; ++u) {
        const int c8 = threadIdx.x + 256 * u;
        if (c8 < vec_per_row) {
            const float4 *src = reinterpret_cast<const float4 *>(x + row * ld_x) + 2 * c8;
            const float4 lo = src[0], hi = src[1];
            f[u][0] = lo.x; f[u][1] = lo.y; f[u][2] = lo.z; f[u][3] = lo.w;
            f[u][4] = hi.x; f[u][5] = hi.y; f[u][6] = hi.z; f[u][7] = hi.w;
#pragma unroll
            for (int i = 0; i < 8; ++i) m = fmaxf(m, fabsf(f[u][i]));
        }
    }
    m = wave_max(m);
    if ((threadIdx.x & 63) == 0) wave_top[threadIdx.x >> 6] = m;
    __syncthreads();
    m = fmaxf(fmaxf(wave_top[0], wave_top[1]), fmaxf(wave_top[2], wave_top[3]));
    const float scale = scale_for_max(__float_as_uint(m));
    if (threadIdx.x == 0) inv_scale[row] = 1.0f / scale;
#pragma unroll
    for (int u = 0; u < SPLIT_ROW_VECS; ++u) {
        const int c8 = threadIdx.x + 256 * u;
        if (c8 >= vec_per_row) continue;
        unsigned h[2][8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const float s = f[u][i] * scale;
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
        for (int b = 0; b < blocks; ++b)
            dst[b * block_stride8] = order.piece[b] == 0 ? packed[0] : packed[1];
    }
}

// out[r][c] (+)= t[r][c] * row_factor[r] * alpha
__global__ void __launch_bounds__(256)
rescale_rows_kernel(const float *__restrict__ t, int64_t ld_t, const float *__restrict__ row_factor,
                    float alpha, float *__restrict__ out, int64_t ld_out, int64_t vecs,
                    int vec_per_row, int accumulate) {
    const int64_t v = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (v >= vecs) return;
    const int64_t row = v / vec_per_row;
    const int c4 = (int)(v - row * vec_per_row);
    const float k = row_factor[row] * alpha;
    const float4 a = reinterpret_cast<const float4 *>(t + row * ld_t)[c4];
    float4 *dst = reinterpret_cast<float4 *>(out + row * ld_out) + c4;
    float4 r = make_float4(a.x * k, a.y * k, a.z * k, a.w * k);
    if (accumulate) {
        const float4 o = *dst;
        r = make_float4(r.x + o.x, r.y + o.y, r.z + o.z, r.w + o.w);
    }
    *dst = r;
}

static bool split_args_ok(const void *x, int64_t rows, int cols, int64_t ld_x, const int *order,
                          int blocks, const void *out, int64_t ld_out, int64_t block_stride,
                          SplitOrder *o) {
    if (!x || !out || !order || rows < 0 || cols <= 0 || cols % 8 != 0 || ld_x < cols ||
        ld_x % 4 != 0 || blocks < 1 || blocks > CTCASR_SPLIT_MAX_BLOCKS || ld_out < cols ||
        ld_out % 8 != 0 || block_stride < cols || block_stride % 8 != 0)
        return false;
    if ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(out)) % 16 != 0) return false;
    for (int b = 0; b < CTCASR_SPLIT_MAX_BLOCKS; ++b) {
        o->piece[b] = b < blocks ? order[b] : 0;
        if (o->piece[b] < 0 || o->piece[b] > 1) return false;
    }
    return true;
}

}  // namespace

extern "C" int ctcasr_colmax_scale(const float *x, int64_t rows, int cols, int64_t ld_x,
                                   void *workspace, float *scale, float *inv_scale,
                                   ctcasr_stream_t stream) {
    if (!x || !workspace || !scale || !inv_scale || rows < 0 || cols <= 0 || cols % 4 != 0 ||
        ld_x < cols || ld_x % 4 != 0 || reinterpret_cast<uintptr_t>(x) % 16 != 0)
        return CTCASR_ERR_BAD_ARGUMENT;
    hipStream_t s = (hipStream_t)stream;
    unsigned *bits = reinterpret_cast<unsigned *>(workspace);
    if (hipMemsetAsync(bits, 0, sizeof(unsigned) * (size_t)cols, s) != hipSuccess)
        return CTCASR_ERR_LAUNCH;
    if (rows > 0) {
        const int rows_per_block = 128;
        dim3 grid((cols / 4 + 255) / 256, (unsigned)((rows + rows_per_block - 1) / rows_per_block));
        colmax_kernel<<<grid, 256, 0, s>>>(x, rows, cols, ld_x, rows_per_block, bits);
    }
    colscale_kernel<<<(cols + 255) / 256, 256, 0, s>>>(bits, cols, scale, inv_scale);
    return ctcasr_launch_status();
}

// The scales of ctcasr_colmax_scale from column maxima somebody else has found (the fp16 backward
// recurrence kernel accumulates them while it writes dxw): no pass over the matrix.
extern "C" int ctcasr_colscale_from_max(const uint32_t *max_bits, int cols, float *scale,
                                        float *inv_scale, ctcasr_stream_t stream) {
    if (!max_bits || !scale || !inv_scale || cols <= 0) return CTCASR_ERR_BAD_ARGUMENT;
    colscale_kernel<<<(cols + 255) / 256, 256, 0, (hipStream_t)stream>>>(max_bits, cols, scale,
                                                                         inv_scale);
    return ctcasr_launch_status();
}

extern "C" int ctcasr_split_f16_cols(const float *x, int64_t rows, int cols, int64_t ld_x,
                                     const float *col_scale, float scale, const int *order,
                                     int blocks, void *out, int64_t ld_out, int64_t block_stride,
                                     ctcasr_stream_t stream) {
    SplitOrder o;
    if (!split_args_ok(x, rows, cols, ld_x, order, blocks, out, ld_out, block_stride, &o) ||
        !col_scale || reinterpret_cast<uintptr_t>(col_scale) % 16 != 0 || !(scale > 0.f))
        return CTCASR_ERR_BAD_ARGUMENT;
    if (rows == 0) return CTCASR_OK;
    const int vec_per_row = cols / 8;
    const int64_t vecs = rows * vec_per_row;
    split_f16_cols_kernel<<<(unsigned)((vecs + 255) / 256), 256, 0, (hipStream_t)stream>>>(
        x, ld_x, vecs, vec_per_row, col_scale, scale, blocks, o, reinterpret_cast<uint4 *>(out),
        ld_out / 8, block_stride / 8);
    return ctcasr_launch_status();
}

extern "C" int ctcasr_split_f16_rows(const float *x, int64_t rows, int cols, int64_t ld_x,
                                     const int *order, int blocks, void *out, int64_t ld_out,
                                     int64_t block_stride, float *inv_scale,
                                     ctcasr_stream_t stream) {
    SplitOrder o;
    if (!split_args_ok(x, rows, cols, ld_x, order, blocks, out, ld_out, block_stride, &o) ||
        !inv_scale || cols > 256 * 8 * SPLIT_ROW_VECS)
        return CTCASR_ERR_BAD_ARGUMENT;
    if (rows == 0) return CTCASR_OK;
    split_f16_rows_kernel<<<(unsigned)rows, 256, 0, (hipStream_t)stream>>>(
        x, ld_x, cols, blocks, o, reinterpret_cast<uint4 *>(out), ld_out / 8, block_stride / 8,
        inv_scale);
    return ctcasr_launch_status();
}

extern "C" int ctcasr_rescale_rows(const float *t, int64_t ld_t, const float *row_factor,
                                   float alpha, float *out, int64_t ld_out, int64_t rows, int cols,
                                   int accumulate, ctcasr_stream_t stream) {
    if (!t || !row_factor || !out || rows < 0 || cols <= 0 || cols % 4 != 0 || ld_t < cols ||
        ld_out < cols || ld_t % 4 != 0 || ld_out % 4 != 0 ||
        (reinterpret_cast<uintptr_t>(t) | reinterpret_cast<uintptr_t>(out)) % 16 != 0)
        return CTCASR_ERR_BAD_ARGUMENT;
    if (rows == 0) return CTCASR_OK;
    const int vec_per_row = cols / 4;
    const int64_t vecs = rows * vec_per_row;
    rescale_rows_kernel<<<(unsigned)((vecs + 255) / 256), 256, 0, (hipStream_t)stream>>>(
        t, ld_t, row_factor, alpha, out, ld_out, vecs, vec_per_row, accumulate);
    return ctcasr_launch_status();
}

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
