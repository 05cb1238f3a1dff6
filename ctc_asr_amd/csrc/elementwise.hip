// HBM-bound elementwise pieces: fused bias + ReLU + clip + dropout epilogue (fwd / bwd with the
// bias-gradient column sum folded in), batched transpose, TensorFlow-form Adam over flat arenas.
// All are sized for >> 256 workgroups and 16-byte accesses where the layout allows.
#include "common.h"

namespace {

// counter-based uniform in [0, 1): splitmix64 finaliser over (seed, element index)
__device__ __forceinline__ float uniform01(uint64_t seed, uint64_t idx) {
    uint64_t z = seed + 0x9E3779B97F4A7C15ull * (idx + 1);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z ^= z >> 31;
    return (float)(z >> 40) * (1.0f / 16777216.0f);
}

__global__ void __launch_bounds__(256)
bias_act_fwd_kernel(float *__restrict__ y, const float *__restrict__ bias, int64_t n, int cols,
                    float cutoff, float rate, float inv_keep, uint64_t seed) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        float v = y[i];
        if (bias) v += bias[i % cols];
        if (cutoff > 0.f) {
            v = fminf(fmaxf(v, 0.f), cutoff);
            if (rate > 0.f) v = uniform01(seed, (uint64_t)i) >= rate ? v * inv_keep : 0.f;
        }
        y[i] = v;
    }
}

// 16-byte variant for cols % 4 == 0
__global__ void __launch_bounds__(256)
bias_act_fwd_kernel_v4(float4 *__restrict__ y, const float4 *__restrict__ bias, int64_t n4,
                       int cols4, float cutoff, float rate, float inv_keep, uint64_t seed) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
        float4 v = y[i];
        if (bias) {
            float4 bb = bias[i % cols4];
            v.x += bb.x; v.y += bb.y; v.z += bb.z; v.w += bb.w;
        }
        if (cutoff > 0.f) {
            v.x = fminf(fmaxf(v.x, 0.f), cutoff); v.y = fminf(fmaxf(v.y, 0.f), cutoff);
            v.z = fminf(fmaxf(v.z, 0.f), cutoff); v.w = fminf(fmaxf(v.w, 0.f), cutoff);
            if (rate > 0.f) {
                const uint64_t e = (uint64_t)i * 4;
                v.x = uniform01(seed, e + 0) >= rate ? v.x * inv_keep : 0.f;
                v.y = uniform01(seed, e + 1) >= rate ? v.y * inv_keep : 0.f;
                v.z = uniform01(seed, e + 2) >= rate ? v.z * inv_keep : 0.f;
                v.w = uniform01(seed, e + 3) >= rate ? v.w * inv_keep : 0.f;
            }
        }
        y[i] = v;
    }
}

// out = in * mask / keep with the mask regenerated from (seed, index): the same call serves the
// forward pass (activations) and the backward pass (gradients).
__global__ void __launch_bounds__(256)
dropout_kernel(const float *__restrict__ in, float *__restrict__ out, int64_t n, float rate,
               float inv_keep, uint64_t seed) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
        out[i] = uniform01(seed, (uint64_t)i) >= rate ? in[i] * inv_keep : 0.f;
}

// dz = dy * [0 < y < cutoff / keep] / keep, and dbias[c] += column sums.
// grid = (ceil(cols / 64), row chunks); each wave strides over the rows of its chunk.  Narrow
// matrices (cols divides 64: the conv channels, C = 32) pack 64 / cols rows into one wave access
// so that every lane works and a wave still touches 256 contiguous bytes.
__global__ void __launch_bounds__(256)
bias_act_bwd_kernel(const float *__restrict__ y, const float *__restrict__ dy,
                    float *__restrict__ dz, float *__restrict__ dbias, int64_t rows, int cols,
                    float upper, float inv_keep, int64_t rows_per_block) {
    __shared__ float part[4][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const bool narrow = cols < 64 && 64 % cols == 0;
    const int rpw = narrow ? 64 / cols : 1;                  // rows per wave access
    const int c = narrow ? lane % cols : blockIdx.x * 64 + lane;
    const int64_t r0 = (int64_t)blockIdx.y * rows_per_block;
    const int64_t r1 = r0 + rows_per_block < rows ? r0 + rows_per_block : rows;
    float sum = 0.f;
    if (c < cols) {
        for (int64_t r = r0 + wave * rpw + (narrow ? lane / cols : 0); r < r1; r += 4 * rpw) {
            const int64_t i = r * cols + c;
            float g = dy[i];
            if (y) {
                const float v = y[i];
                g = (v > 0.f && v < upper) ? g * inv_keep : 0.f;
            }
            if (dz) dz[i] = g;
            sum += g;
        }
    }
    if (!dbias) return;
    part[wave][lane] = sum;
    __syncthreads();
    if (wave == 0 && lane < (narrow ? cols : 64) && c < cols) {
        float total = 0.f;
        for (int j = lane; j < 64; j += narrow ? cols : 64)
            total += part[0][j] + part[1][j] + part[2][j] + part[3][j];
        atomicAdd(&dbias[c], total);
    }
}

__global__ void __launch_bounds__(256)
transpose_kernel(const float *__restrict__ in, float *__restrict__ out, int rows, int cols) {
    __shared__ float tile[32][33];
    const size_t base = (size_t)blockIdx.z * rows * cols;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 32 x 8
    const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
    for (int j = ty; j < 32; j += 8) {
        int r = r0 + j, c = c0 + tx;
        if (r < rows && c < cols) tile[j][tx] = in[base + (size_t)r * cols + c];
    }
    __syncthreads();
    for (int j = ty; j < 32; j += 8) {
        int c = c0 + j, r = r0 + tx;
        if (r < rows && c < cols) out[base + (size_t)c * rows + r] = tile[tx][j];
    }
}

__global__ void __launch_bounds__(256)
adam_kernel(float *__restrict__ p, const float *__restrict__ g, float *__restrict__ m,
            float *__restrict__ v, int64_t n, float lr_t, float b1, float b2, float eps,
            float gscale, const int32_t *__restrict__ skip) {
    // a step whose gradients are known to be garbage (infeasible CTC alignment, non-finite loss,
    // a persistent recurrence that gave up at a barrier: ctcasr_step_guard) must not touch the
    // parameters or the moments - decided on the device, the host finds out later
    if (skip && *skip) return;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    const int64_t n4 = n >> 2;
    float4 *p4 = reinterpret_cast<float4 *>(p);
    const float4 *g4 = reinterpret_cast<const float4 *>(g);
    float4 *m4 = reinterpret_cast<float4 *>(m);
    float4 *v4 = reinterpret_cast<float4 *>(v);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
        float4 pp = p4[i], gg = g4[i], mm = m4[i], vv = v4[i];
#define ADAM_LANE(f)                                                      \
        {                                                                 \
            float gr = gg.f * gscale;                                     \
            mm.f = b1 * mm.f + (1.f - b1) * gr;                           \
            vv.f = b2 * vv.f + (1.f - b2) * gr * gr;                      \
            pp.f -= lr_t * mm.f / (sqrtf(vv.f) + eps);                    \
        }
        ADAM_LANE(x) ADAM_LANE(y) ADAM_LANE(z) ADAM_LANE(w)
#undef ADAM_LANE
        p4[i] = pp; m4[i] = mm; v4[i] = vv;
    }
    // tail (n % 4 elements)
    for (int64_t i = (n4 << 2) + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += stride) {
        float gr = g[i] * gscale;
        float mm = b1 * m[i] + (1.f - b1) * gr;
        float vv = b2 * v[i] + (1.f - b2) * gr * gr;
        m[i] = mm; v[i] = vv;
        p[i] -= lr_t * mm / (sqrtf(vv) + eps);
    }
}

// skip[0] = any CTC status word != 0 | any per-utterance loss not finite | any time-out word set;
// skip[1] = the time-out words or-ed together, bit 30 = the weight-gradient kernel's give-up word
// (what the host polls, see engine.Trainer)
__global__ void step_guard_kernel(const int32_t *__restrict__ status,
                                  const float *__restrict__ loss, int batch,
                                  const unsigned *err0, const unsigned *err1,
                                  const int32_t *wgrad, int32_t *__restrict__ skip) {
    int bad = 0;
    for (int b = threadIdx.x; b < batch; b += blockDim.x) {
        if (status && status[b] != 0) bad = 1;
        if (loss && !isfinite(loss[b])) bad = 1;
    }
    bad = __any(bad) ? 1 : 0;
    __shared__ int any_bad[4];
    if ((threadIdx.x & 63) == 0) any_bad[threadIdx.x >> 6] = bad;
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned err = (err0 ? *err0 : 0u) | (err1 ? *err1 : 0u) |
                             (wgrad && *wgrad != 0 ? 1u << 30 : 0u);
        skip[0] = (any_bad[0] | any_bad[1] | any_bad[2] | any_bad[3] | (err != 0u)) ? 1 : 0;
        skip[1] = (int32_t)err;
    }
}

// largest magnitude of a vector, as its bit pattern (the caller zeroes the word)
__global__ void __launch_bounds__(256)
absmax_kernel(const float *__restrict__ x, int64_t n, unsigned *__restrict__ out) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    float m = 0.f;
    const float4 *x4 = reinterpret_cast<const float4 *>(x);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < (n >> 2); i += stride) {
        const float4 v = x4[i];
        m = fmaxf(fmaxf(m, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
    }
    for (int64_t i = ((n >> 2) << 2) + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += stride)
        m = fmaxf(m, fabsf(x[i]));
    m = wave_max(m);
    // (NaN compares false everywhere above: a NaN weight shows as a non-finite loss, not here)
    // one atomic per workgroup: thousands of them on one word serialise in the L2 (the version
    // with one per wave took 100 us for 67 MB, 0.7 TB/s)
    __shared__ float wmax[4];
    if ((threadIdx.x & 63) == 0) wmax[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0)
        atomicMax(out, __float_as_uint(fmaxf(fmaxf(wmax[0], wmax[1]), fmaxf(wmax[2], wmax[3]))));
}

// Diagnostic stand-in for the CU footprint of a collective: `workgroups` workgroups of 256 threads
// that hold their CUs for `busy_ticks` of the 100 MHz wall clock and touch nothing.  RCCL's ring
// kernels are exactly that from the point of view of a persistent recurrence launch - resident
// workgroups that do not yield - and NCCL refuses two ranks on one device, so the one-GPU boxes
// cannot run the real thing beside the recurrences (tests/test_gpu_dp_rccl.py, bench.py
// --collective-stand-in).
__global__ void __launch_bounds__(256) occupy_kernel(unsigned long long busy_ticks) {
    const unsigned long long start = wall_clock64();
    while (wall_clock64() - start < busy_ticks) __builtin_amdgcn_s_sleep(32);
}

// ... and with a ring all-reduce's memory traffic as well: the workgroups stream `total4` float4
// sums a[i] += b[i] through a scratch pair (16 B read twice, written once per element), paced so
// that the whole launch takes `busy_ticks` - resident like a collective's ring kernels, but loading
// the L2s, the fabric and HBM the way a reduce-scatter + all-gather over the payload does.
__global__ void __launch_bounds__(256) collective_traffic_kernel(float4 *a, const float4 *b,
                                                                 long long n4, long long total4,
                                                                 unsigned long long busy_ticks) {
    const unsigned long long start = wall_clock64();
    const long long share = total4 / gridDim.x;
    const long long first = (long long)blockIdx.x * share;
    constexpr long long SLAB = 256 * 16;                // float4 per pacing slab (64 KB)
    for (long long base = 0; base < share; base += SLAB) {
#pragma unroll 4
        for (long long i = base + threadIdx.x; i < min(base + SLAB, share); i += 256) {
            const long long idx = (first + i) % n4;
            const float4 x = a[idx], y = b[idx];
            a[idx] = make_float4(x.x + y.x, x.y + y.y, x.z + y.z, x.w + y.w);
        }
        const unsigned long long due = (unsigned long long)(
            (double)busy_ticks * (double)(base + SLAB) / (double)share);
        while (wall_clock64() - start < due) __builtin_amdgcn_s_sleep(8);
    }
    while (wall_clock64() - start < busy_ticks) __builtin_amdgcn_s_sleep(32);
}

int grid_for(int64_t work_items) {
    int64_t blocks = (work_items + 255) / 256;
    if (blocks > 2048) blocks = 2048;   // 256 CUs x 8, grid-stride the rest
    if (blocks < 1) blocks = 1;
    return (int)blocks;
}

}  // namespace

extern "C" int ctcasr_bias_act_fwd(float *y, const float *bias, int64_t rows, int cols,
                                   float cutoff, float dropout_rate, uint64_t seed,
                                   ctcasr_stream_t stream) {
    if (!y || rows < 0 || cols <= 0 || dropout_rate < 0.f || dropout_rate >= 1.f)
        return CTCASR_ERR_BAD_ARGUMENT;
    if (rows == 0) return CTCASR_OK;
    const int64_t n = rows * cols;
    const float inv_keep = 1.f / (1.f - dropout_rate);
    hipStream_t s = (hipStream_t)stream;
    const bool aligned = (reinterpret_cast<uintptr_t>(y) % 16 == 0) &&
                         (!bias || reinterpret_cast<uintptr_t>(bias) % 16 == 0);
    if (cols % 4 == 0 && aligned) {
        bias_act_fwd_kernel_v4<<<grid_for(n / 4), 256, 0, s>>>(
            reinterpret_cast<float4 *>(y), reinterpret_cast<const float4 *>(bias), n / 4,
            cols / 4, cutoff, dropout_rate, inv_keep, seed);
    } else {
        bias_act_fwd_kernel<<<grid_for(n), 256, 0, s>>>(y, bias, n, cols, cutoff, dropout_rate,
                                                        inv_keep, seed);
    }
    return ctcasr_launch_status();
}

static int launch_bwd(const float *y, const float *dy, float *dz, float *dbias, int64_t rows,
                      int cols, float upper, float inv_keep, hipStream_t s) {
    const int col_blocks = (cols + 63) / 64;
    int64_t row_blocks = (2048 + col_blocks - 1) / col_blocks;
    if (row_blocks > (rows + 15) / 16) row_blocks = (rows + 15) / 16;
    if (row_blocks < 1) row_blocks = 1;
    if (row_blocks > 65535) row_blocks = 65535;
    const int64_t rows_per_block = (rows + row_blocks - 1) / row_blocks;
    dim3 grid(col_blocks, (unsigned)((rows + rows_per_block - 1) / rows_per_block));
    bias_act_bwd_kernel<<<grid, 256, 0, s>>>(y, dy, dz, dbias, rows, cols, upper, inv_keep,
                                            rows_per_block);
    return ctcasr_launch_status();
}

extern "C" int ctcasr_bias_act_bwd(const float *y, const float *dy, float *dz, float *dbias,
                                   int64_t rows, int cols, float cutoff, float dropout_rate,
                                   ctcasr_stream_t stream) {
    if (!y || !dy || !dz || rows < 0 || cols <= 0 || cutoff <= 0.f || dropout_rate < 0.f ||
        dropout_rate >= 1.f)
        return CTCASR_ERR_BAD_ARGUMENT;
    if (rows == 0) return CTCASR_OK;
    const float inv_keep = 1.f / (1.f - dropout_rate);
    return launch_bwd(y, dy, dz, dbias, rows, cols, cutoff * inv_keep, inv_keep,
                      (hipStream_t)stream);
}

extern "C" int ctcasr_colsum_accumulate(const float *dz, float *dbias, int64_t rows, int cols,
                                        ctcasr_stream_t stream) {
    if (!dz || !dbias || rows < 0 || cols <= 0) return CTCASR_ERR_BAD_ARGUMENT;
    if (rows == 0) return CTCASR_OK;
    return launch_bwd(nullptr, dz, nullptr, dbias, rows, cols, 0.f, 1.f, (hipStream_t)stream);
}

extern "C" int ctcasr_dropout(const float *in, float *out, int64_t n, float dropout_rate,
                              uint64_t seed, ctcasr_stream_t stream) {
    if (!in || !out || n < 0 || dropout_rate < 0.f || dropout_rate >= 1.f)
        return CTCASR_ERR_BAD_ARGUMENT;
    if (n == 0) return CTCASR_OK;
    dropout_kernel<<<grid_for(n), 256, 0, (hipStream_t)stream>>>(
        in, out, n, dropout_rate, 1.f / (1.f - dropout_rate), seed);
    return ctcasr_launch_status();
}

extern "C" int ctcasr_transpose_batched(const float *in, float *out, int batch, int rows, int cols,
                                        ctcasr_stream_t stream) {
    if (!in || !out || batch <= 0 || rows <= 0 || cols <= 0) return CTCASR_ERR_BAD_ARGUMENT;
    dim3 grid((cols + 31) / 32, (rows + 31) / 32, batch);
    transpose_kernel<<<grid, 256, 0, (hipStream_t)stream>>>(in, out, rows, cols);
    return ctcasr_launch_status();
}

extern "C" int ctcasr_adam_step(float *param, const float *grad, float *m, float *v, int64_t n,
                                float lr, float beta1, float beta2, float epsilon, int64_t step,
                                float grad_scale, const int32_t *skip, ctcasr_stream_t stream) {
    if (!param || !grad || !m || !v || n < 0 || step < 1) return CTCASR_ERR_BAD_ARGUMENT;
    if (n == 0) return CTCASR_OK;
    if ((reinterpret_cast<uintptr_t>(param) | reinterpret_cast<uintptr_t>(grad) |
         reinterpret_cast<uintptr_t>(m) | reinterpret_cast<uintptr_t>(v)) % 16 != 0)
        return CTCASR_ERR_BAD_ARGUMENT;
    const double lr_t = (double)lr * sqrt(1.0 - pow((double)beta2, (double)step)) /
                        (1.0 - pow((double)beta1, (double)step));
    // grid-stride loop over float4s: many short workgroups keep more of the seven streams' requests
    // in flight than 8 per CU do (122 M parameters, tools/adam_probe.py: 2048 workgroups 0.71 ms =
    // 4.8 TB/s, 16384 0.61, 65536 0.57 = 6.0 TB/s of HBM3E's 8, 131072 0.59)
#ifndef ADAM_BLOCKS
#define ADAM_BLOCKS 65536
#endif
    int64_t want = (n / 4 + 1 + 255) / 256;
    const int blocks = (int)(want > ADAM_BLOCKS ? ADAM_BLOCKS : want < 1 ? 1 : want);
    adam_kernel<<<blocks, 256, 0, (hipStream_t)stream>>>(
        param, grad, m, v, n, (float)lr_t, beta1, beta2, epsilon, grad_scale, skip);
    return ctcasr_launch_status();
}

extern "C" int ctcasr_step_guard(const int32_t *ctc_status, const float *per_utterance_loss,
                                 int batch, const uint32_t *timeout_word0,
                                 const uint32_t *timeout_word1, const int32_t *wgrad_word,
                                 int32_t *skip, ctcasr_stream_t stream) {
    if (!skip || batch < 0) return CTCASR_ERR_BAD_ARGUMENT;
    step_guard_kernel<<<1, 256, 0, (hipStream_t)stream>>>(ctc_status, per_utterance_loss, batch,
                                                          timeout_word0, timeout_word1, wgrad_word,
                                                          skip);
    return ctcasr_launch_status();
}

extern "C" int ctcasr_occupy_cus(int workgroups, int busy_us, ctcasr_stream_t stream) {
    if (workgroups < 1 || workgroups > 256 || busy_us < 0 || busy_us > 100000)
        return CTCASR_ERR_BAD_ARGUMENT;
    occupy_kernel<<<workgroups, 256, 0, (hipStream_t)stream>>>((unsigned long long)busy_us * 100ull);
    return ctcasr_launch_status();
}

extern "C" int ctcasr_collective_traffic(int workgroups, int busy_us, float *a, const float *b,
                                         int64_t scratch_floats, int64_t payload_bytes,
                                         ctcasr_stream_t stream) {
    if (workgroups < 1 || workgroups > 256 || busy_us < 0 || busy_us > 100000 || !a || !b ||
        scratch_floats < 1024 || scratch_floats % 4 != 0 || payload_bytes < 0 ||
        (reinterpret_cast<uintptr_t>(a) | reinterpret_cast<uintptr_t>(b)) % 16 != 0)
        return CTCASR_ERR_BAD_ARGUMENT;
    // two phases (reduce-scatter, all-gather), each one pass over the payload
    const long long total4 = 2 * (payload_bytes / 16);
    collective_traffic_kernel<<<workgroups, 256, 0, (hipStream_t)stream>>>(
        reinterpret_cast<float4 *>(a), reinterpret_cast<const float4 *>(b), scratch_floats / 4,
        total4, (unsigned long long)busy_us * 100ull);
    return ctcasr_launch_status();
}

extern "C" int ctcasr_absmax(const float *x, int64_t n, uint32_t *max_bits,
                             ctcasr_stream_t stream) {
    if (!x || !max_bits || n < 0 || reinterpret_cast<uintptr_t>(x) % 16 != 0)
        return CTCASR_ERR_BAD_ARGUMENT;
    if (n == 0) return CTCASR_OK;
    const int blocks = grid_for(n / 4 + 1);
    absmax_kernel<<<blocks < 1024 ? blocks : 1024, 256, 0, (hipStream_t)stream>>>(x, n, max_bits);
    return ctcasr_launch_status();
}
