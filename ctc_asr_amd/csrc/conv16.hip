// The convolutions of the DS2 front end (conv.hip) with their products on the fp16 matrix pipe
// (round 4): forward pass, data gradient and kernel gradient of the 11 x 21, stride (1, 2) layers
// over 32 input channels, and - at the end of the file - forward pass and kernel gradient of the
// first layer (1 -> 32 channels, 11 x 41, stride (2, 2)).
//
// conv.hip's kernels run at 110 - 140 TFLOP/s of fp32 MFMA (157 peak): MFMA-issue bound, the last
// fp32-MFMA kernels of the training step.  The forward product's operands are both bounded - the
// input is the output of the previous layer's fused min(max(., 0), relu_cutoff), a kernel's
// weights by their own largest magnitude - so the two-piece fp16 form of DESIGN.md 4.4 / 4.1d
// applies: x s_x = x1 + x2, w s_w = w1 + w2 (x1 = rne_f16(.), x2 = rne_f16(. - x1): 11 + 11 bits),
// y = (x1 w1 + x1 w2 + x2 w1) / (s_x s_w) accumulated in fp32 by v_mfma_f32_16x16x32_f16 (dropped:
// x2 w2 <= 2^-22 |x w|).  One tap (kt, kf) x 32 input channels is exactly ONE K = 32 step: 3 MFMAs
// of ~16 cycles per (M tile, N tile) and tap where the fp32 kernel issues 8 of 32 cycles.
//
// Geometry, patch, parity planes and the tile walk are conv.hip's (one workgroup = one utterance
// x TT output frames x all frequencies x all output channels; a wave owns 5 M tiles of 16 rows).
// What changes: the patch cell keeps its 144-byte pitch but holds [32 first pieces | 32 second
// pieces | pad] - the A fragment of a row (lane: row l & 15, channels 8 (l >> 4) .. + 7) is one
// ds_read_b128 per piece at the address pattern of the fp32 fragments (conflict-free for the same
// reason) - the conversion happens once, while the input is staged; the weights are packed per
// step into fragment order as pieces ([tap][N tile][piece][lane] x 16 bytes: a wave's B fragment
// is one contiguous 1 KB), scaled by the power of two found on the device from the kernel's
// largest magnitude (no host round trip; nothing is assumed about the size of a weight).
#include "common.h"

namespace {

constexpr int C16_CIN = 32, C16_KT = 11, C16_KF = 21;
constexpr int C16_CELL = 144;           // bytes per (frame, position) cell of the patch

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
union Frag16 {
    u32x4 u;
    f16x8 h;
};

// the two fp16 pieces of s (already scaled): first piece in the low half, second in the high half
__device__ __forceinline__ unsigned f16_pieces(float s) {
    const _Float16 h1 = (_Float16)s;
    const _Float16 h2 = (_Float16)(s - (float)h1);
    return (unsigned)__builtin_bit_cast(unsigned short, h1) |
           ((unsigned)__builtin_bit_cast(unsigned short, h2) << 16);
}

// scale that puts a magnitude with these bits into [2^14, 2^15) (1 for zero)
__device__ __forceinline__ float scale_below_f16_max(unsigned max_bits) {
    const int e = (int)((max_bits >> 23) & 0xFF) - 127;
    const int se = max_bits == 0u ? 0 : min(max(14 - e, -60), 60);
    return __uint_as_float((unsigned)(se + 127) << 23);
}

template <int COUT, int FI>
struct Geometry16 {
    static constexpr int FO = FI / 2;
    static constexpr int TT = FO == 20 ? 16 : 32;
    static constexpr int PF = FO + 10;
    static constexpr int PT = TT + C16_KT - 1;
    static constexpr int NT = COUT / 16;
    static constexpr size_t LDS = (size_t)PT * PF * C16_CELL;
    static_assert((TT / 4) * FO == 80, "a wave owns 5 M tiles");
};

__global__ void __launch_bounds__(256)
conv16_absmax_kernel(const float *__restrict__ w, int n, unsigned *__restrict__ max_bits) {
    float m = 0.f;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x)
        m = fmaxf(m, fabsf(w[i]));
    m = wave_max(m);
    if ((threadIdx.x & 63) == 0) atomicMax(max_bits, __float_as_uint(m));
}

// packed[((tap * NT + nt) * 2 + piece) * 64 + kg * 16 + n] = 8 halves: piece of
// w[co = 16 nt + n][ci = 8 kg .. 8 kg + 7][kt][kf] * s_w, tap = kt * 21 + kf
__global__ void __launch_bounds__(256)
conv16_pack_fwd_kernel(const float *__restrict__ w, const unsigned *__restrict__ max_bits,
                       u32x4 *__restrict__ packed, int cout) {
    const int nt_count = cout / 16;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;        // (tap, nt, lane)
    if (i >= C16_KT * C16_KF * nt_count * 64) return;
    const int lane = i & 63, nt = (i >> 6) % nt_count, tap = (i >> 6) / nt_count;
    const int n = lane & 15, kg = lane >> 4, kt = tap / C16_KF, kf = tap % C16_KF;
    const float s_w = scale_below_f16_max(*max_bits);
    unsigned q[8];
#pragma unroll
    for (int e = 0; e < 8; ++e)
        q[e] = f16_pieces(w[(((size_t)(16 * nt + n) * C16_CIN + 8 * kg + e) * C16_KT + kt) * C16_KF +
                            kf] * s_w);
    u32x4 *dst = packed + ((size_t)(tap * nt_count + nt) * 2) * 64 + lane;
    dst[0] = (u32x4){(q[0] & 0xFFFFu) | (q[1] << 16), (q[2] & 0xFFFFu) | (q[3] << 16),
                     (q[4] & 0xFFFFu) | (q[5] << 16), (q[6] & 0xFFFFu) | (q[7] << 16)};
    dst[64] = (u32x4){(q[0] >> 16) | (q[1] & 0xFFFF0000u), (q[2] >> 16) | (q[3] & 0xFFFF0000u),
                      (q[4] >> 16) | (q[5] & 0xFFFF0000u), (q[6] >> 16) | (q[7] & 0xFFFF0000u)};
}

// backward order: packed[(((tap * PASSES + pass) * 2 + nt) * 2 + piece) * 64 + kg * 16 + n] =
// 8 halves: piece of w[co = 32 pass + 8 kg .. + 7][ci = 16 nt + n][kt][kf] * s_w
__global__ void __launch_bounds__(256)
conv16_pack_bwd_kernel(const float *__restrict__ w, const unsigned *__restrict__ max_bits,
                       u32x4 *__restrict__ packed, int cout) {
    const int passes = cout / 32;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;        // (tap, pass, nt, lane)
    if (i >= C16_KT * C16_KF * passes * 2 * 64) return;
    const int lane = i & 63, nt = (i >> 6) & 1, pass = (i >> 7) % passes, tap = (i >> 7) / passes;
    const int n = lane & 15, kg = lane >> 4, kt = tap / C16_KF, kf = tap % C16_KF;
    const float s_w = scale_below_f16_max(*max_bits);
    unsigned q[8];
#pragma unroll
    for (int e = 0; e < 8; ++e)
        q[e] = f16_pieces(w[(((size_t)(32 * pass + 8 * kg + e) * C16_CIN + 16 * nt + n) * C16_KT + kt) *
                                C16_KF + kf] * s_w);
    u32x4 *dst = packed + ((size_t)((tap * passes + pass) * 2 + nt) * 2) * 64 + lane;
    dst[0] = (u32x4){(q[0] & 0xFFFFu) | (q[1] << 16), (q[2] & 0xFFFFu) | (q[3] << 16),
                     (q[4] & 0xFFFFu) | (q[5] << 16), (q[6] & 0xFFFFu) | (q[7] << 16)};
    dst[64] = (u32x4){(q[0] >> 16) | (q[1] & 0xFFFF0000u), (q[2] >> 16) | (q[3] & 0xFFFF0000u),
                      (q[4] >> 16) | (q[5] & 0xFFFF0000u), (q[6] >> 16) | (q[7] & 0xFFFF0000u)};
}

template <int CTRL>
__device__ __forceinline__ float dpp_f32(float v) {
    return __uint_as_float((unsigned)__builtin_amdgcn_update_dpp(0, (int)__float_as_uint(v), CTRL,
                                                                 0xF, 0xF, false));
}

__device__ __forceinline__ float4 mask_dz16(float4 g, const float *act, size_t index4, float upper) {
    if (act) {
        const float4 v = reinterpret_cast<const float4 *>(act)[index4];
        g.x = (v.x > 0.f && v.x < upper) ? g.x : 0.f;
        g.y = (v.y > 0.f && v.y < upper) ? g.y : 0.f;
        g.z = (v.z > 0.f && v.z < upper) ? g.z : 0.f;
        g.w = (v.w > 0.f && v.w < upper) ? g.w : 0.f;
    }
    return g;
}

// ---------------------------------------------------------------------------------------------
// data gradient.  dz has no bound (gradients span decades) - but a product only needs a common
// scale along its K axis, and all taps of one kt pair an output row with dz cells of ONE frame:
// every dz FRAME of the patch (all positions x the 32 channels of the pass) carries its own power
// of two.  Staging is two-phase: the slice is loaded into registers while an LDS atomicMax per
// frame finds the largest magnitude, then scaled into [2^13, 2^14), split into two fp16 pieces
// and written to the patch.  The 10 / 11 taps of a (kt, parity) accumulate in fp32 by MFMA from
// a zero accumulator; the result enters the total times the frame's inverse scale (4 FMAs per
// ~32 MFMAs): 22 significand bits relative to the largest gradient of a frame of the patch - the
// granularity of the data gradient GEMM's per-row scales.
// ---------------------------------------------------------------------------------------------
template <int COUT, int FI>
__global__ void __launch_bounds__(256)
conv16_bwd_data_kernel(const float *__restrict__ dz, const u32x4 *__restrict__ wp,
                       const unsigned *__restrict__ w_max_bits, float *__restrict__ dx, int T,
                       int dz_time_major, const float *__restrict__ act, float upper) {
    using G = Geometry16<COUT, FI>;
    constexpr int PASSES = COUT / 32;
    constexpr int CH = G::PT > 32 ? 2 : 1;                  // staging chunks (whole frames)
    constexpr int FPC = (G::PT + CH - 1) / CH;              // frames per chunk
    constexpr int NV = (FPC * G::PF * 8 + 255) / 256;       // float4 per thread and chunk
    extern __shared__ __attribute__((aligned(16))) char patch[];    // [PT][PF][C16_CELL], then
    unsigned *fmax = reinterpret_cast<unsigned *>(patch + G::LDS);  // per frame: max bits,
    float *finv = reinterpret_cast<float *>(fmax + G::PT);          // inverse scale
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int t0 = blockIdx.x * G::TT, b = blockIdx.y;
    const int kg = lane >> 4, n = lane & 15;
    const float out_scale = 1.0f / scale_below_f16_max(*w_max_bits);

    int base_a[5], frame_c[5][4];       // byte offset of the A-fragment rows / patch frame of the
#pragma unroll                          // accumulator rows (before the kt shift)
    for (int ti = 0; ti < 5; ++ti) {
        const int row = ti * 16 + n, tt = row / G::FO, j = row % G::FO;
        base_a[ti] = (((G::TT / 4) * wave + tt) * G::PF + j) * C16_CELL + 16 * kg;
#pragma unroll
        for (int r = 0; r < 4; ++r) frame_c[ti][r] = (G::TT / 4) * wave + (ti * 16 + 4 * kg + r) / G::FO;
    }
    f32x4 acc[2][5][2];
#pragma unroll
    for (int p = 0; p < 2; ++p)
#pragma unroll
        for (int ti = 0; ti < 5; ++ti)
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) acc[p][ti][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};

#pragma unroll 1
    for (int pass = 0; pass < PASSES; ++pass) {
        // ---- stage dz[b, t0-5 .. , :, 32 pass .. 32 pass + 31] with 5 zero positions each side ----
        __syncthreads();                        // (everyone is done with the previous patch)
        if (tid < G::PT) fmax[tid] = 0u;
        __syncthreads();
        // (in CH chunks of whole frames, so that a thread holds at most NV float4 at a time)
#pragma unroll 1
        for (int ch = 0; ch < CH; ++ch) {
            float4 v[NV];
#pragma unroll
            for (int k = 0; k < NV; ++k) {
                const int i = ch * FPC * G::PF * 8 + tid + k * 256;
                const int c4 = i & 7, pos = (i >> 3) % G::PF, pr = i / (8 * G::PF);
                const int ts = t0 - 5 + pr, fo = pos - 5;
                v[k] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (tid + k * 256 < FPC * G::PF * 8 && pr < G::PT && ts >= 0 && ts < T && fo >= 0 &&
                    fo < G::FO) {
                    const size_t cell = dz_time_major ? (size_t)ts * gridDim.y + b
                                                      : (size_t)b * T + ts;
                    const size_t at = (cell * G::FO + fo) * (COUT / 4) + pass * 8 + c4;
                    v[k] = mask_dz16(reinterpret_cast<const float4 *>(dz)[at], act, at, upper);
                    const float m = fmaxf(fmaxf(fabsf(v[k].x), fabsf(v[k].y)),
                                          fmaxf(fabsf(v[k].z), fabsf(v[k].w)));
                    if (m > 0.f) atomicMax(fmax + pr, __float_as_uint(m));
                }
            }
            __syncthreads();
#pragma unroll
            for (int k = 0; k < NV; ++k) {
                const int i = ch * FPC * G::PF * 8 + tid + k * 256;
                const int c4 = i & 7, pos = (i >> 3) % G::PF, pr = i / (8 * G::PF);
                if (tid + k * 256 >= FPC * G::PF * 8 || pr >= G::PT) continue;
                const unsigned mbits = fmax[pr];
                const int me = (int)((mbits >> 23) & 0xFF) - 127;
                const int mse = (mbits == 0u || me > 127) ? 0 : min(max(13 - me, -100), 100);
                const float sc = __uint_as_float((unsigned)(mse + 127) << 23);
                const unsigned q0 = f16_pieces(v[k].x * sc), q1 = f16_pieces(v[k].y * sc),
                               q2 = f16_pieces(v[k].z * sc), q3 = f16_pieces(v[k].w * sc);
                char *cellp = patch + (pr * G::PF + pos) * C16_CELL;
                *reinterpret_cast<u32x2 *>(cellp + 8 * c4) =
                    (u32x2){(q0 & 0xFFFFu) | (q1 << 16), (q2 & 0xFFFFu) | (q3 << 16)};
                *reinterpret_cast<u32x2 *>(cellp + 64 + 8 * c4) =
                    (u32x2){(q0 >> 16) | (q1 & 0xFFFF0000u), (q2 >> 16) | (q3 & 0xFFFF0000u)};
                if (c4 == 0 && pos == 0) finv[pr] = __uint_as_float((unsigned)(127 - mse) << 23);
            }
        }
        __syncthreads();

#pragma unroll 1
        for (int kt = 0; kt < C16_KT; ++kt) {
#pragma unroll
            for (int p = 0; p < 2; ++p) {
                // even output frequencies (f = 2j) meet the odd kf, odd ones (f = 2j+1) the even
                // kf; dz position = j + 4 + p - m (+5 for the zero border) with m = kf / 2
                const int taps = p == 0 ? 10 : 11;
                f32x4 part[5][2];
#pragma unroll
                for (int m = 0; m < taps; ++m) {
                    const int kf = p == 0 ? 2 * m + 1 : 2 * m;
                    const u32x4 *wt = wp + (size_t)(((kt * C16_KF + kf) * PASSES + pass) * 2) * 2 * 64 +
                                      lane;
                    const int tap_off = ((10 - kt) * G::PF + 9 + p - m) * C16_CELL;
                    Frag16 a1[5], a2[5];
#pragma unroll
                    for (int ti = 0; ti < 5; ++ti) {
                        a1[ti].u = *reinterpret_cast<const u32x4 *>(patch + base_a[ti] + tap_off);
                        a2[ti].u = *reinterpret_cast<const u32x4 *>(patch + base_a[ti] + tap_off + 64);
                    }
#pragma unroll
                    for (int nt = 0; nt < 2; ++nt) {
                        Frag16 w1, w2;
                        w1.u = wt[nt * 128];
                        w2.u = wt[nt * 128 + 64];
#pragma unroll
                        for (int ti = 0; ti < 5; ++ti)
                            part[ti][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(
                                a1[ti].h, w1.h, m == 0 ? zero : part[ti][nt], 0, 0, 0);
#pragma unroll
                        for (int ti = 0; ti < 5; ++ti)
                            part[ti][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(
                                a1[ti].h, w2.h, part[ti][nt], 0, 0, 0);
#pragma unroll
                        for (int ti = 0; ti < 5; ++ti)
                            part[ti][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(
                                a2[ti].h, w1.h, part[ti][nt], 0, 0, 0);
                    }
                }
                // the taps of this kt read dz frame (row frame + 10 - kt): its inverse scale
#pragma unroll
                for (int ti = 0; ti < 5; ++ti)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float inv = finv[frame_c[ti][r] + 10 - kt];
                        acc[p][ti][0][r] += part[ti][0][r] * inv;
                        acc[p][ti][1][r] += part[ti][1][r] * inv;
                    }
            }
        }
    }

    // ---- write dx[b, t, 2j + p, ci] ---------------------------------------------------------------
#pragma unroll
    for (int p = 0; p < 2; ++p)
#pragma unroll
        for (int ti = 0; ti < 5; ++ti)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = ti * 16 + 4 * kg + r, tt = row / G::FO, j = row % G::FO;
                const int t = t0 + (G::TT / 4) * wave + tt;
                if (t < T) {
                    float *out = dx + ((size_t)(b * T + t) * FI + 2 * j + p) * C16_CIN + n;
                    out[0] = acc[p][ti][0][r] * out_scale;
                    out[16] = acc[p][ti][1][r] * out_scale;
                }
            }
}

template <int COUT, int FI>
__global__ void __launch_bounds__(256)
conv16_fwd_kernel(const float *__restrict__ x, float x_scale, const u32x4 *__restrict__ wp,
                  const unsigned *__restrict__ w_max_bits, const float *__restrict__ bias,
                  float *__restrict__ y, int T, float cutoff, int y_time_major) {
    using G = Geometry16<COUT, FI>;
    constexpr int NT = G::NT;
    extern __shared__ __attribute__((aligned(16))) char patch[];    // [PT][PF][C16_CELL]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int t0 = blockIdx.x * G::TT, b = blockIdx.y;
    const int kg = lane >> 4, n = lane & 15;
    const float out_scale = 1.0f / (x_scale * scale_below_f16_max(*w_max_bits));

    int base_a[5];                      // byte offset of this lane's row cell + its channel group
#pragma unroll
    for (int ti = 0; ti < 5; ++ti) {
        const int row = ti * 16 + n, tt = row / G::FO, fo = row % G::FO;
        base_a[ti] = (((G::TT / 4) * wave + tt) * G::PF + fo) * C16_CELL + 16 * kg;
    }
    f32x4 acc[5][NT];
#pragma unroll
    for (int ti = 0; ti < 5; ++ti)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[ti][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // the 2 * NT B fragments (piece, N tile) of tap (kt, kf): contiguous 1 KB each
    auto load_b = [&](int kt, int kf, u32x4 (&dst)[2][NT]) {
        const u32x4 *wt = wp + (size_t)((kt * C16_KF + kf) * NT) * 2 * 64 + lane;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            dst[0][nt] = wt[(nt * 2) * 64];
            dst[1][nt] = wt[(nt * 2 + 1) * 64];
        }
    };

#pragma unroll
    for (int par = 0; par < 2; ++par) {    // the two frequency-parity planes, one after the other
        if (par) __syncthreads();
        for (int i = tid; i < G::PT * G::PF * 8; i += 256) {
            const int c4 = i & 7, pos = (i >> 3) % G::PF, pr = i / (8 * G::PF);
            const int ts = t0 - 5 + pr, fi = 2 * pos + par - 9;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (ts >= 0 && ts < T && fi >= 0 && fi < FI)
                v = reinterpret_cast<const float4 *>(x)[((size_t)(b * T + ts) * FI + fi) * 8 + c4];
            // (saturating: an input outside the bound its scale was chosen for must not turn
            // into inf - the caller only takes this kernel behind the clipped ReLU)
            const unsigned q0 = f16_pieces(fminf(fmaxf(v.x * x_scale, -65504.f), 65504.f)),
                           q1 = f16_pieces(fminf(fmaxf(v.y * x_scale, -65504.f), 65504.f)),
                           q2 = f16_pieces(fminf(fmaxf(v.z * x_scale, -65504.f), 65504.f)),
                           q3 = f16_pieces(fminf(fmaxf(v.w * x_scale, -65504.f), 65504.f));
            char *cell = patch + (pr * G::PF + pos) * C16_CELL + 8 * c4;
            *reinterpret_cast<u32x2 *>(cell) =
                (u32x2){(q0 & 0xFFFFu) | (q1 << 16), (q2 & 0xFFFFu) | (q3 << 16)};
            *reinterpret_cast<u32x2 *>(cell + 64) =
                (u32x2){(q0 >> 16) | (q1 & 0xFFFF0000u), (q2 >> 16) | (q3 & 0xFFFF0000u)};
        }
        __syncthreads();
        const int taps = par == 0 ? 11 : 10;
        // B fragments (weights, from L2) run one tap ahead of the MFMAs that use them
        u32x4 cur[2][NT], nxt[2][NT];
        load_b(0, par, cur);
        for (int kt = 0; kt < C16_KT; ++kt) {
#pragma unroll
            for (int m = 0; m < taps; ++m) {
                const bool wrap = m + 1 == taps;
                load_b(wrap ? min(kt + 1, C16_KT - 1) : kt, wrap ? par : 2 * (m + 1) + par, nxt);
                __builtin_amdgcn_sched_barrier(0);
                const int tap_off = (kt * G::PF + m) * C16_CELL;
                Frag16 a1[5], a2[5];
#pragma unroll
                for (int ti = 0; ti < 5; ++ti) {
                    a1[ti].u = *reinterpret_cast<const u32x4 *>(patch + base_a[ti] + tap_off);
                    a2[ti].u = *reinterpret_cast<const u32x4 *>(patch + base_a[ti] + tap_off + 64);
                }
                // x1 w1, x1 w2, x2 w1: a tile's accumulator comes back after all the others
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    Frag16 w1, w2;
                    w1.u = cur[0][nt];
                    w2.u = cur[1][nt];
#pragma unroll
                    for (int ti = 0; ti < 5; ++ti)
                        acc[ti][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1[ti].h, w1.h,
                                                                             acc[ti][nt], 0, 0, 0);
#pragma unroll
                    for (int ti = 0; ti < 5; ++ti)
                        acc[ti][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1[ti].h, w2.h,
                                                                             acc[ti][nt], 0, 0, 0);
#pragma unroll
                    for (int ti = 0; ti < 5; ++ti)
                        acc[ti][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a2[ti].h, w1.h,
                                                                             acc[ti][nt], 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int p = 0; p < 2; ++p)
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) cur[p][nt] = nxt[p][nt];
            }
        }
    }

    float bias_v[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) bias_v[nt] = bias ? bias[nt * 16 + n] : 0.f;
#pragma unroll
    for (int ti = 0; ti < 5; ++ti)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = ti * 16 + 4 * kg + r, tt = row / G::FO, fo = row % G::FO;
            const int t = t0 + (G::TT / 4) * wave + tt;
            if (t < T) {
                const size_t cell = y_time_major ? (size_t)t * gridDim.y + b : (size_t)b * T + t;
                float *out = y + (cell * G::FO + fo) * COUT + n;
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    float v = acc[ti][nt][r] * out_scale + bias_v[nt];
                    if (cutoff > 0.f) v = fminf(fmaxf(v, 0.f), cutoff);
                    out[nt * 16] = v;
                }
            }
        }
}

template <int COUT, int FI>
int launch_fwd16(const float *x, float x_scale, const void *packed, const unsigned *w_max_bits,
                 const float *bias, float *y, int B, int T, float cutoff, int y_time_major,
                 hipStream_t s) {
    using G = Geometry16<COUT, FI>;
    if (hipFuncSetAttribute(reinterpret_cast<const void *>(&conv16_fwd_kernel<COUT, FI>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)G::LDS) != hipSuccess)
        return CTCASR_ERR_LAUNCH;
    dim3 grid((T + G::TT - 1) / G::TT, B);
    conv16_fwd_kernel<COUT, FI><<<grid, 256, G::LDS, s>>>(
        x, x_scale, reinterpret_cast<const u32x4 *>(packed), w_max_bits, bias, y, T, cutoff,
        y_time_major);
    return ctcasr_launch_status();
}

template <int COUT, int FI>
int launch_bwd16(const float *dz, const void *packed, const unsigned *w_max_bits, float *dx, int B,
                 int T, int dz_time_major, const float *act, float upper, hipStream_t s) {
    using G = Geometry16<COUT, FI>;
    const size_t lds = G::LDS + (size_t)G::PT * 8;          // + per-frame maxima / inverse scales
    if (hipFuncSetAttribute(reinterpret_cast<const void *>(&conv16_bwd_data_kernel<COUT, FI>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
        return CTCASR_ERR_LAUNCH;
    dim3 grid((T + G::TT - 1) / G::TT, B);
    conv16_bwd_data_kernel<COUT, FI><<<grid, 256, lds, s>>>(
        dz, reinterpret_cast<const u32x4 *>(packed), w_max_bits, dx, T, dz_time_major, act, upper);
    return ctcasr_launch_status();
}

// ---------------------------------------------------------------------------------------------
// Kernel gradient on the fp16 pipe:
//   dw[co, ci, kt, kf] = sum_{b,t,fo} dz[b, t, fo, co] * x[b, t + kt - 5, 2 fo + kf - 9, ci]
// The summation axis (b, t, fo) is the MFMA's K axis, and a lane's 8 consecutive k must sit in 16
// contiguous, aligned bytes for BOTH operands whatever the tap: kt shifts t, kf shifts 2 fo - the
// utterance index b is the one axis no tap shifts.  So a pack pass (HBM-bound, ~50 us at C3)
// first writes both operands as fp16 pieces with 8 utterances innermost:
//   x16  [b / 8][t][f ][piece][ci  ][b % 8]      (x * x_scale; x is bounded, see the forward pass)
//   dz16 [b / 8][t][fo][piece][cout][b % 8]      (masked dz * s[co])
// with a power-of-two scale per output channel (the non-contracted index of dz) from the channel's
// largest |dz| (found by wrw16_colmax_kernel, which also leaves the bias gradient) - the
// convention of the weight-gradient GEMMs (split_gemm.wgrad16): 22 significand bits relative to
// the channel's largest gradient.  The conversion happens ONCE per element, not once per kt
// workgroup (11 x), and the main kernel's staging is a flat 16-byte copy: a tile = one frame t x
// 8 utterances = FO K slots of 8; its x slice [FI][2][32][8] halves lands at position 9 of an LDS
// line [FI + 20 positions] whose borders stay zero; a K step = 4 slots (fo = 4 q + g for k group
// g): A fragment dzs[fo][piece][co], B fragment xs[2 fo + kf][piece][ci] - one ds_read_b128 each,
// contiguous per 16 lanes, conflict-free.  Grid (splits, 11 kt, cout / 32) as in conv.hip; a wave
// owns one ci tile x both co tiles x half of the 21 kf taps: per K step 4 A reads, 2 B reads per
// tap and 6 MFMAs per tap (88 accumulator registers).  80 KB of LDS at FI = 40: two workgroups
// per CU, one staging while the other multiplies.
// ---------------------------------------------------------------------------------------------
template <int FI>
struct Wrw16Geometry {
    static constexpr int FO = FI / 2;
    static constexpr int KS = (FO + 3) / 4;              // K steps per tile
    static constexpr int SLOTS = 4 * KS;                 // dz slots (those >= FO stay zero)
    static constexpr int PF = 2 * (SLOTS - 1) + C16_KF;  // positions 2 fo + kf
    static constexpr int PFA = (PF + 3) & ~3;
    static constexpr int X_CELLS = FI * 64;              // 16-byte cells of a tile's x slice
    static constexpr int DZ_CELLS = FO * 64;             // ... of its dz slice (32 channels)
    static constexpr int X_PER = (X_CELLS + 255) / 256, DZ_PER = (DZ_CELLS + 255) / 256;
    static constexpr size_t LDS = ((size_t)PFA + SLOTS) * 1024;
};

// Column maxima and column sums of the masked dz [rows, cout]: max_bits[cout] (zeroed by the
// caller) raised to the bit patterns of max |dz|, dbias[cout] += sums (optional).
// gridDim.x * 256 must be a multiple of cout / 4.
__global__ void __launch_bounds__(256)
wrw16_colmax_kernel(const float *__restrict__ dz, const float *__restrict__ act, float upper,
                    long cells4, int cout, unsigned *__restrict__ max_bits,
                    float *__restrict__ dbias) {
    __shared__ unsigned lmax[96];
    __shared__ float lsum[96];
    const int tid = threadIdx.x;
    if (tid < 96) {
        lmax[tid] = 0u;
        lsum[tid] = 0.f;
    }
    __syncthreads();
    const long stride = (long)gridDim.x * 256;
    const long first = (long)blockIdx.x * 256 + tid;
    const int c4 = (int)(first % (cout / 4));
    float4 m = make_float4(0.f, 0.f, 0.f, 0.f), sum = make_float4(0.f, 0.f, 0.f, 0.f);
    for (long i = first; i < cells4; i += stride) {
        const float4 v = mask_dz16(reinterpret_cast<const float4 *>(dz)[i], act, (size_t)i, upper);
        m.x = fmaxf(m.x, fabsf(v.x)); m.y = fmaxf(m.y, fabsf(v.y));
        m.z = fmaxf(m.z, fabsf(v.z)); m.w = fmaxf(m.w, fabsf(v.w));
        sum.x += v.x; sum.y += v.y; sum.z += v.z; sum.w += v.w;
    }
    atomicMax(&lmax[4 * c4 + 0], __float_as_uint(m.x));
    atomicMax(&lmax[4 * c4 + 1], __float_as_uint(m.y));
    atomicMax(&lmax[4 * c4 + 2], __float_as_uint(m.z));
    atomicMax(&lmax[4 * c4 + 3], __float_as_uint(m.w));
    if (dbias) {
        atomicAdd(&lsum[4 * c4 + 0], sum.x);
        atomicAdd(&lsum[4 * c4 + 1], sum.y);
        atomicAdd(&lsum[4 * c4 + 2], sum.z);
        atomicAdd(&lsum[4 * c4 + 3], sum.w);
    }
    __syncthreads();
    if (tid < cout) {
        atomicMax(max_bits + tid, lmax[tid]);
        if (dbias) atomicAdd(dbias + tid, lsum[tid]);
    }
}

__device__ __forceinline__ void store_pieces8(u32x4 *dst, int piece_stride, const unsigned (&q)[8]) {
    dst[0] = (u32x4){(q[0] & 0xFFFFu) | (q[1] << 16), (q[2] & 0xFFFFu) | (q[3] << 16),
                     (q[4] & 0xFFFFu) | (q[5] << 16), (q[6] & 0xFFFFu) | (q[7] << 16)};
    dst[piece_stride] =
        (u32x4){(q[0] >> 16) | (q[1] & 0xFFFF0000u), (q[2] >> 16) | (q[3] & 0xFFFF0000u),
                (q[4] >> 16) | (q[5] & 0xFFFF0000u), (q[6] >> 16) | (q[7] & 0xFFFF0000u)};
}

// x f32[B, T, FI, 32] -> x16 [(b / 8) T FI][piece][ci] cells of 8 halves (utterances b % 8)
__global__ void __launch_bounds__(256)
wrw16_pack_x_kernel(const float *__restrict__ x, float x_scale, u32x4 *__restrict__ x16, int B,
                    long per_utt /* T FI 32 */, long cells /* ceil(B / 8) per_utt */) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= cells) return;
    const long within = i % per_utt;
    const int b0 = (int)(i / per_utt) * 8;
    unsigned q[8];
#pragma unroll
    for (int e = 0; e < 8; ++e)
        q[e] = b0 + e < B ? f16_pieces(x[(long)(b0 + e) * per_utt + within] * x_scale) : 0u;
    // cell index: ((bblk T FI + t FI + f) 2 + piece) 32 + ci
    const long row = i / C16_CIN;
    store_pieces8(x16 + row * 64 + (i % C16_CIN), 32, q);
}

// dz f32[B, T, FO, cout] (or [T, B, FO, cout]) -> dz16 [(b / 8) T FO][piece][cout] cells, masked
// and scaled per channel
__global__ void __launch_bounds__(256)
wrw16_pack_dz_kernel(const float *__restrict__ dz, const float *__restrict__ act, float upper,
                     const unsigned *__restrict__ max_bits, u32x4 *__restrict__ dz16, int B, int T,
                     int fo_cout /* FO cout */, int cout, int time_major, long cells) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= cells) return;
    const long per_utt = (long)T * fo_cout;
    const long within = i % per_utt;
    const int b0 = (int)(i / per_utt) * 8;
    const int co = (int)(within % cout);
    const int t = (int)(within / fo_cout);
    const long in_frame = within % fo_cout;
    const float s = scale_below_f16_max(max_bits[co]);
    unsigned q[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        q[e] = 0u;
        if (b0 + e < B) {
            const long at = time_major ? ((long)t * B + b0 + e) * fo_cout + in_frame
                                       : (long)(b0 + e) * per_utt + within;
            float v = dz[at];
            if (act) {
                const float a = act[at];
                v = (a > 0.f && a < upper) ? v : 0.f;
            }
            q[e] = f16_pieces(v * s);
        }
    }
    const long row = i / cout;                       // (bblk, t, fo)
    store_pieces8(dz16 + row * 2 * cout + co, cout, q);
}

template <int FI>
__global__ void __launch_bounds__(256, 2)
wrw16_kernel(const u32x4 *__restrict__ x16, const u32x4 *__restrict__ dz16,
             float *__restrict__ partial, int nblk, int T, int cout) {
    using G = Wrw16Geometry<FI>;
    extern __shared__ __attribute__((aligned(16))) unsigned char wsm16[];
    u32x4 *xs = reinterpret_cast<u32x4 *>(wsm16);                    // [PFA][2][32] cells
    u32x4 *dzs = xs + G::PFA * 64;                                   // [SLOTS][2][32] cells
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int split = blockIdx.x, nsplit = gridDim.x, kt = blockIdx.y, cg = blockIdx.z;
    const int n = lane & 15, g = lane >> 4;
    const int ci_tile = wave & 1, half = wave >> 1;
    const int kf0 = half ? 11 : 0;
    constexpr int TAPS = 11;                    // half 1 owns 10: its last one is skipped
    const int tiles = nblk * T;

    // borders of the position line and the unused dz slots: zero for the whole kernel
    for (int i = tid; i < (G::PFA + G::SLOTS) * 64; i += 256) xs[i] = (u32x4){0u, 0u, 0u, 0u};

    f32x4 acc[TAPS][2];
#pragma unroll
    for (int k = 0; k < TAPS; ++k) {
        acc[k][0] = (f32x4){0.f, 0.f, 0.f, 0.f};
        acc[k][1] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }

    // tile = t * nblk + bblk; it contributes if frame t + kt - 5 of x exists
    auto next_valid = [&](int tile) {
        while (tile < tiles) {
            const int ts = tile / nblk + kt - 5;
            if (ts >= 0 && ts < T) break;
            tile += nsplit;
        }
        return tile;
    };
    u32x4 rx[G::X_PER], rdz[G::DZ_PER];
    auto fetch = [&](int tile) {
        const int t = tile / nblk, bblk = tile % nblk;
        const u32x4 *xsrc = x16 + ((size_t)bblk * T + (t + kt - 5)) * G::X_CELLS;
#pragma unroll
        for (int j = 0; j < G::X_PER; ++j) {
            const int i = tid + j * 256;
            if (G::X_CELLS % 256 == 0 || i < G::X_CELLS) rx[j] = xsrc[i];
        }
        // dz cells of a (slot, piece): cout of them, this workgroup's 32 at 32 cg
        const u32x4 *dsrc = dz16 + ((size_t)bblk * T + t) * G::FO * 2 * cout + cg * 32;
#pragma unroll
        for (int j = 0; j < G::DZ_PER; ++j) {
            const int i = tid + j * 256;
            if (G::DZ_CELLS % 256 == 0 || i < G::DZ_CELLS) rdz[j] = dsrc[(i >> 5) * cout + (i & 31)];
        }
    };
    auto stage = [&]() {
#pragma unroll
        for (int j = 0; j < G::X_PER; ++j) {
            const int i = tid + j * 256;
            if (G::X_CELLS % 256 == 0 || i < G::X_CELLS) xs[9 * 64 + i] = rx[j];
        }
#pragma unroll
        for (int j = 0; j < G::DZ_PER; ++j) {
            const int i = tid + j * 256;
            if (G::DZ_CELLS % 256 == 0 || i < G::DZ_CELLS) dzs[i] = rdz[j];
        }
    };

    int tile = next_valid(split);
    if (tile < tiles) fetch(tile);
    while (tile < tiles) {
        __syncthreads();                 // the previous tile's fragment reads (first: the zeros)
        stage();
        __syncthreads();
        const int next = next_valid(tile + nsplit);
        if (next < tiles) fetch(next);
#pragma unroll 1
        for (int q = 0; q < G::KS; ++q) {
            const int slot = 4 * q + g;
            Frag16 a1[2], a2[2];
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                a1[c].u = dzs[(slot * 2 + 0) * 32 + c * 16 + n];
                a2[c].u = dzs[(slot * 2 + 1) * 32 + c * 16 + n];
            }
            const u32x4 *xb = xs + ((2 * slot + kf0) * 2) * 32 + ci_tile * 16 + n;
#pragma unroll
            for (int k = 0; k < TAPS; ++k) {
                if (k < TAPS - 1 || !half) {               // wave-uniform
                    Frag16 b1, b2;
                    b1.u = xb[k * 64];
                    b2.u = xb[k * 64 + 32];
#pragma unroll
                    for (int c = 0; c < 2; ++c) {
                        acc[k][c] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1[c].h, b1.h, acc[k][c], 0, 0, 0);
                        acc[k][c] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1[c].h, b2.h, acc[k][c], 0, 0, 0);
                        acc[k][c] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a2[c].h, b1.h, acc[k][c], 0, 0, 0);
                    }
                }
            }
        }
        tile = next;
    }
    // D[row = 4 g + r][col = n]: co = 16 c + 4 g + r, ci = 16 ci_tile + n
    float *out = partial + (((size_t)split * gridDim.z + cg) * C16_KT + kt) * C16_KF * 1024;
#pragma unroll
    for (int k = 0; k < TAPS; ++k) {
        if (k < TAPS - 1 || !half) {
#pragma unroll
            for (int c = 0; c < 2; ++c)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    out[(size_t)(kf0 + k) * 1024 + (c * 16 + 4 * g + r) * C16_CIN + ci_tile * 16 +
                        n] = acc[k][c][r];
        }
    }
}

// dw[co][ci][kt][kf] = sum over splits of partial[split][co / 32][kt][kf][co % 32][ci], scales out.
// One thread per element in the PARTIALS' order (coalesced reads of every split; the 0.9 MB of
// results are scattered instead - in dw's order the reads were 4 KB apart: 146 us at C3).
__global__ void wrw16_reduce_kernel(const float *__restrict__ partial,
                                    const unsigned *__restrict__ max_bits, float inv_x_scale,
                                    float *__restrict__ dw, int nsplit, int cout) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    const int total = cout * C16_CIN * C16_KT * C16_KF;
    if (j >= total) return;
    const int ci = j & 31, co32 = (j >> 5) & 31, tap = (j >> 10) % (C16_KT * C16_KF);
    const int co = (j >> 10) / (C16_KT * C16_KF) * 32 + co32;
    float sum = 0.f;
#pragma unroll 4
    for (int sp = 0; sp < nsplit; ++sp) sum += partial[(size_t)sp * total + j];
    dw[((size_t)co * C16_CIN + ci) * (C16_KT * C16_KF) + tap] =
        sum * (inv_x_scale / scale_below_f16_max(max_bits[co]));
}

int wrw16_splits(int B, int T, int cout) {
    // two workgroups per CU (80 KB of LDS each at 40 input frequencies), never more than tiles
    const int tiles = ((B + 7) / 8) * T;
    const int want = 512 / (C16_KT * (cout / 32));
    return tiles < want ? tiles : want;
}

struct Wrw16Workspace {
    size_t max_bits, x16, dz16, partial, total;
};
Wrw16Workspace wrw16_workspace(int B, int T, int freq_in, int cout) {
    const size_t nblk = (size_t)(B + 7) / 8;
    Wrw16Workspace w;
    w.max_bits = 0;
    w.x16 = 512;
    w.dz16 = w.x16 + nblk * T * freq_in * 1024;
    w.partial = w.dz16 + nblk * T * (freq_in / 2) * 2 * cout * 16;
    w.total = w.partial + (size_t)wrw16_splits(B, T, cout) * cout * C16_CIN * C16_KT * C16_KF * 4;
    return w;
}

bool covered16(int freq_in, int cout) {
    return (freq_in == 40 && cout == 32) || (freq_in == 20 && cout == 96);
}

}  // namespace

// bytes of the fragment-ordered fp16 pieces of a layer's kernel (+ 16 for the magnitude word)
static size_t pack16_order_bytes(int cout) {     // one order: 2 pieces x 2 bytes per weight
    return (size_t)C16_KT * C16_KF * C16_CIN * cout * 4;
}
extern "C" size_t ctcasr_conv_s12_pack16_bytes(int cout) {
    if (cout != 32 && cout != 96) return 0;
    return 16 + 2 * pack16_order_bytes(cout);   // magnitude word, forward order, backward order
}

// w [cout, 32, 11, 21] -> `packed`: one word with the bit pattern of max |w| (16 bytes reserved),
// then the two fp16 pieces of w * s_w in forward fragment order, s_w = the power of two that puts
// max |w| into [2^14, 2^15).  Everything on the device; the weights change every step.
extern "C" int ctcasr_conv_s12_pack_weights16(const float *w, void *packed, int cout,
                                              ctcasr_stream_t stream) {
    if (!w || !packed || (cout != 32 && cout != 96)) return CTCASR_ERR_BAD_ARGUMENT;
    hipStream_t s = (hipStream_t)stream;
    unsigned *max_bits = reinterpret_cast<unsigned *>(packed);
    if (hipMemsetAsync(max_bits, 0, 16, s) != hipSuccess) return CTCASR_ERR_LAUNCH;
    const int n = cout * C16_CIN * C16_KT * C16_KF;
    conv16_absmax_kernel<<<64, 256, 0, s>>>(w, n, max_bits);
    const int threads = C16_KT * C16_KF * (cout / 16) * 64;
    conv16_pack_fwd_kernel<<<(threads + 255) / 256, 256, 0, s>>>(
        w, max_bits, reinterpret_cast<u32x4 *>(reinterpret_cast<char *>(packed) + 16), cout);
    const int threads_b = C16_KT * C16_KF * (cout / 32) * 2 * 64;
    conv16_pack_bwd_kernel<<<(threads_b + 255) / 256, 256, 0, s>>>(
        w, max_bits,
        reinterpret_cast<u32x4 *>(reinterpret_cast<char *>(packed) + 16 + pack16_order_bytes(cout)),
        cout);
    return ctcasr_launch_status();
}

// dx = data gradient like ctcasr_conv_s12_bwd_data, the products as fp16 x 3 with a power-of-two
// scale per dz frame of a workgroup's patch found while it is staged: no bound on dz is assumed.
extern "C" int ctcasr_conv_s12_bwd_data16(const float *dz, const void *packed, float *dx, int B,
                                          int T, int freq_in, int cout, int dz_time_major,
                                          const float *act, float relu_cutoff,
                                          ctcasr_stream_t stream) {
    if (!dz || !packed || !dx || B <= 0 || T <= 0 || (act && relu_cutoff <= 0.f))
        return CTCASR_ERR_BAD_ARGUMENT;
    if (!covered16(freq_in, cout) || B > 65535) return CTCASR_ERR_UNSUPPORTED;
    hipStream_t s = (hipStream_t)stream;
    const unsigned *max_bits = reinterpret_cast<const unsigned *>(packed);
    const void *pieces = reinterpret_cast<const char *>(packed) + 16 + pack16_order_bytes(cout);
    if (cout == 32)
        return launch_bwd16<32, 40>(dz, pieces, max_bits, dx, B, T, dz_time_major, act,
                                    relu_cutoff, s);
    return launch_bwd16<96, 20>(dz, pieces, max_bits, dx, B, T, dz_time_major, act, relu_cutoff, s);
}

// y = conv(x) + bias like ctcasr_conv_s12_fwd, the products as fp16 x 3 on the 16-bit matrix pipe.
// x must lie in [-bound, bound] with bound * x_scale < 65504 (x_scale a power of two): the caller
// takes this entry point behind a clipped ReLU only (larger inputs saturate - wrong, not inf).
extern "C" int ctcasr_conv_s12_fwd16(const float *x, float x_scale, const void *packed,
                                     const float *bias, float *y, int B, int T, int freq_in,
                                     int cout, float relu_cutoff, int y_time_major,
                                     ctcasr_stream_t stream) {
    if (!x || !packed || !y || B <= 0 || T <= 0 || !(x_scale > 0.f)) return CTCASR_ERR_BAD_ARGUMENT;
    if (!covered16(freq_in, cout) || B > 65535) return CTCASR_ERR_UNSUPPORTED;
    hipStream_t s = (hipStream_t)stream;
    const unsigned *max_bits = reinterpret_cast<const unsigned *>(packed);
    const void *pieces = reinterpret_cast<const char *>(packed) + 16;
    if (cout == 32)
        return launch_fwd16<32, 40>(x, x_scale, pieces, max_bits, bias, y, B, T, relu_cutoff,
                                    y_time_major, s);
    return launch_fwd16<96, 20>(x, x_scale, pieces, max_bits, bias, y, B, T, relu_cutoff,
                                y_time_major, s);
}

extern "C" size_t ctcasr_conv_s12_wrw16_workspace_bytes(int B, int T, int freq_in, int cout) {
    if (B <= 0 || T <= 0 || !covered16(freq_in, cout)) return 0;
    return wrw16_workspace(B, T, freq_in, cout).total;
}

// dw = kernel gradient like ctcasr_conv_s12_wrw, the products as fp16 x 3: x (bounded:
// |x| x_scale < 65504, the layer's input behind the clipped ReLU) and the masked dz, scaled per
// output channel by the power of two that its largest magnitude asks for, are first written as
// fp16 pieces with 8 utterances innermost (workspace), then one launch multiplies.  dbias as in
// ctcasr_conv_s12_wrw.  Deterministic up to dbias (atomics).
extern "C" int ctcasr_conv_s12_wrw16(const float *dz, const float *x, float x_scale, float *dw,
                                     int B, int T, int freq_in, int cout, int dz_time_major,
                                     const float *act, float relu_cutoff, float *dbias,
                                     void *workspace, size_t workspace_bytes,
                                     ctcasr_stream_t stream) {
    if (!dz || !x || !dw || B <= 0 || T <= 0 || (act && relu_cutoff <= 0.f) || !(x_scale > 0.f))
        return CTCASR_ERR_BAD_ARGUMENT;
    if (!covered16(freq_in, cout)) return CTCASR_ERR_UNSUPPORTED;
    const Wrw16Workspace w = wrw16_workspace(B, T, freq_in, cout);
    if (!workspace || workspace_bytes < w.total) return CTCASR_ERR_WORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    char *base = reinterpret_cast<char *>(workspace);
    unsigned *max_bits = reinterpret_cast<unsigned *>(base + w.max_bits);
    u32x4 *x16 = reinterpret_cast<u32x4 *>(base + w.x16);
    u32x4 *dz16 = reinterpret_cast<u32x4 *>(base + w.dz16);
    float *partial = reinterpret_cast<float *>(base + w.partial);
    const int fo = freq_in / 2, nblk = (B + 7) / 8;
    if (hipMemsetAsync(max_bits, 0, 512, s) != hipSuccess) return CTCASR_ERR_LAUNCH;
    const long cells4 = (long)B * T * fo * cout / 4;
    wrw16_colmax_kernel<<<240, 256, 0, s>>>(dz, act, relu_cutoff, cells4, cout, max_bits, dbias);
    const long per_utt = (long)T * freq_in * C16_CIN, xcells = (long)nblk * per_utt;
    wrw16_pack_x_kernel<<<(unsigned)((xcells + 255) / 256), 256, 0, s>>>(x, x_scale, x16, B,
                                                                        per_utt, xcells);
    const long dcells = (long)nblk * T * fo * cout;
    wrw16_pack_dz_kernel<<<(unsigned)((dcells + 255) / 256), 256, 0, s>>>(
        dz, act, relu_cutoff, max_bits, dz16, B, T, fo * cout, cout, dz_time_major, dcells);
    const int nsplit = wrw16_splits(B, T, cout);
    dim3 grid(nsplit, C16_KT, cout / 32);
    if (freq_in == 40) {
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(&wrw16_kernel<40>),
                                hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)Wrw16Geometry<40>::LDS) != hipSuccess)
            return CTCASR_ERR_LAUNCH;
        wrw16_kernel<40><<<grid, 256, Wrw16Geometry<40>::LDS, s>>>(x16, dz16, partial, nblk, T, cout);
    } else {
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(&wrw16_kernel<20>),
                                hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)Wrw16Geometry<20>::LDS) != hipSuccess)
            return CTCASR_ERR_LAUNCH;
        wrw16_kernel<20><<<grid, 256, Wrw16Geometry<20>::LDS, s>>>(x16, dz16, partial, nblk, T, cout);
    }
    const int total = cout * C16_CIN * C16_KT * C16_KF;
    wrw16_reduce_kernel<<<(total + 255) / 256, 256, 0, s>>>(partial, max_bits, 1.0f / x_scale, dw,
                                                            nsplit, cout);
    return ctcasr_launch_status();
}

// =============================================================================================
// First convolution of the stack (1 -> 32 channels, 11 x 41 taps, stride (2, 2), SAME padding;
// conv.hip's conv0 kernels) on the fp16 pipe.  conv.hip feeds one float per lane and fp32 MFMA
// (K = 451 taps only): 24 - 26 TFLOP/s.  Here:
//
// forward   y[b, t, fo, co] = bias[co] + sum_{kt,kf} x[b, 2t + kt - pt0, 2fo + kf - 19] w[co, kt, kf]
//   M = (t, fo) rows, N = co, K = taps in groups of 8 consecutive kf of one kt (6 groups per kt,
//   the last one holds kf = 40 only; 66 groups = 16.5 K steps of 4 groups, 83 % of the K axis is
//   real).  A lane's A fragment is 8 consecutive positions p = 2fo + kf .. + 7 of one input line -
//   2 bytes apart from its neighbour row's, so never 16-byte aligned as it stands: the patch sits
//   in LDS twice, as it is and shifted by two positions, and an M tile takes 16 rows of EQUAL fo
//   parity (fo = 2i + rho, copy rho): fragment offset 4i + 8 kb halves = 8-byte aligned, two
//   ds_read_b64.  The features have no bound: the workgroup scales its own patch by the power of
//   two its largest magnitude asks for (found while it is staged) - a row's sum only ever reads
//   that patch.  Weights: fp16 pieces in fragment order, packed per step, scale found on the
//   device, whole set (68 KB) in LDS.
//
// kernel gradient   dw[co, kt, kf] = sum_{b,t,fo} dz[b, t, fo, co] x[b, 2t + kt - pt0, 2fo + kf - 19]
//   the scheme of wrw16 above: both operands as fp16 pieces with 8 utterances innermost, dz
//   scaled per output channel, x by the tensor's largest magnitude; a tile = one output frame x 8
//   utterances (40 K slots), its 11 input lines [2 pieces][128 positions] and the dz slice in
//   LDS; a wave owns the taps kt = wave, wave + 4, wave + 8 x 3 kf tiles x 2 co tiles.
// =============================================================================================
namespace {

constexpr int Z0_KT = 11, Z0_KF = 41, Z0_FI = 80, Z0_FO = 40, Z0_TT = 16;
constexpr int Z0_PW = 128;                          // positions of a line: fi = position - 19
constexpr int Z0_PT = 2 * Z0_TT + Z0_KT - 2;        // 41 input lines per 16 output frames
constexpr int Z0_GROUPS = Z0_KT * 6;                // K groups (kt, 8 kf)
constexpr int Z0_QS = (Z0_GROUPS + 3) / 4;          // 17 K steps
constexpr int Z0_W_CELLS = Z0_QS * 2 * 2 * 64;      // packed weights: [q][nt][piece][lane]
constexpr int Z0_DW = 32 * Z0_KT * Z0_KF;
constexpr size_t Z0_FWD_LDS = (size_t)2 * 2 * Z0_PT * Z0_PW * 2 + (size_t)Z0_W_CELLS * 16 + 16;
constexpr size_t Z0_WRW_LDS = ((size_t)Z0_KT * 2 * Z0_PW + (size_t)Z0_FO * 2 * 32) * 16;
constexpr int Z0_WRW_GRID = 256;

// packed[((q * 2 + nt) * 2 + piece) * 64 + g * 16 + n] = 8 halves: piece of
// w[co = 16 nt + n][kt][kf = 8 kb .. + 7] * s_w for K group 4 q + g = 6 kt + kb (zero beyond)
__global__ void __launch_bounds__(256)
conv0_pack_w16_kernel(const float *__restrict__ w, const unsigned *__restrict__ max_bits,
                      u32x4 *__restrict__ packed) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;          // (q, nt, lane)
    if (i >= Z0_QS * 2 * 64) return;
    const int lane = i & 63, nt = (i >> 6) & 1, q = i >> 7;
    const int n = lane & 15, g = lane >> 4, gi = 4 * q + g, kt = gi / 6, kb = gi % 6;
    const float s_w = scale_below_f16_max(*max_bits);
    unsigned p[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int kf = 8 * kb + e;
        p[e] = (gi < Z0_GROUPS && kf < Z0_KF)
                   ? f16_pieces(w[((16 * nt + n) * Z0_KT + kt) * Z0_KF + kf] * s_w) : 0u;
    }
    store_pieces8(packed + ((size_t)(q * 2 + nt) * 2) * 64 + lane, 64, p);
}

__global__ void __launch_bounds__(256)
conv0_fwd16_kernel(const float *__restrict__ x, const u32x4 *__restrict__ packed,
                   const unsigned *__restrict__ w_max_bits, const float *__restrict__ bias,
                   float *__restrict__ y, int T, int t_out, int pt0, float cutoff) {
    extern __shared__ __attribute__((aligned(16))) unsigned char zsm[];
    _Float16 *copies = reinterpret_cast<_Float16 *>(zsm);      // [rho][piece][Z0_PT][Z0_PW]
    u32x4 *wl = reinterpret_cast<u32x4 *>(zsm + (size_t)2 * 2 * Z0_PT * Z0_PW * 2);
    unsigned *pmax = reinterpret_cast<unsigned *>(wl + Z0_W_CELLS);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int t0 = blockIdx.x * Z0_TT, b = blockIdx.y;
    const int g = lane >> 4, n = lane & 15;
    constexpr int PLANE = Z0_PT * Z0_PW;                        // halves per (copy, piece)

    if (tid == 0) *pmax = 0u;
    __syncthreads();
    constexpr int PER = (PLANE + 255) / 256;
    float v[PER];
    float m = 0.f;
#pragma unroll
    for (int j = 0; j < PER; ++j) {
        const int i = tid + j * 256, col = i % Z0_PW, pr = i / Z0_PW;
        const int ts = 2 * t0 - pt0 + pr, fi = col - 19;
        v[j] = (i < PLANE && ts >= 0 && ts < T && fi >= 0 && fi < Z0_FI)
                   ? x[((size_t)b * T + ts) * Z0_FI + fi] : 0.f;
        m = fmaxf(m, fabsf(v[j]));
    }
    m = wave_max(m);
    if (lane == 0) atomicMax(pmax, __float_as_uint(m));
    for (int i = tid; i < Z0_W_CELLS; i += 256) wl[i] = packed[i];
    __syncthreads();
    const float s_x = scale_below_f16_max(*pmax);
    // the copy for odd fo holds the line two positions to the left: its tail stays zero
    for (int i = tid; i < 2 * Z0_PT; i += 256) {
        _Float16 *tail = copies + (size_t)(2 + i / Z0_PT) * PLANE + (i % Z0_PT) * Z0_PW + Z0_PW - 2;
        tail[0] = (_Float16)0.f;
        tail[1] = (_Float16)0.f;
    }
#pragma unroll
    for (int j = 0; j < PER; ++j) {
        const int i = tid + j * 256, col = i % Z0_PW;
        if (i < PLANE) {
            const unsigned p = f16_pieces(v[j] * s_x);
            const _Float16 h1 = __builtin_bit_cast(_Float16, (unsigned short)(p & 0xFFFFu));
            const _Float16 h2 = __builtin_bit_cast(_Float16, (unsigned short)(p >> 16));
            copies[i] = h1;
            copies[PLANE + i] = h2;
            if (col >= 2) {
                copies[2 * PLANE + i - 2] = h1;
                copies[3 * PLANE + i - 2] = h2;
            }
        }
    }
    __syncthreads();

    // a wave owns 4 output frames x 40 frequencies: per parity 80 rows (tt, i), fo = 2 i + rho,
    // = 5 M tiles; tile 5 rho + k, lane row 16 k + n
    int base_a[10];
#pragma unroll
    for (int ti = 0; ti < 10; ++ti) {
        const int rho = ti / 5, mrow = (ti % 5) * 16 + n, tt = mrow / 20, i = mrow % 20;
        base_a[ti] = rho * 2 * PLANE + 2 * (4 * wave + tt) * Z0_PW + 4 * i;
    }
    f32x4 acc[10][2];
#pragma unroll
    for (int ti = 0; ti < 10; ++ti) {
        acc[ti][0] = (f32x4){0.f, 0.f, 0.f, 0.f};
        acc[ti][1] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll 1
    for (int q = 0; q < Z0_QS; ++q) {
        const int gi = 4 * q + g, kt = gi / 6, kb = gi - 6 * kt;
        const int off = gi < Z0_GROUPS ? kt * Z0_PW + 8 * kb : 0;     // (weights are zero beyond)
        Frag16 b1[2], b2[2];
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
            b1[nt].u = wl[((q * 2 + nt) * 2 + 0) * 64 + lane];
            b2[nt].u = wl[((q * 2 + nt) * 2 + 1) * 64 + lane];
        }
#pragma unroll
        for (int ti = 0; ti < 10; ++ti) {
            const u32x2 *p1 = reinterpret_cast<const u32x2 *>(copies + base_a[ti] + off);
            const u32x2 *p2 = reinterpret_cast<const u32x2 *>(copies + base_a[ti] + off + PLANE);
            const u32x2 lo1 = p1[0], hi1 = p1[1], lo2 = p2[0], hi2 = p2[1];
            Frag16 a1, a2;
            a1.u = (u32x4){lo1.x, lo1.y, hi1.x, hi1.y};
            a2.u = (u32x4){lo2.x, lo2.y, hi2.x, hi2.y};
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) {
                acc[ti][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1.h, b1[nt].h, acc[ti][nt], 0, 0, 0);
                acc[ti][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1.h, b2[nt].h, acc[ti][nt], 0, 0, 0);
                acc[ti][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a2.h, b1[nt].h, acc[ti][nt], 0, 0, 0);
            }
        }
    }

    const float inv = 1.0f / (s_x * scale_below_f16_max(*w_max_bits));
    const float bias0 = bias ? bias[n] : 0.f, bias1 = bias ? bias[16 + n] : 0.f;
#pragma unroll
    for (int ti = 0; ti < 10; ++ti)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int rho = ti / 5, mrow = (ti % 5) * 16 + 4 * g + r, tt = mrow / 20;
            const int fo = 2 * (mrow % 20) + rho, t = t0 + 4 * wave + tt;
            if (t < t_out) {
                float *out = y + ((size_t)(b * t_out + t) * Z0_FO + fo) * 32 + n;
                float v0 = acc[ti][0][r] * inv + bias0, v1 = acc[ti][1][r] * inv + bias1;
                if (cutoff > 0.f) {
                    v0 = fminf(fmaxf(v0, 0.f), cutoff);
                    v1 = fminf(fmaxf(v1, 0.f), cutoff);
                }
                out[0] = v0;
                out[16] = v1;
            }
        }
}

// x f32[B, T, 80] -> x16 [(b / 8) T][piece][80] cells of 8 halves (utterances b % 8), scaled by
// the power of two for the tensor's largest magnitude (*max_bits)
__global__ void __launch_bounds__(256)
conv0_pack_x_kernel(const float *__restrict__ x, const unsigned *__restrict__ max_bits,
                    u32x4 *__restrict__ x16, int B, long per_utt /* T 80 */, long cells) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= cells) return;
    const long within = i % per_utt;
    const int b0 = (int)(i / per_utt) * 8;
    const float s = scale_below_f16_max(*max_bits);
    unsigned p[8];
#pragma unroll
    for (int e = 0; e < 8; ++e)
        p[e] = b0 + e < B ? f16_pieces(x[(long)(b0 + e) * per_utt + within] * s) : 0u;
    const long line = i / Z0_FI;                      // (bblk, t)
    store_pieces8(x16 + line * 2 * Z0_FI + (i % Z0_FI), Z0_FI, p);
}

__global__ void __launch_bounds__(256)
conv0_wrw16_kernel(const u32x4 *__restrict__ x16, const u32x4 *__restrict__ dz16,
                   float *__restrict__ partial, int nblk, int T, int t_out, int pt0) {
    extern __shared__ __attribute__((aligned(16))) unsigned char zsm[];
    u32x4 *xl = reinterpret_cast<u32x4 *>(zsm);                 // [kt][piece][Z0_PW] cells
    u32x4 *dzs = xl + Z0_KT * 2 * Z0_PW;                        // [fo][piece][32 co] cells
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = lane & 15, g = lane >> 4;
    const int tiles = nblk * t_out;
    constexpr int X_CELLS = Z0_KT * 2 * Z0_FI;                  // 1760 real cells per tile
    constexpr int X_PER = (X_CELLS + 255) / 256, DZ_PER = Z0_FO * 64 / 256;

    for (int i = tid; i < Z0_KT * 2 * Z0_PW; i += 256) xl[i] = (u32x4){0u, 0u, 0u, 0u};

    f32x4 acc[3][3][2];                      // [kt = wave + 4 k][kf tile][co tile]
#pragma unroll
    for (int k = 0; k < 3; ++k)
#pragma unroll
        for (int nt = 0; nt < 3; ++nt) {
            acc[k][nt][0] = (f32x4){0.f, 0.f, 0.f, 0.f};
            acc[k][nt][1] = (f32x4){0.f, 0.f, 0.f, 0.f};
        }

    u32x4 rx[X_PER], rdz[DZ_PER];
    auto fetch = [&](int tile) {             // tile = t * nblk + bblk
        const int t = tile / nblk, bblk = tile % nblk;
#pragma unroll
        for (int j = 0; j < X_PER; ++j) {
            const int i = tid + j * 256, kt = i / (2 * Z0_FI), rest = i % (2 * Z0_FI);
            const int ts = 2 * t + kt - pt0;
            rx[j] = (u32x4){0u, 0u, 0u, 0u};
            if (i < X_CELLS && ts >= 0 && ts < T)
                rx[j] = x16[((size_t)bblk * T + ts) * 2 * Z0_FI + rest];
        }
        const u32x4 *dsrc = dz16 + ((size_t)bblk * t_out + t) * Z0_FO * 64;
#pragma unroll
        for (int j = 0; j < DZ_PER; ++j) rdz[j] = dsrc[tid + j * 256];
    };
    auto stage = [&]() {
#pragma unroll
        for (int j = 0; j < X_PER; ++j) {
            const int i = tid + j * 256, kt = i / (2 * Z0_FI), rest = i % (2 * Z0_FI);
            if (i < X_CELLS)
                xl[(kt * 2 + rest / Z0_FI) * Z0_PW + 19 + rest % Z0_FI] = rx[j];
        }
#pragma unroll
        for (int j = 0; j < DZ_PER; ++j) dzs[tid + j * 256] = rdz[j];
    };

    int tile = blockIdx.x;
    if (tile < tiles) fetch(tile);
    while (tile < tiles) {
        __syncthreads();
        stage();
        __syncthreads();
        const int next = tile + gridDim.x;
        if (next < tiles) fetch(next);
#pragma unroll 1
        for (int q = 0; q < Z0_FO / 4; ++q) {
            const int slot = 4 * q + g;
            Frag16 a1[2], a2[2];
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) {
                a1[mt].u = dzs[(slot * 2 + 0) * 32 + mt * 16 + n];
                a2[mt].u = dzs[(slot * 2 + 1) * 32 + mt * 16 + n];
            }
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const int kt = wave + 4 * k;
                if (kt < Z0_KT) {                              // wave-uniform
                    const u32x4 *xb = xl + (kt * 2) * Z0_PW + 2 * slot + n;
#pragma unroll
                    for (int nt = 0; nt < 3; ++nt) {
                        Frag16 b1, b2;
                        b1.u = xb[16 * nt];
                        b2.u = xb[16 * nt + Z0_PW];
#pragma unroll
                        for (int mt = 0; mt < 2; ++mt) {
                            acc[k][nt][mt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(
                                a1[mt].h, b1.h, acc[k][nt][mt], 0, 0, 0);
                            acc[k][nt][mt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(
                                a1[mt].h, b2.h, acc[k][nt][mt], 0, 0, 0);
                            acc[k][nt][mt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(
                                a2[mt].h, b1.h, acc[k][nt][mt], 0, 0, 0);
                        }
                    }
                }
            }
        }
        tile = next;
    }
    float *out = partial + (size_t)blockIdx.x * Z0_DW;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const int kt = wave + 4 * k;
        if (kt < Z0_KT) {
#pragma unroll
            for (int nt = 0; nt < 3; ++nt) {
                const int kf = 16 * nt + n;
                if (kf < Z0_KF) {
#pragma unroll
                    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                            out[((16 * mt + 4 * g + r) * Z0_KT + kt) * Z0_KF + kf] =
                                acc[k][nt][mt][r];
                }
            }
        }
    }
}

// dw[co][kt][kf] = sum of the workgroups' partials, scales out (ch_bits: per output channel,
// x_bits: the features')
__global__ void __launch_bounds__(256)
conv0_wrw16_reduce_kernel(const float *__restrict__ partial, const unsigned *__restrict__ ch_bits,
                          const unsigned *__restrict__ x_bits, float *__restrict__ dw, int parts) {
    // 64 elements x 4 bands of parts per workgroup (one thread per element alone would walk all
    // parts serially on 57 workgroups)
    __shared__ float band_sum[4][64];
    const int e = threadIdx.x & 63, band = threadIdx.x >> 6, i = blockIdx.x * 64 + e;
    float sum = 0.f;
    if (i < Z0_DW) {
#pragma unroll 8
        for (int part = band; part < parts; part += 4) sum += partial[(size_t)part * Z0_DW + i];
    }
    band_sum[band][e] = sum;
    __syncthreads();
    if (band == 0 && i < Z0_DW)
        dw[i] = (band_sum[0][e] + band_sum[1][e] + (band_sum[2][e] + band_sum[3][e])) /
                (scale_below_f16_max(ch_bits[i / (Z0_KT * Z0_KF)]) * scale_below_f16_max(*x_bits));
}

struct Z0Workspace {
    size_t x16, dz16, partial, total;
    int parts;
};
Z0Workspace z0_workspace(int B, int T) {
    const size_t nblk = (size_t)(B + 7) / 8, t_out = (size_t)(T + 1) / 2;
    Z0Workspace w;
    w.x16 = 512;                                       // [0, 128): channel maxima; [256]: max |x|
    w.dz16 = w.x16 + nblk * T * 2 * Z0_FI * 16;
    w.partial = w.dz16 + nblk * t_out * Z0_FO * 64 * 16;
    const size_t tiles = nblk * t_out;
    w.parts = (int)(tiles < (size_t)Z0_WRW_GRID ? tiles : (size_t)Z0_WRW_GRID);
    w.total = w.partial + (size_t)w.parts * Z0_DW * 4;
    return w;
}

int conv0_pt0(int T) {
    const int t_out = (T + 1) / 2;
    const int total = (t_out - 1) * 2 + Z0_KT - T;          // TensorFlow SAME padding
    return total > 0 ? total / 2 : 0;
}

}  // namespace

extern "C" size_t ctcasr_conv0_pack16_bytes(void) { return 16 + (size_t)Z0_W_CELLS * 16; }

// w [32, 1, 11, 41] -> `packed16`: the bit pattern of max |w| (16 bytes reserved), then the fp16
// pieces of w * s_w in the forward kernel's fragment order.
extern "C" int ctcasr_conv0_pack_weights16(const float *w, void *packed16, ctcasr_stream_t stream) {
    if (!w || !packed16) return CTCASR_ERR_BAD_ARGUMENT;
    hipStream_t s = (hipStream_t)stream;
    unsigned *max_bits = reinterpret_cast<unsigned *>(packed16);
    if (hipMemsetAsync(max_bits, 0, 16, s) != hipSuccess) return CTCASR_ERR_LAUNCH;
    conv16_absmax_kernel<<<8, 256, 0, s>>>(w, Z0_DW, max_bits);
    conv0_pack_w16_kernel<<<(Z0_QS * 2 * 64 + 255) / 256, 256, 0, s>>>(
        w, max_bits, reinterpret_cast<u32x4 *>(reinterpret_cast<char *>(packed16) + 16));
    return ctcasr_launch_status();
}

// y = conv(x) + bias like ctcasr_conv0_fwd, the products as fp16 x 3; no bound on x is assumed
// (a power-of-two scale per workgroup patch, found while it is staged).
extern "C" int ctcasr_conv0_fwd16(const float *x, const void *packed16, const float *bias, float *y,
                                  int B, int T, float relu_cutoff, ctcasr_stream_t stream) {
    if (!x || !packed16 || !y || B <= 0 || T <= 0) return CTCASR_ERR_BAD_ARGUMENT;
    if (B > 65535) return CTCASR_ERR_UNSUPPORTED;
    const int t_out = (T + 1) / 2;
    if (hipFuncSetAttribute(reinterpret_cast<const void *>(&conv0_fwd16_kernel),
                            hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)Z0_FWD_LDS) != hipSuccess)
        return CTCASR_ERR_LAUNCH;
    dim3 grid((t_out + Z0_TT - 1) / Z0_TT, B);
    conv0_fwd16_kernel<<<grid, 256, Z0_FWD_LDS, (hipStream_t)stream>>>(
        x, reinterpret_cast<const u32x4 *>(reinterpret_cast<const char *>(packed16) + 16),
        reinterpret_cast<const unsigned *>(packed16), bias, y, T, t_out, conv0_pt0(T), relu_cutoff);
    return ctcasr_launch_status();
}

extern "C" size_t ctcasr_conv0_wrw16_workspace_bytes(int B, int T) {
    if (B <= 0 || T <= 0) return 0;
    return z0_workspace(B, T).total;
}

// dw = kernel gradient like ctcasr_conv0_wrw, the products as fp16 x 3 (dz scaled per output
// channel, x by the tensor's largest magnitude, both found on the device).
extern "C" int ctcasr_conv0_wrw16(const float *dz, const float *x, float *dw, int B, int T,
                                  const float *act, float relu_cutoff, float *dbias,
                                  void *workspace, size_t workspace_bytes,
                                  ctcasr_stream_t stream) {
    if (!dz || !x || !dw || B <= 0 || T <= 0 || (act && relu_cutoff <= 0.f))
        return CTCASR_ERR_BAD_ARGUMENT;
    const Z0Workspace w = z0_workspace(B, T);
    if (!workspace || workspace_bytes < w.total) return CTCASR_ERR_WORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    char *base = reinterpret_cast<char *>(workspace);
    unsigned *ch_bits = reinterpret_cast<unsigned *>(base);
    unsigned *x_bits = reinterpret_cast<unsigned *>(base + 256);
    u32x4 *x16 = reinterpret_cast<u32x4 *>(base + w.x16);
    u32x4 *dz16 = reinterpret_cast<u32x4 *>(base + w.dz16);
    float *partial = reinterpret_cast<float *>(base + w.partial);
    const int t_out = (T + 1) / 2, nblk = (B + 7) / 8;
    if (hipMemsetAsync(base, 0, 512, s) != hipSuccess) return CTCASR_ERR_LAUNCH;
    const long cells4 = (long)B * t_out * Z0_FO * 32 / 4;
    wrw16_colmax_kernel<<<240, 256, 0, s>>>(dz, act, relu_cutoff, cells4, 32, ch_bits, dbias);
    const long n_x = (long)B * T * Z0_FI;
    conv16_absmax_kernel<<<512, 256, 0, s>>>(x, (int)n_x, x_bits);
    const long per_utt = (long)T * Z0_FI, xcells = (long)nblk * per_utt;
    conv0_pack_x_kernel<<<(unsigned)((xcells + 255) / 256), 256, 0, s>>>(x, x_bits, x16, B, per_utt,
                                                                        xcells);
    const long dcells = (long)nblk * t_out * Z0_FO * 32;
    wrw16_pack_dz_kernel<<<(unsigned)((dcells + 255) / 256), 256, 0, s>>>(
        dz, act, relu_cutoff, ch_bits, dz16, B, t_out, Z0_FO * 32, 32, 0, dcells);
    if (hipFuncSetAttribute(reinterpret_cast<const void *>(&conv0_wrw16_kernel),
                            hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)Z0_WRW_LDS) != hipSuccess)
        return CTCASR_ERR_LAUNCH;
    conv0_wrw16_kernel<<<w.parts, 256, Z0_WRW_LDS, s>>>(x16, dz16, partial, nblk, T, t_out,
                                                        conv0_pt0(T));
    conv0_wrw16_reduce_kernel<<<(Z0_DW + 63) / 64, 256, 0, s>>>(partial, ch_bits, x_bits, dw,
                                                                  w.parts);
    return ctcasr_launch_status();
}
