// The 11 x 21, stride (1, 2) convolutions over 32 input channels (conv.hip) with their products on
// the fp16 matrix pipe: FORWARD pass (round 4).
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

bool covered16(int freq_in, int cout) {
    return (freq_in == 40 && cout == 32) || (freq_in == 20 && cout == 96);
}

}  // namespace

// bytes of the fragment-ordered fp16 pieces of a layer's kernel (+ 16 for the magnitude word)
extern "C" size_t ctcasr_conv_s12_pack16_bytes(int cout) {
    if (cout != 32 && cout != 96) return 0;
    return (size_t)C16_KT * C16_KF * (cout / 16) * 2 * 64 * 16 + 16;
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
    return ctcasr_launch_status();
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
