// The 11 x 21, stride (1, 2) convolutions over 32 input channels of the DS2 stack (layers 2 and 3 of
// tf_contrib.conv_layers, asr/util/tf_contrib.py:64-146: 32 -> 32 channels on 40 input frequencies,
// 32 -> 96 on 20) as implicit GEMMs on the fp32 MFMA units: forward pass and data gradient.
//
// MIOpen's best kernels for the 32 -> 32 layer reach 81 TFLOP/s forward and ~60 TFLOP/s
// backward-data, the latter plus a zero-fill of its padded output and a strided copy of the
// interior afterwards (2.5 ms of the 25 ms C2 step together).  Here the padding never exists:
// out-of-range taps read zeros from the LDS patch and results go straight to the unpadded NHWC
// tensor.  Any number of frames (grid = ceil(T / TT) x B), so variable-length batches need no
// fixed-shape tiling for these layers.
//
//   forward   y[b, t, fo, co] = bias[co] + sum_{kt,kf,ci} x[b, t+kt-5, 2fo+kf-9, ci] w[co,ci,kt,kf]
//   backward  dx[b, t, f, ci] = sum_{kt,kf,co} dz[b, t+5-kt, (f+9-kf)/2, co] w[co,ci,kt,kf]
//             (terms with f + 9 - kf odd or indices out of range vanish)
//
// One workgroup = one utterance x TT output frames x all frequencies x all channels; its input
// slice sits in LDS as [TT + 10 frames][FO + 10 positions][36-float pitch] (32 channels per cell;
// the pitch makes the 16-byte fragment reads of 16 neighbouring rows conflict-free).  The stride-2
// frequency axis is handled by parity.  In padded coordinates fp = 2 fo + kf, so
//   forward:  a tap of parity par = kf & 1 only reads input frequencies of that parity - the patch
//             holds one parity plane at a time, position = fo + kf / 2;
//   backward: output rows of equal frequency parity see the same kf taps - M tiles are built per
//             parity, dz position = j + 4 + p - kf / 2 for output frequency f = 2 j + p;
// either way the A fragment of a row is "lane base address + tap-uniform offset".  A wave owns
// TT / 4 frames = 80 rows = 5 M tiles.  The weights are re-packed once per step into fragment
// order (L2 resident); the B fragments of a tap are shared by the wave's 5 M tiles.  With 96
// output channels the backward pass stages dz in three 32-channel passes (the patch would not
// fit otherwise) and keeps accumulating.
#include "common.h"

namespace {

// Backward of the fused epilogue min(max(z, 0), cutoff): the gradient passes where the stored
// OUTPUT lies strictly inside (0, cutoff).  Applied to dz while it is staged, so the backward
// kernels take the upstream gradient as it is (no elementwise pass in front of them); `act` NULL
// = dz is the pre-activation gradient already.
__device__ __forceinline__ float4 mask_dz(float4 g, const float *act, size_t index4, float upper) {
    if (act) {
        const float4 v = reinterpret_cast<const float4 *>(act)[index4];
        g.x = (v.x > 0.f && v.x < upper) ? g.x : 0.f;
        g.y = (v.y > 0.f && v.y < upper) ? g.y : 0.f;
        g.z = (v.z > 0.f && v.z < upper) ? g.z : 0.f;
        g.w = (v.w > 0.f && v.w < upper) ? g.w : 0.f;
    }
    return g;
}

constexpr int CV_CIN = 32;
constexpr int CV_KT = 11, CV_KF = 21;
constexpr int CV_PITCH = 36;       // floats per (frame, position) cell of the patch

typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void mma4(f32x4 &acc, const float4 &a, const float4 &b) {
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, b.x, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, b.y, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, b.z, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, b.w, acc, 0, 0, 0);
}

template <int COUT, int FI>
struct Geometry {
    static constexpr int FO = FI / 2;            // output frequencies
    static constexpr int TT = FO == 20 ? 16 : 32;   // output frames per workgroup: 80 rows per wave
    static constexpr int PF = FO + 10;           // patch positions per frame
    static constexpr int PT = TT + CV_KT - 1;    // patch frames
    static constexpr int NT = COUT / 16;         // N tiles forward, K chunks backward
    static constexpr size_t LDS = (size_t)PT * PF * CV_PITCH * sizeof(float);
    static_assert((TT / 4) * FO == 80, "a wave owns 5 M tiles");
};

// Packed weights, forward order: [kt][kf][q (2)][kg (4)][co (COUT)][r (4)], ci = 16 q + 4 kg + r:
// lane (n = co & 15, kg) of N tile co / 16 reads one float4 per K chunk q.
// Backward order: [kt][kf][qc (COUT / 16)][kg (4)][ci (32)][r (4)], co = 16 qc + 4 kg + r.
// w is [COUT, 32, kt, kf] (the arena's compute layout).
__global__ void conv_pack_kernel(const float *__restrict__ w, float *__restrict__ bwd,
                                 float *__restrict__ fwd, int cout) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int per_tap = CV_CIN * cout;
    if (i >= CV_KT * CV_KF * per_tap) return;
    const int tap = i / per_tap, rest = i % per_tap, kf = tap % CV_KF, kt = tap / CV_KF;
    const int r = rest & 3;
    {   // backward order
        const int ci = (rest >> 2) & 31, kg = (rest >> 7) & 3, qc = rest >> 9;
        const int co = 16 * qc + 4 * kg + r;
        bwd[i] = w[((co * CV_CIN + ci) * CV_KT + kt) * CV_KF + kf];
    }
    {   // forward order
        const int co = (rest >> 2) % cout, kg = ((rest >> 2) / cout) & 3, q = (rest >> 2) / (4 * cout);
        const int ci = 16 * q + 4 * kg + r;
        fwd[i] = w[((co * CV_CIN + ci) * CV_KT + kt) * CV_KF + kf];
    }
}

// ---------------------------------------------------------------------------------------------
// forward
// ---------------------------------------------------------------------------------------------
template <int COUT, int FI>
__global__ void __launch_bounds__(256)
conv_fwd_kernel(const float *__restrict__ x, const float4 *__restrict__ wp,
                const float *__restrict__ bias, float *__restrict__ y, int T, float cutoff,
                int y_time_major) {
    using G = Geometry<COUT, FI>;
    constexpr int NT = G::NT;
    extern __shared__ __attribute__((aligned(16))) float patch[];   // [PT][PF][CV_PITCH]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int t0 = blockIdx.x * G::TT, b = blockIdx.y;
    const int kg = lane >> 4, n = lane & 15;
    float4 *patch4 = reinterpret_cast<float4 *>(patch);

    int base_a[5];                      // float index of (frame, position) of this lane's row
#pragma unroll
    for (int ti = 0; ti < 5; ++ti) {
        const int row = ti * 16 + n, tt = row / G::FO, fo = row % G::FO;
        base_a[ti] = (((G::TT / 4) * wave + tt) * G::PF + fo) * CV_PITCH + 4 * kg;
    }
    f32x4 acc[5][NT];
#pragma unroll
    for (int ti = 0; ti < 5; ++ti)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[ti][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // the 2 * NT B fragments of tap (kt, kf): K chunk q, N tile nt
    auto load_b = [&](int kt, int kf, float4 (&dst)[2][NT]) {
        const float4 *wt = wp + (size_t)((kt * CV_KF + kf) * 8 + kg) * COUT + n;
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) dst[q][nt] = wt[q * 4 * COUT + nt * 16];
    };

#pragma unroll
    for (int par = 0; par < 2; ++par) {    // unrolled: compile-time tap counts in both copies
        if (par) __syncthreads();          // everyone is done reading the other plane
        for (int i = tid; i < G::PT * G::PF * 8; i += 256) {
            const int c4 = i & 7, pos = (i >> 3) % G::PF, pr = i / (8 * G::PF);
            const int ts = t0 - 5 + pr, fi = 2 * pos + par - 9;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (ts >= 0 && ts < T && fi >= 0 && fi < FI)
                v = reinterpret_cast<const float4 *>(x)[((size_t)(b * T + ts) * FI + fi) * 8 + c4];
            patch4[((pr * G::PF + pos) * CV_PITCH) / 4 + c4] = v;
        }
        __syncthreads();
        const int taps = par == 0 ? 11 : 10;
        // B fragments (weights, from L2) run one tap ahead of the MFMAs that use them; the
        // scheduling barriers keep hipcc from sinking the loads next to their first use
        float4 cur[2][NT], nxt[2][NT];
        load_b(0, par, cur);
        for (int kt = 0; kt < CV_KT; ++kt) {
#pragma unroll
            for (int m = 0; m < taps; ++m) {
                const bool wrap = m + 1 == taps;
                load_b(wrap ? min(kt + 1, CV_KT - 1) : kt, wrap ? par : 2 * (m + 1) + par, nxt);
                __builtin_amdgcn_sched_barrier(0);
                const int tap_off = (kt * G::PF + m) * CV_PITCH;
#pragma unroll
                for (int ti = 0; ti < 5; ++ti) {
                    const float4 a0 = *reinterpret_cast<const float4 *>(patch + base_a[ti] + tap_off);
                    const float4 a1 =
                        *reinterpret_cast<const float4 *>(patch + base_a[ti] + tap_off + 16);
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) mma4(acc[ti][nt], a0, cur[0][nt]);
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) mma4(acc[ti][nt], a1, cur[1][nt]);
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int q = 0; q < 2; ++q)
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) cur[q][nt] = nxt[q][nt];
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
                // epilogue: bias, then min(max(., 0), cutoff) (tf_contrib.conv_layers' ReLU +
                // tf.minimum) when cutoff > 0; the last layer of the stack writes time-major
                // [T, B, FO, COUT] - the layout the recurrent stack reads - instead of NHWC
                const size_t cell = y_time_major ? (size_t)t * gridDim.y + b : (size_t)b * T + t;
                float *out = y + (cell * G::FO + fo) * COUT + n;
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    float v = acc[ti][nt][r] + bias_v[nt];
                    if (cutoff > 0.f) v = fminf(fmaxf(v, 0.f), cutoff);
                    out[nt * 16] = v;
                }
            }
        }
}

// ---------------------------------------------------------------------------------------------
// backward data
// ---------------------------------------------------------------------------------------------
template <int COUT, int FI>
__global__ void __launch_bounds__(256)
conv_bwd_data_kernel(const float *__restrict__ dz, const float4 *__restrict__ wp,
                     float *__restrict__ dx, int T, int dz_time_major,
                     const float *__restrict__ act, float upper) {
    using G = Geometry<COUT, FI>;
    constexpr int PASSES = COUT / 32;           // 32 dz channels staged at a time
    extern __shared__ __attribute__((aligned(16))) float patch[];   // [PT][PF][CV_PITCH]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int t0 = blockIdx.x * G::TT, b = blockIdx.y;
    const int kg = lane >> 4, n = lane & 15;
    float4 *patch4 = reinterpret_cast<float4 *>(patch);

    int base_a[5];
#pragma unroll
    for (int ti = 0; ti < 5; ++ti) {
        const int row = ti * 16 + n, tt = row / G::FO, j = row % G::FO;
        base_a[ti] = (((G::TT / 4) * wave + tt) * G::PF + j) * CV_PITCH + 4 * kg;
    }
    f32x4 acc[2][5][2];
#pragma unroll
    for (int p = 0; p < 2; ++p)
#pragma unroll
        for (int ti = 0; ti < 5; ++ti)
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) acc[p][ti][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};

#pragma unroll 1
    for (int pass = 0; pass < PASSES; ++pass) {
        // ---- stage dz[b, t0-5 .. , :, 32 pass .. 32 pass + 31] with 5 zero positions each side ----
        if (pass) __syncthreads();
        for (int i = tid; i < G::PT * G::PF * 8; i += 256) {
            const int c4 = i & 7, pos = (i >> 3) % G::PF, pr = i / (8 * G::PF);
            const int ts = t0 - 5 + pr, fo = pos - 5;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (ts >= 0 && ts < T && fo >= 0 && fo < G::FO) {
                const size_t cell = dz_time_major ? (size_t)ts * gridDim.y + b
                                                  : (size_t)b * T + ts;
                const size_t at = (cell * G::FO + fo) * (COUT / 4) + pass * 8 + c4;
                v = mask_dz(reinterpret_cast<const float4 *>(dz)[at], act, at, upper);
            }
            patch4[((pr * G::PF + pos) * CV_PITCH) / 4 + c4] = v;
        }
        __syncthreads();

        // ---- taps (no explicit weight prefetch: hipcc's own schedule of this loop nest reaches
        // 138 TFLOP/s on the 32 -> 32 layer, the pinned one-tap-ahead variant 131) -------------
        for (int kt = 0; kt < CV_KT; ++kt) {
#pragma unroll
            for (int p = 0; p < 2; ++p) {
                // even output frequencies (f = 2j) meet the odd kf, odd ones (f = 2j+1) the even
                // kf; dz position = j + 4 + p - m (+5 for the zero border) with m = kf / 2
                const int taps = p == 0 ? 10 : 11;
                for (int m = 0; m < taps; ++m) {
                    const int kf = p == 0 ? 2 * m + 1 : 2 * m;
                    // K chunks 2 pass, 2 pass + 1 of this tap; N tiles ci 0..15, 16..31
                    const float4 *wt =
                        wp + (size_t)(((kt * CV_KF + kf) * (COUT / 16) + 2 * pass) * 4 + kg) * 32 + n;
                    const float4 b00 = wt[0], b01 = wt[16], b10 = wt[4 * 32], b11 = wt[4 * 32 + 16];
                    const int tap_off = ((10 - kt) * G::PF + 9 + p - m) * CV_PITCH;
#pragma unroll
                    for (int ti = 0; ti < 5; ++ti) {
                        const float4 a0 =
                            *reinterpret_cast<const float4 *>(patch + base_a[ti] + tap_off);
                        const float4 a1 =
                            *reinterpret_cast<const float4 *>(patch + base_a[ti] + tap_off + 16);
                        mma4(acc[p][ti][0], a0, b00);
                        mma4(acc[p][ti][1], a0, b01);
                        mma4(acc[p][ti][0], a1, b10);
                        mma4(acc[p][ti][1], a1, b11);
                    }
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
                    float *out = dx + ((size_t)(b * T + t) * FI + 2 * j + p) * CV_CIN + n;
                    out[0] = acc[p][ti][0][r];
                    out[16] = acc[p][ti][1][r];
                }
            }
}

template <int COUT, int FI>
int launch_fwd(const float *x, const float *packed, const float *bias, float *y, int B, int T,
               float cutoff, int y_time_major, hipStream_t s) {
    using G = Geometry<COUT, FI>;
    if (hipFuncSetAttribute(reinterpret_cast<const void *>(&conv_fwd_kernel<COUT, FI>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)G::LDS) != hipSuccess)
        return CTCASR_ERR_LAUNCH;
    dim3 grid((T + G::TT - 1) / G::TT, B);
    const size_t per_order = (size_t)CV_KT * CV_KF * CV_CIN * COUT;
    conv_fwd_kernel<COUT, FI><<<grid, 256, G::LDS, s>>>(
        x, reinterpret_cast<const float4 *>(packed + per_order), bias, y, T, cutoff, y_time_major);
    return ctcasr_launch_status();
}

template <int COUT, int FI>
int launch_bwd(const float *dz, const float *packed, float *dx, int B, int T, int dz_time_major,
               const float *act, float upper, hipStream_t s) {
    using G = Geometry<COUT, FI>;
    if (hipFuncSetAttribute(reinterpret_cast<const void *>(&conv_bwd_data_kernel<COUT, FI>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)G::LDS) != hipSuccess)
        return CTCASR_ERR_LAUNCH;
    dim3 grid((T + G::TT - 1) / G::TT, B);
    conv_bwd_data_kernel<COUT, FI><<<grid, 256, G::LDS, s>>>(
        dz, reinterpret_cast<const float4 *>(packed), dx, T, dz_time_major, act, upper);
    return ctcasr_launch_status();
}

bool covered(int freq_in, int cout) {
    return (freq_in == 40 && cout == 32) || (freq_in == 20 && cout == 96);
}

// ---------------------------------------------------------------------------------------------
// Kernel gradient of the 11 x 21, stride (1, 2) layers:
//   dw[co, ci, kt, kf] = sum_{b,t,fo} dz[b, t, fo, co] * x[b, t + kt - 5, 2 fo + kf - 9, ci]
// Per tap a GEMM  [co x rows] x [rows x ci]  with rows = (b, t, fo) as the K axis; 231 taps.
// grid = (splits of the K axis, 11 kt, cout / 32): a workgroup owns ONE kt row (for a fixed kt
// the x frames a tile of output frames needs are just that tile shifted - no time halo) and one
// group of 32 output channels, and walks its share of the (b, frame-tile) list.  Per tile of 80
// rows it stages dz [80][32 (+16: pitch 48)] and the x rows [frames][FI + 20 positions][32 (+8:
// pitch 40)] in LDS - the pitches make the scalar fragment reads of 4 consecutive rows x 16
// channels conflict-free.  Operands are fed with one ds_read_b32 per lane (16-byte fragments would
// need the rows of a tap contiguous and aligned, which the kf shift rules out); a wave owns one
// of the four 16 x 16 (co, ci) tiles for ALL 21 kf taps: per group of 4 rows it reads its A value
// (dz) once and 21 B values (x, one immediate offset per tap) for 21 MFMAs - balanced over the
// waves, 84 accumulator registers.  The next tile's global loads are issued before the MFMAs of
// the current one.  Partial results go to a workspace [split][cout/32][kt][kf][32 co][32 ci];
// conv_wrw_reduce_kernel sums the splits in a fixed order (deterministic) into
// dw [cout, 32, 11, 21].  MIOpen's kernel for this reaches 61-73 TFLOP/s.
// ---------------------------------------------------------------------------------------------
constexpr int WR_DZP = 48;          // dz pitch in LDS (floats)
constexpr int WR_XP = 40;           // x pitch per (frame, position) cell
constexpr int WR_ROWS = 80;         // rows (frame, fo) per tile
constexpr int WR_TAP = CV_CIN * 32; // floats per tap and channel group in the partial layout

template <int FI>
struct WrwGeometry {
    static constexpr int FO = FI / 2;
    static constexpr int TT = WR_ROWS / FO;          // output frames per tile (4 or 8)
    static constexpr int PF = FI + 20;               // positions 2 fo + kf in [0, FI + 18]
    static constexpr int DZ4 = WR_ROWS * 8;          // float4 loads per dz tile
    static constexpr int X4 = TT * PF * 8;           // float4 slots per x tile
    static constexpr int DZ_PER = (DZ4 + 255) / 256, X_PER = (X4 + 255) / 256;
    static constexpr size_t LDS =
        ((size_t)WR_ROWS * WR_DZP + (size_t)TT * PF * WR_XP) * sizeof(float);
};

template <int FI>
__global__ void __launch_bounds__(256, 2)
conv_wrw_kernel(const float *__restrict__ dz, const float *__restrict__ x,
                float *__restrict__ partial, int B, int T, int cout, int dz_time_major,
                const float *__restrict__ act, float upper, float *__restrict__ dbias) {
    using G = WrwGeometry<FI>;
    extern __shared__ __attribute__((aligned(16))) float wsm[];
    float *dzs = wsm;                                   // [WR_ROWS][WR_DZP]
    float *xs = wsm + WR_ROWS * WR_DZP;                 // [TT][PF][WR_XP]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int split = blockIdx.x, nsplit = gridDim.x, kt = blockIdx.y, cg = blockIdx.z;
    const int g = lane >> 4, n = lane & 15;
    const int co_tile = wave >> 1, ci_tile = wave & 1;
    const int tiles_per_b = (T + G::TT - 1) / G::TT, tiles = B * tiles_per_b;

    f32x4 acc[CV_KF];
#pragma unroll
    for (int kf = 0; kf < CV_KF; ++kf) acc[kf] = (f32x4){0.f, 0.f, 0.f, 0.f};

    float4 rdz[G::DZ_PER], rx[G::X_PER];
    // bias gradient = column sums of the (masked) dz: every dz element is staged by the 11
    // workgroups of its kt rows, the kt = 0 ones add it up (thread: channels 4 c4 .. 4 c4 + 3)
    const bool sum_bias = dbias != nullptr && kt == 0;
    float4 bsum = make_float4(0.f, 0.f, 0.f, 0.f);
    auto fetch = [&](int tile) {                 // global -> registers (zeros outside the tensor)
        const int b = tile / tiles_per_b, t0 = (tile % tiles_per_b) * G::TT;
#pragma unroll
        for (int j = 0; j < G::DZ_PER; ++j) {
            const int i = tid + j * 256, c4 = i & 7, row = i >> 3;
            const int t = t0 + row / G::FO, fo = row % G::FO;
            rdz[j] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (i < G::DZ4 && t < T) {
                const size_t cell = dz_time_major ? (size_t)t * B + b : (size_t)b * T + t;
                const size_t at = ((cell * G::FO + fo) * cout + cg * 32 + 4 * c4) / 4;
                rdz[j] = mask_dz(reinterpret_cast<const float4 *>(dz)[at], act, at, upper);
                if (sum_bias) {
                    bsum.x += rdz[j].x; bsum.y += rdz[j].y; bsum.z += rdz[j].z; bsum.w += rdz[j].w;
                }
            }
        }
#pragma unroll
        for (int j = 0; j < G::X_PER; ++j) {
            const int i = tid + j * 256, c4 = i & 7, pos = (i >> 3) % G::PF, tl = i / (8 * G::PF);
            const int ts = t0 + tl + kt - 5, f = pos - 9;
            rx[j] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (i < G::X4 && ts >= 0 && ts < T && f >= 0 && f < FI)
                rx[j] = *reinterpret_cast<const float4 *>(
                    x + ((size_t)(b * T + ts) * FI + f) * CV_CIN + 4 * c4);
        }
    };
    auto stage = [&]() {                         // registers -> LDS
#pragma unroll
        for (int j = 0; j < G::DZ_PER; ++j) {
            const int i = tid + j * 256;
            if (i < G::DZ4)
                *reinterpret_cast<float4 *>(dzs + (i >> 3) * WR_DZP + 4 * (i & 7)) = rdz[j];
        }
#pragma unroll
        for (int j = 0; j < G::X_PER; ++j) {
            const int i = tid + j * 256;
            if (i < G::X4)
                *reinterpret_cast<float4 *>(xs + (i >> 3) * WR_XP + 4 * (i & 7)) = rx[j];
        }
    };

    int tile = split;
    if (tile < tiles) fetch(tile);
    for (; tile < tiles; tile += nsplit) {
        __syncthreads();                         // everyone is done reading the previous tile
        stage();
        __syncthreads();
        if (tile + nsplit < tiles) fetch(tile + nsplit);
        // lane (m = n, k = g) of A: dz[row = 4 q + g][co_tile * 16 + n]
        // lane (k = g, n) of B: x[frame(row)][2 fo(row) + kf][ci_tile * 16 + n]
#pragma unroll 2
        for (int q = 0; q < WR_ROWS / 4; ++q) {
            const int row = 4 * q + g, tl = row / G::FO, fo = row % G::FO;
            const float a = dzs[row * WR_DZP + co_tile * 16 + n];
            const float *xb = xs + (tl * G::PF + 2 * fo) * WR_XP + ci_tile * 16 + n;
#pragma unroll
            for (int kf = 0; kf < CV_KF; ++kf)
                acc[kf] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, xb[kf * WR_XP], acc[kf], 0, 0, 0);
        }
    }
    if (sum_bias) {      // 32 threads per c4: fold them through LDS, one atomic per channel
        __syncthreads();
        reinterpret_cast<float4 *>(wsm)[tid] = bsum;
        __syncthreads();
        if (tid < 32) {
            float sum = 0.f;
            for (int r = 0; r < 32; ++r) sum += wsm[(r * 8 + (tid >> 2)) * 4 + (tid & 3)];
            atomicAdd(dbias + cg * 32 + tid, sum);
        }
    }
    // D[m = 4 g + r][n]: co = co_tile * 16 + 4 g + r, ci = ci_tile * 16 + n
    float *out = partial + (((size_t)split * gridDim.z + cg) * CV_KT + kt) * CV_KF * WR_TAP;
#pragma unroll
    for (int kf = 0; kf < CV_KF; ++kf)
#pragma unroll
        for (int r = 0; r < 4; ++r)
            out[(size_t)kf * WR_TAP + (co_tile * 16 + 4 * g + r) * CV_CIN + ci_tile * 16 + n] =
                acc[kf][r];
}

// dw[co][ci][kt][kf] = sum over splits of partial[split][co / 32][kt][kf][co % 32][ci]: one
// thread per element in the partials' order (coalesced reads; the results are scattered)
__global__ void conv_wrw_reduce_kernel(const float *__restrict__ partial, float *__restrict__ dw,
                                       int nsplit, int cout) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    const int total = cout * CV_CIN * CV_KT * CV_KF;
    if (j >= total) return;
    const int ci = j & 31, co32 = (j >> 5) & 31, tap = (j >> 10) % (CV_KT * CV_KF);
    const int co = (j >> 10) / (CV_KT * CV_KF) * 32 + co32;
    float sum = 0.f;
#pragma unroll 4
    for (int sp = 0; sp < nsplit; ++sp) sum += partial[(size_t)sp * total + j];
    dw[((size_t)co * CV_CIN + ci) * (CV_KT * CV_KF) + tap] = sum;
}

int wrw_splits(int B, int T, int freq_in, int cout) {
    // two workgroups per CU and not one more (a second round of workgroups would double the
    // kernel's duration: 528 workgroups took 1.05 ms, 506 take 0.6); never more splits than tiles
    const int tt = WR_ROWS / (freq_in / 2);
    const int tiles = B * ((T + tt - 1) / tt);
    const int want = 512 / (CV_KT * (cout / 32));
    return tiles < want ? tiles : want;
}

}  // namespace

extern "C" size_t ctcasr_conv_s12_wrw_workspace_bytes(int B, int T, int freq_in, int cout) {
    if (B <= 0 || T <= 0 || !covered(freq_in, cout)) return 0;
    return (size_t)wrw_splits(B, T, freq_in, cout) * cout * CV_CIN * CV_KT * CV_KF * sizeof(float);
}

// dz [B, T, freq_in / 2, cout], x [B, T, freq_in, 32] (both NHWC) -> dw [cout, 32, 11, 21]
// (overwritten); deterministic (fixed summation order).
extern "C" int ctcasr_conv_s12_wrw(const float *dz, const float *x, float *dw, int B, int T,
                                   int freq_in, int cout, int dz_time_major, const float *act,
                                   float relu_cutoff, float *dbias, void *workspace,
                                   size_t workspace_bytes, ctcasr_stream_t stream) {
    if (!dz || !x || !dw || B <= 0 || T <= 0 || (act && relu_cutoff <= 0.f))
        return CTCASR_ERR_BAD_ARGUMENT;
    if (!covered(freq_in, cout)) return CTCASR_ERR_UNSUPPORTED;
    if (!workspace || workspace_bytes < ctcasr_conv_s12_wrw_workspace_bytes(B, T, freq_in, cout))
        return CTCASR_ERR_WORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    const int nsplit = wrw_splits(B, T, freq_in, cout);
    float *partial = reinterpret_cast<float *>(workspace);
    dim3 grid(nsplit, CV_KT, cout / 32);
    if (freq_in == 40) {
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(&conv_wrw_kernel<40>),
                                hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)WrwGeometry<40>::LDS) != hipSuccess)
            return CTCASR_ERR_LAUNCH;
        conv_wrw_kernel<40><<<grid, 256, WrwGeometry<40>::LDS, s>>>(
            dz, x, partial, B, T, cout, dz_time_major, act, relu_cutoff, dbias);
    } else {
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(&conv_wrw_kernel<20>),
                                hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)WrwGeometry<20>::LDS) != hipSuccess)
            return CTCASR_ERR_LAUNCH;
        conv_wrw_kernel<20><<<grid, 256, WrwGeometry<20>::LDS, s>>>(
            dz, x, partial, B, T, cout, dz_time_major, act, relu_cutoff, dbias);
    }
    const int total = cout * CV_CIN * CV_KT * CV_KF;
    conv_wrw_reduce_kernel<<<(total + 255) / 256, 256, 0, s>>>(partial, dw, nsplit, cout);
    return ctcasr_launch_status();
}

// 1 for the (input frequencies, output channels) pairs the kernels are instantiated for: the
// second (40, 32) and third (20, 96) convolution of the reference's stack.
extern "C" int ctcasr_conv_s12_supported(int freq_in, int cout) { return covered(freq_in, cout); }

// Fragment-ordered copies of a layer's kernel w [cout, 32, 11, 21]: `packed` holds
// 2 * 11*21*32*cout floats (backward order, then forward order).  The weights change every step.
extern "C" int ctcasr_conv_s12_pack_weights(const float *w, float *packed, int cout,
                                            ctcasr_stream_t stream) {
    if (!w || !packed || (cout != 32 && cout != 96)) return CTCASR_ERR_BAD_ARGUMENT;
    const int n = CV_KT * CV_KF * CV_CIN * cout;
    conv_pack_kernel<<<(n + 255) / 256, 256, 0, (hipStream_t)stream>>>(w, packed, packed + n, cout);
    return ctcasr_launch_status();
}

// x [B, T, freq_in, 32] (NHWC) -> y [B, T, freq_in / 2, cout] = conv(x) + bias (bias may be NULL).
extern "C" int ctcasr_conv_s12_fwd(const float *x, const float *packed, const float *bias, float *y,
                                   int B, int T, int freq_in, int cout, float relu_cutoff,
                                   int y_time_major, ctcasr_stream_t stream) {
    if (!x || !packed || !y || B <= 0 || T <= 0) return CTCASR_ERR_BAD_ARGUMENT;
    if (!covered(freq_in, cout) || B > 65535) return CTCASR_ERR_UNSUPPORTED;
    hipStream_t s = (hipStream_t)stream;
    if (cout == 32) return launch_fwd<32, 40>(x, packed, bias, y, B, T, relu_cutoff, y_time_major, s);
    return launch_fwd<96, 20>(x, packed, bias, y, B, T, relu_cutoff, y_time_major, s);
}

// dz [B, T, freq_in / 2, cout] (NHWC, gradient w.r.t. the layer's pre-activation output)
// -> dx [B, T, freq_in, 32].
extern "C" int ctcasr_conv_s12_bwd_data(const float *dz, const float *packed, float *dx, int B,
                                        int T, int freq_in, int cout, int dz_time_major,
                                        const float *act, float relu_cutoff,
                                        ctcasr_stream_t stream) {
    if (!dz || !packed || !dx || B <= 0 || T <= 0 || (act && relu_cutoff <= 0.f))
        return CTCASR_ERR_BAD_ARGUMENT;
    if (!covered(freq_in, cout) || B > 65535) return CTCASR_ERR_UNSUPPORTED;
    hipStream_t s = (hipStream_t)stream;
    if (cout == 32)
        return launch_bwd<32, 40>(dz, packed, dx, B, T, dz_time_major, act, relu_cutoff, s);
    return launch_bwd<96, 20>(dz, packed, dx, B, T, dz_time_major, act, relu_cutoff, s);
}

// =============================================================================================
// First convolution of the stack: 1 -> 32 channels, 11 x 41 taps, stride (2, 2), SAME padding
// (asr/util/tf_contrib.py:64-146, layer 1).  K = 451 taps only, so the operands are fed with
// scalar LDS reads (one float per lane and MFMA) instead of 16-byte fragments: per (kt, group of
// four kf) a wave reads its B values (weights) once and one A value per M tile.
//   y[b, t, fo, co] = bias[co] + sum_{kt,kf} x[b, 2t + kt - pt0, 2fo + kf - 19] w[co, 0, kt, kf]
// pt0 = 5 for an odd number of input frames, 4 for an even one (TensorFlow puts the odd padding
// element at the end).  One workgroup = one utterance x 16 output frames x 40 frequencies.
// =============================================================================================
namespace {

constexpr int C0_CO = 32, C0_KT = 11, C0_KF = 41, C0_KFP = 44;   // kf padded to a multiple of 4
constexpr int C0_FI = 80, C0_FO = 40, C0_TT = 16;
constexpr int C0_PW = 124;                       // patch width: 19 + 80 + 20 (+ reach of the kf pad)
constexpr int C0_PT = 2 * C0_TT + C0_KT - 2;     // patch frames: 2 (TT - 1) + 11
constexpr size_t C0_LDS = ((size_t)C0_PT * C0_PW + (size_t)C0_KT * C0_KFP * C0_CO) * sizeof(float);

__global__ void __launch_bounds__(256)
conv0_fwd_kernel(const float *__restrict__ x, const float *__restrict__ w,
                 const float *__restrict__ bias, float *__restrict__ y, int T, int t_out, int pt0,
                 float cutoff) {
    extern __shared__ __attribute__((aligned(16))) float smem0[];
    float *patch = smem0;                               // [C0_PT][C0_PW]
    float *wl = smem0 + C0_PT * C0_PW;                  // [kt][kf padded][co]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int t0 = blockIdx.x * C0_TT, b = blockIdx.y;
    const int kg = lane >> 4, n = lane & 15;

    for (int i = tid; i < C0_PT * C0_PW; i += 256) {
        const int col = i % C0_PW, pr = i / C0_PW;
        const int ts = 2 * t0 - pt0 + pr, fi = col - 19;
        patch[i] = (ts >= 0 && ts < T && fi >= 0 && fi < C0_FI)
                       ? x[((size_t)b * T + ts) * C0_FI + fi] : 0.f;
    }
    for (int i = tid; i < C0_KT * C0_KFP * C0_CO; i += 256) {
        const int co = i & 31, kf = (i >> 5) % C0_KFP, kt = i / (32 * C0_KFP);
        wl[i] = kf < C0_KF ? w[(co * C0_KT + kt) * C0_KF + kf] : 0.f;
    }
    __syncthreads();

    // a wave owns 4 output frames x 40 frequencies = 160 rows = 10 M tiles
    int base_a[10];
#pragma unroll
    for (int ti = 0; ti < 10; ++ti) {
        const int row = ti * 16 + n, tt = row / C0_FO, fo = row % C0_FO;
        base_a[ti] = 2 * (4 * wave + tt) * C0_PW + 2 * fo + kg;
    }
    f32x4 acc[10][2];
#pragma unroll
    for (int ti = 0; ti < 10; ++ti)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) acc[ti][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};

    for (int kt = 0; kt < C0_KT; ++kt) {
#pragma unroll
        for (int j = 0; j < C0_KFP / 4; ++j) {
            const float *wrow = wl + ((kt * C0_KFP + 4 * j + kg) * C0_CO) + n;
            const float b0 = wrow[0], b1 = wrow[16];
            const int off = kt * C0_PW + 4 * j;
#pragma unroll
            for (int ti = 0; ti < 10; ++ti) {
                const float a = patch[base_a[ti] + off];
                acc[ti][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b0, acc[ti][0], 0, 0, 0);
                acc[ti][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b1, acc[ti][1], 0, 0, 0);
            }
        }
    }

    const float bias0 = bias ? bias[n] : 0.f, bias1 = bias ? bias[16 + n] : 0.f;
#pragma unroll
    for (int ti = 0; ti < 10; ++ti)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = ti * 16 + 4 * kg + r, tt = row / C0_FO, fo = row % C0_FO;
            const int t = t0 + 4 * wave + tt;
            if (t < t_out) {
                float *out = y + ((size_t)(b * t_out + t) * C0_FO + fo) * C0_CO + n;
                float v0 = acc[ti][0][r] + bias0, v1 = acc[ti][1][r] + bias1;
                if (cutoff > 0.f) {          // ReLU + tf.minimum(., relu_cutoff) of conv_layers
                    v0 = fminf(fmaxf(v0, 0.f), cutoff);
                    v1 = fminf(fmaxf(v1, 0.f), cutoff);
                }
                out[0] = v0;
                out[16] = v1;
            }
        }
}

}  // namespace

// x [B, T, 80] (one channel) -> y [B, ceil(T / 2), 40, 32] (NHWC) = conv(x) + bias (bias may be
// NULL); w [32, 1, 11, 41].
extern "C" int ctcasr_conv0_fwd(const float *x, const float *w, const float *bias, float *y, int B,
                                int T, float relu_cutoff, ctcasr_stream_t stream) {
    if (!x || !w || !y || B <= 0 || T <= 0) return CTCASR_ERR_BAD_ARGUMENT;
    if (B > 65535) return CTCASR_ERR_UNSUPPORTED;
    const int t_out = (T + 1) / 2;
    const int total = (t_out - 1) * 2 + C0_KT - T;          // TensorFlow SAME padding
    const int pt0 = total > 0 ? total / 2 : 0;
    if (hipFuncSetAttribute(reinterpret_cast<const void *>(&conv0_fwd_kernel),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)C0_LDS) != hipSuccess)
        return CTCASR_ERR_LAUNCH;
    dim3 grid((t_out + C0_TT - 1) / C0_TT, B);
    conv0_fwd_kernel<<<grid, 256, C0_LDS, (hipStream_t)stream>>>(x, w, bias, y, T, t_out, pt0,
                                                                 relu_cutoff);
    return ctcasr_launch_status();
}

// ---------------------------------------------------------------------------------------------
// Kernel gradient of the first convolution:
//   dw[co, 0, kt, kf] = sum_{b,t,fo} dz[b, t, fo, co] * x[b, 2t + kt - pt0, 2fo + kf - 19]
// GEMM per kt: M = co (2 tiles), N = kf (41 -> 3 tiles), K = output positions (t, fo).  One
// workgroup = one utterance x 16 output frames = 640 positions; its dz slice (pitch 48 floats:
// conflict-free scalar reads of 4 consecutive positions x 16 channels) and x patch sit in LDS; a
// wave owns the taps kt = wave, wave + 4, wave + 8 and walks all 160 groups of 4 positions.
// Partial results go to a workspace [workgroups][32 * 11 * 41]; conv0_wrw_reduce_kernel sums them.
// ---------------------------------------------------------------------------------------------
namespace {

constexpr int C0_DZP = 48;                       // dz pitch in LDS
constexpr int C0_PW2 = 128;                      // x patch width for kf up to 47
constexpr size_t C0_WRW_LDS =
    ((size_t)C0_PT * C0_PW2 + (size_t)C0_TT * C0_FO * C0_DZP) * sizeof(float);
constexpr int C0_DW = C0_CO * C0_KT * C0_KF;     // 14432 floats

__global__ void __launch_bounds__(256)
conv0_wrw_kernel(const float *__restrict__ dz, const float *__restrict__ x,
                 float *__restrict__ partial, int T, int t_out, int pt0,
                 const float *__restrict__ act, float upper, float *__restrict__ dbias) {
    extern __shared__ __attribute__((aligned(16))) float smem0[];
    float *patch = smem0;                               // [C0_PT][C0_PW2]
    float *dzl = smem0 + C0_PT * C0_PW2;                // [640 positions][C0_DZP]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int t0 = blockIdx.x * C0_TT, b = blockIdx.y;
    const int kg = lane >> 4, n = lane & 15;

    float4 bsum = make_float4(0.f, 0.f, 0.f, 0.f);       // (thread: channels 4 c4 .. 4 c4 + 3)
    for (int i = tid; i < C0_TT * C0_FO * 8; i += 256) {
        const int c4 = i & 7, pos = i >> 3, t = t0 + pos / C0_FO;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (t < t_out) {
            const size_t at = ((size_t)(b * t_out + t) * C0_FO + pos % C0_FO) * 8 + c4;
            v = mask_dz(reinterpret_cast<const float4 *>(dz)[at], act, at, upper);
        }
        bsum.x += v.x; bsum.y += v.y; bsum.z += v.z; bsum.w += v.w;
        *reinterpret_cast<float4 *>(dzl + pos * C0_DZP + 4 * c4) = v;
    }
    if (dbias) {         // bias gradient: every dz element is staged exactly once, by this kernel
        reinterpret_cast<float4 *>(patch)[tid] = bsum;     // (scratch: the x patch comes next)
        __syncthreads();
        if (tid < 32) {
            float sum = 0.f;
            for (int r = 0; r < 32; ++r) sum += patch[(r * 8 + (tid >> 2)) * 4 + (tid & 3)];
            atomicAdd(dbias + tid, sum);
        }
        __syncthreads();
    }
    for (int i = tid; i < C0_PT * C0_PW2; i += 256) {
        const int col = i % C0_PW2, pr = i / C0_PW2;
        const int ts = 2 * t0 - pt0 + pr, fi = col - 19;
        patch[i] = (ts >= 0 && ts < T && fi >= 0 && fi < C0_FI)
                       ? x[((size_t)b * T + ts) * C0_FI + fi] : 0.f;
    }
    __syncthreads();

    f32x4 acc[3][3][2];     // [kt of this wave][kf tile][co tile]
#pragma unroll
    for (int k = 0; k < 3; ++k)
#pragma unroll
        for (int nt = 0; nt < 3; ++nt)
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) acc[k][nt][mt] = (f32x4){0.f, 0.f, 0.f, 0.f};

#pragma unroll 4
    for (int s = 0; s < C0_TT * C0_FO / 4; ++s) {
        const int pos0 = 4 * s, tl = pos0 / C0_FO, fo0 = pos0 % C0_FO;
        const float a0 = dzl[(pos0 + kg) * C0_DZP + n], a1 = dzl[(pos0 + kg) * C0_DZP + 16 + n];
        const float *xrow = patch + 2 * tl * C0_PW2 + 2 * (fo0 + kg) + n;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const int kt = wave + 4 * k;
            if (kt < C0_KT) {
#pragma unroll
                for (int nt = 0; nt < 3; ++nt) {
                    const float bv = xrow[kt * C0_PW2 + 16 * nt];
                    acc[k][nt][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, bv, acc[k][nt][0], 0, 0, 0);
                    acc[k][nt][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, bv, acc[k][nt][1], 0, 0, 0);
                }
            }
        }
    }

    float *out = partial + (size_t)(blockIdx.y * gridDim.x + blockIdx.x) * C0_DW;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const int kt = wave + 4 * k;
        if (kt >= C0_KT) continue;
#pragma unroll
        for (int nt = 0; nt < 3; ++nt) {
            const int kf = 16 * nt + n;
            if (kf >= C0_KF) continue;
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int co = 16 * mt + 4 * kg + r;
                    out[(co * C0_KT + kt) * C0_KF + kf] = acc[k][nt][mt][r];
                }
        }
    }
}

// Two deterministic stages (no atomics): grid (ceil(C0_DW / 256), C0_BANDS) sums a band of the
// partials per workgroup row into bands[band][i]; a second launch sums the bands.  One thread per
// element alone would walk all parts serially.
constexpr int C0_BANDS = 16;
__global__ void conv0_wrw_reduce_kernel(const float *__restrict__ src, float *__restrict__ dst,
                                        int parts) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= C0_DW) return;
    const int band = (parts + gridDim.y - 1) / gridDim.y;
    const int lo = blockIdx.y * band, hi = min(parts, lo + band);
    float sum = 0.f;
#pragma unroll 8
    for (int part = lo; part < hi; ++part) sum += src[(size_t)part * C0_DW + i];
    dst[(size_t)blockIdx.y * C0_DW + i] = sum;
}

}  // namespace

extern "C" size_t ctcasr_conv0_wrw_workspace_bytes(int B, int T) {
    if (B <= 0 || T <= 0) return 0;
    const int t_out = (T + 1) / 2;
    return ((size_t)B * ((t_out + C0_TT - 1) / C0_TT) + C0_BANDS) * C0_DW * sizeof(float);
}

// dz [B, ceil(T/2), 40, 32] (NHWC), x [B, T, 80] -> dw [32, 1, 11, 41] (overwritten).
extern "C" int ctcasr_conv0_wrw(const float *dz, const float *x, float *dw, int B, int T,
                                const float *act, float relu_cutoff, float *dbias,
                                void *workspace, size_t workspace_bytes, ctcasr_stream_t stream) {
    if (!dz || !x || !dw || B <= 0 || T <= 0 || (act && relu_cutoff <= 0.f))
        return CTCASR_ERR_BAD_ARGUMENT;
    if (B > 65535) return CTCASR_ERR_UNSUPPORTED;
    if (!workspace || workspace_bytes < ctcasr_conv0_wrw_workspace_bytes(B, T))
        return CTCASR_ERR_WORKSPACE;
    const int t_out = (T + 1) / 2;
    const int total = (t_out - 1) * 2 + C0_KT - T;
    const int pt0 = total > 0 ? total / 2 : 0;
    if (hipFuncSetAttribute(reinterpret_cast<const void *>(&conv0_wrw_kernel),
                            hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)C0_WRW_LDS) != hipSuccess)
        return CTCASR_ERR_LAUNCH;
    hipStream_t s = (hipStream_t)stream;
    dim3 grid((t_out + C0_TT - 1) / C0_TT, B);
    float *partial = reinterpret_cast<float *>(workspace);
    conv0_wrw_kernel<<<grid, 256, C0_WRW_LDS, s>>>(dz, x, partial, T, t_out, pt0, act,
                                                   relu_cutoff, dbias);
    const int parts = (int)(grid.x * grid.y);
    float *bands = partial + (size_t)parts * C0_DW;
    conv0_wrw_reduce_kernel<<<dim3((C0_DW + 255) / 256, C0_BANDS), 256, 0, s>>>(partial, bands, parts);
    conv0_wrw_reduce_kernel<<<dim3((C0_DW + 255) / 256, 1), 256, 0, s>>>(bands, dw, C0_BANDS);
    return ctcasr_launch_status();
}
