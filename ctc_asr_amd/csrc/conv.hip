// Data gradient of the DS2 stack's 11 x 21, stride (1, 2), 32 -> 32 channel convolution (the second
// conv layer, asr/util/tf_contrib.py:64-146) as an implicit GEMM on the fp32 MFMA units.
//
// MIOpen's best kernel for this layer's backward-data runs at ~60 TFLOP/s and needs a zero-fill of
// its padded output plus a strided copy of the interior afterwards (1.6 ms of the 25 ms C2 step).
// Here the padding never exists: out-of-range taps read zeros from the LDS patch and the result is
// written straight into the unpadded NHWC tensor the layer below consumes.
//
//   dx[b, t, f, ci] = sum_{kt, kf, co} dz[b, t + 5 - kt, (f + 9 - kf) / 2, co] * w[co, ci, kt, kf]
//                     (terms with f + 9 - kf odd or indices out of range vanish)
//
// One workgroup = one utterance x 16 output frames x all 40 frequencies x all 32 channels.  Its
// slice of dz (26 frames x 20 frequencies x 32 channels, zero border of 5 frequencies each side)
// sits in LDS with a row pitch of 36 floats (conflict-free 16-byte fragment reads).  Output rows
// of equal frequency parity see the same set of kf taps, so the M tiles are built per parity:
// 4 frames x 20 positions = 80 rows = 5 tiles; a wave owns 4 frames = 10 M tiles x 2 N tiles.
// Per tap the B fragments (weights, pre-packed in fragment order, L2-resident) are shared by a
// wave's 5 tiles; the A fragment of a row is the patch shifted by a tap-uniform offset.
#include "common.h"

namespace {

constexpr int CV_C = 32;           // channels in and out
constexpr int CV_FO = 20;          // frequencies of dz (conv output)
constexpr int CV_FI = 40;          // frequencies of dx (conv input)
constexpr int CV_KT = 11, CV_KF = 21;
constexpr int CV_TT = 16;          // output frames per workgroup
constexpr int CV_PF = CV_FO + 10;  // patch frequencies: 5 zero columns each side
constexpr int CV_PITCH = 36;       // floats per (frame, frequency) cell of the patch
constexpr int CV_PT = CV_TT + CV_KT - 1;

typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void mma4(f32x4 &acc, const float4 &a, const float4 &b) {
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, b.x, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, b.y, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, b.z, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, b.w, acc, 0, 0, 0);
}

// The four B fragments of a tap: K chunks q = 0, 1 x N tiles 0, 1 (packed order, see below).
struct BFrag { float4 b00, b01, b10, b11; };
__device__ __forceinline__ BFrag load_b(const float4 *__restrict__ wp, int kt, int kf, int kg,
                                        int n) {
    const float4 *wt = wp + (size_t)((kt * CV_KF + kf) * 8 + kg) * 32 + n;
    return BFrag{wt[0], wt[16], wt[4 * 32], wt[4 * 32 + 16]};
}

// w [Cout, Cin, kt, kf] (the arena's compute layout) -> packed[kt][kf][q][kg][ci][r] with
// co = 16 q + 4 kg + r: lane (ci & 15, kg) of N tile ci / 16 reads one float4.
__global__ void conv_pack_bwd_kernel(const float *__restrict__ w, float *__restrict__ packed) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= CV_KT * CV_KF * CV_C * CV_C) return;
    const int r = i & 3, ci = (i >> 2) & 31, kg = (i >> 7) & 3, q = (i >> 9) & 1;
    const int tap = i >> 10, kf = tap % CV_KF, kt = tap / CV_KF;
    const int co = 16 * q + 4 * kg + r;
    packed[i] = w[((co * CV_C + ci) * CV_KT + kt) * CV_KF + kf];
}

// Forward pass of the same layer, same machinery:
//   y[b, t, fo, co] = bias[co] + sum_{kt, kf, ci} x[b, t + kt - 5, 2 fo + kf - 9, ci] * w[co, ci, kt, kf]
// In padded coordinates fp = 2 fo + kf, so a tap of parity par = kf & 1 only ever reads input
// frequencies of that parity: the patch is staged twice, once per parity plane
// (26 frames x 30 positions x 32 channels), position = fo + kf / 2 - which makes the A fragment
// of a row, again, the lane's base address plus a tap-uniform offset.
// packed (forward): [kt][kf][q][kg][co][r] with ci = 16 q + 4 kg + r.
__global__ void conv_pack_fwd_kernel(const float *__restrict__ w, float *__restrict__ packed) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= CV_KT * CV_KF * CV_C * CV_C) return;
    const int r = i & 3, co = (i >> 2) & 31, kg = (i >> 7) & 3, q = (i >> 9) & 1;
    const int tap = i >> 10, kf = tap % CV_KF, kt = tap / CV_KF;
    const int ci = 16 * q + 4 * kg + r;
    packed[i] = w[((co * CV_C + ci) * CV_KT + kt) * CV_KF + kf];
}

__global__ void __launch_bounds__(256)
conv_fwd_kernel(const float *__restrict__ x, const float4 *__restrict__ wp,
                const float *__restrict__ bias, float *__restrict__ y, int T) {
    extern __shared__ __attribute__((aligned(16))) float patch[];   // [CV_PT][CV_PF][CV_PITCH]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int t0 = blockIdx.x * CV_TT, b = blockIdx.y;
    const int kg = lane >> 4, n = lane & 15;
    float4 *patch4 = reinterpret_cast<float4 *>(patch);

    int base_a[5];
#pragma unroll
    for (int ti = 0; ti < 5; ++ti) {
        const int row = ti * 16 + n, tt = row / CV_FO, fo = row % CV_FO;
        base_a[ti] = ((4 * wave + tt) * CV_PF + fo) * CV_PITCH + 4 * kg;
    }
    f32x4 acc[5][2];
#pragma unroll
    for (int ti = 0; ti < 5; ++ti)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) acc[ti][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};

#pragma unroll
    for (int par = 0; par < 2; ++par) {    // unrolled: compile-time tap counts in both copies
        if (par) __syncthreads();          // everyone is done reading the other plane
        for (int i = tid; i < CV_PT * CV_PF * 8; i += 256) {
            const int c4 = i & 7, pos = (i >> 3) % CV_PF, pr = i / (8 * CV_PF);
            const int ts = t0 - 5 + pr, fi = 2 * pos + par - 9;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (ts >= 0 && ts < T && fi >= 0 && fi < CV_FI)
                v = reinterpret_cast<const float4 *>(x)[((size_t)(b * T + ts) * CV_FI + fi) * 8 + c4];
            patch4[((pr * CV_PF + pos) * CV_PITCH) / 4 + c4] = v;
        }
        __syncthreads();
        const int taps = par == 0 ? 11 : 10;
        // B fragments (weights, from L2) run one tap ahead of the MFMAs that use them; the
        // scheduling barriers keep hipcc from sinking the loads next to their first use
        BFrag cur = load_b(wp, 0, par, kg, n);
        for (int kt = 0; kt < CV_KT; ++kt) {
#pragma unroll
            for (int m = 0; m < taps; ++m) {
                const bool wrap = m + 1 == taps;
                const BFrag nxt = load_b(wp, wrap ? min(kt + 1, CV_KT - 1) : kt,
                                         wrap ? par : 2 * (m + 1) + par, kg, n);
                __builtin_amdgcn_sched_barrier(0);
                const int tap_off = (kt * CV_PF + m) * CV_PITCH;
#pragma unroll
                for (int ti = 0; ti < 5; ++ti) {
                    const float4 a0 = *reinterpret_cast<const float4 *>(patch + base_a[ti] + tap_off);
                    const float4 a1 =
                        *reinterpret_cast<const float4 *>(patch + base_a[ti] + tap_off + 16);
                    mma4(acc[ti][0], a0, cur.b00);
                    mma4(acc[ti][1], a0, cur.b01);
                    mma4(acc[ti][0], a1, cur.b10);
                    mma4(acc[ti][1], a1, cur.b11);
                }
                __builtin_amdgcn_sched_barrier(0);
                cur = nxt;
            }
        }
    }

    const float bias0 = bias ? bias[n] : 0.f, bias1 = bias ? bias[16 + n] : 0.f;
#pragma unroll
    for (int ti = 0; ti < 5; ++ti)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = ti * 16 + 4 * kg + r, tt = row / CV_FO, fo = row % CV_FO;
            const int t = t0 + 4 * wave + tt;
            if (t < T) {
                float *out = y + ((size_t)(b * T + t) * CV_FO + fo) * CV_C + n;
                out[0] = acc[ti][0][r] + bias0;
                out[16] = acc[ti][1][r] + bias1;
            }
        }
}

__global__ void __launch_bounds__(256)
conv_bwd_data_kernel(const float *__restrict__ dz, const float4 *__restrict__ wp,
                     float *__restrict__ dx, int T) {
    extern __shared__ __attribute__((aligned(16))) float patch[];   // [CV_PT][CV_PF][CV_PITCH]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int t0 = blockIdx.x * CV_TT, b = blockIdx.y;

    // ---- stage dz[b, t0-5 .. t0+20, :, :] with its zero border ------------------------------
    float4 *patch4 = reinterpret_cast<float4 *>(patch);
    for (int i = tid; i < CV_PT * CV_PF * (CV_PITCH / 4); i += 256)
        patch4[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    __syncthreads();
    for (int i = tid; i < CV_PT * CV_FO * (CV_C / 4); i += 256) {
        const int c4 = i & 7, fo = (i >> 3) % CV_FO, pr = i / (8 * CV_FO);
        const int ts = t0 - 5 + pr;
        if (ts >= 0 && ts < T)
            patch4[((pr * CV_PF + fo + 5) * CV_PITCH) / 4 + c4] =
                reinterpret_cast<const float4 *>(dz)[((size_t)(b * T + ts) * CV_FO + fo) * 8 + c4];
    }
    __syncthreads();

    // ---- per-lane fragment addresses -----------------------------------------------------------
    const int kg = lane >> 4, n = lane & 15;
    int base_a[5];                       // float index of (frame, position) of this lane's row
#pragma unroll
    for (int ti = 0; ti < 5; ++ti) {
        const int row = ti * 16 + n, tt = row / CV_FO, j = row % CV_FO;
        base_a[ti] = ((4 * wave + tt) * CV_PF + j) * CV_PITCH + 4 * kg;
    }
    f32x4 acc[2][5][2];
#pragma unroll
    for (int p = 0; p < 2; ++p)
#pragma unroll
        for (int ti = 0; ti < 5; ++ti)
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) acc[p][ti][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // ---- taps -----------------------------------------------------------------------------------
    for (int kt = 0; kt < CV_KT; ++kt) {
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            // even output frequencies (f = 2j) meet the odd kf, odd ones (f = 2j+1) the even kf;
            // dz frequency = j + 4 + p - m with m = kf / 2
            const int taps = p == 0 ? 10 : 11;
            // (no explicit weight prefetch here: hipcc's own schedule of this loop nest reaches
            // 138 TFLOP/s, the pinned one-tap-ahead variant of the forward kernel 131)
            for (int m = 0; m < taps; ++m) {
                const BFrag cur = load_b(wp, kt, p == 0 ? 2 * m + 1 : 2 * m, kg, n);
                const int tap_off = ((10 - kt) * CV_PF + 9 + p - m) * CV_PITCH;
#pragma unroll
                for (int ti = 0; ti < 5; ++ti) {
                    const float4 a0 = *reinterpret_cast<const float4 *>(patch + base_a[ti] + tap_off);
                    const float4 a1 =
                        *reinterpret_cast<const float4 *>(patch + base_a[ti] + tap_off + 16);
                    mma4(acc[p][ti][0], a0, cur.b00);
                    mma4(acc[p][ti][1], a0, cur.b01);
                    mma4(acc[p][ti][0], a1, cur.b10);
                    mma4(acc[p][ti][1], a1, cur.b11);
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
                const int row = ti * 16 + 4 * kg + r, tt = row / CV_FO, j = row % CV_FO;
                const int t = t0 + 4 * wave + tt;
                if (t < T) {
                    float *out = dx + ((size_t)(b * T + t) * CV_FI + 2 * j + p) * CV_C + n;
                    out[0] = acc[p][ti][0][r];
                    out[16] = acc[p][ti][1][r];
                }
            }
}

}  // namespace

// Fragment-ordered copies of the layer's kernel for ctcasr_conv_s12_bwd_data and _fwd (2 x 946 KB;
// the weights change every step).  w: [32, 32, 11, 21] = [Cout, Cin, kt, kf].
extern "C" int ctcasr_conv_s12_pack_weights(const float *w, float *packed, ctcasr_stream_t stream) {
    if (!w || !packed) return CTCASR_ERR_BAD_ARGUMENT;
    const int n = CV_KT * CV_KF * CV_C * CV_C;
    conv_pack_bwd_kernel<<<(n + 255) / 256, 256, 0, (hipStream_t)stream>>>(w, packed);
    conv_pack_fwd_kernel<<<(n + 255) / 256, 256, 0, (hipStream_t)stream>>>(w, packed + n);
    return ctcasr_launch_status();
}

// x [B, T, 40, 32] (NHWC) -> y [B, T, 20, 32] = conv(x) + bias (bias may be NULL).  `packed` holds
// 2 x 11*21*32*32 floats: ctcasr_conv_s12_pack_weights fills the backward order first, then the
// forward order.
extern "C" int ctcasr_conv_s12_fwd(const float *x, const float *packed, const float *bias, float *y,
                                   int B, int T, ctcasr_stream_t stream) {
    if (!x || !packed || !y || B <= 0 || T <= 0) return CTCASR_ERR_BAD_ARGUMENT;
    if (B > 65535) return CTCASR_ERR_UNSUPPORTED;
    const size_t lds = (size_t)CV_PT * CV_PF * CV_PITCH * sizeof(float);
    if (hipFuncSetAttribute(reinterpret_cast<const void *>(&conv_fwd_kernel),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
        return CTCASR_ERR_LAUNCH;
    dim3 grid((T + CV_TT - 1) / CV_TT, B);
    conv_fwd_kernel<<<grid, 256, lds, (hipStream_t)stream>>>(
        x, reinterpret_cast<const float4 *>(packed + CV_KT * CV_KF * CV_C * CV_C), bias, y, T);
    return ctcasr_launch_status();
}

// dz [B, T, 20, 32] (NHWC, gradient w.r.t. the layer's pre-activation output) -> dx [B, T, 40, 32].
extern "C" int ctcasr_conv_s12_bwd_data(const float *dz, const float *packed, float *dx, int B,
                                        int T, ctcasr_stream_t stream) {
    if (!dz || !packed || !dx || B <= 0 || T <= 0) return CTCASR_ERR_BAD_ARGUMENT;
    if (B > 65535) return CTCASR_ERR_UNSUPPORTED;
    const size_t lds = (size_t)CV_PT * CV_PF * CV_PITCH * sizeof(float);
    if (hipFuncSetAttribute(reinterpret_cast<const void *>(&conv_bwd_data_kernel),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
        return CTCASR_ERR_LAUNCH;
    dim3 grid((T + CV_TT - 1) / CV_TT, B);
    conv_bwd_data_kernel<<<grid, 256, lds, (hipStream_t)stream>>>(
        dz, reinterpret_cast<const float4 *>(packed), dx, T);
    return ctcasr_launch_status();
}
