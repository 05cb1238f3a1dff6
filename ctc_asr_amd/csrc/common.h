// Shared helpers of the gfx950 kernels behind include/ctcasr.h.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <math.h>

#include "../../include/ctcasr.h"

#define CTCASR_WAVE 64

typedef float f32x4 __attribute__((ext_vector_type(4)));

static inline int ctcasr_launch_status() {
    return hipGetLastError() == hipSuccess ? CTCASR_OK : CTCASR_ERR_LAUNCH;
}

static inline size_t ctcasr_align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off, 64));
    return v;
}
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}
// Gate non-linearities on the hardware transcendentals (v_exp_f32 / v_rcp_f32, ~1 ulp each) and
// branch-free: the library expf / tanhf (range reduction, two divergent tanh paths, IEEE division)
// cost ~250 VALU instructions per LSTM cell update inside every time step of the recurrence.
// Absolute error ~1e-7, far inside the 1e-3 parity bar.  (Measured: the gate phase of the
// persistent forward step drops from 0.59 to 0.49 us; the step itself stays barrier-bound.)
__device__ __forceinline__ float sigmoidf_(float x) {
    return __frcp_rn(1.0f + __expf(-x));
}
__device__ __forceinline__ float tanhf_(float x) {
    const float ax = fabsf(x);
    const float e = __expf(-2.0f * ax);                          // in (0, 1]
    const float big = (1.0f - e) * __frcp_rn(1.0f + e);          // fine once 1 - e is not tiny
    const float x2 = ax * ax;                                    // odd series below 0.05
    const float small = ax * (1.0f + x2 * (-0.33333334f + x2 * (0.13333334f - x2 * 0.053968254f)));
    return copysignf(ax < 0.05f ? small : big, x);
}
