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
__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }
