// Round 6 probe: synthetic co-runners for the 128 CUs a half-chip backward recurrence leaves free:
// which resource of a matrix kernel beside it costs the recurrence its ~1.2 us per time step?
//   burn_mfma   MFMAs on registers only (power, no memory)
//   burn_lds    LDS reads only
//   burn_hbm    streaming reads of a big buffer (HBM / fabric / L2), no MFMA
//   burn_valu   FMAs on registers only
// each: `blocks` workgroups of 256 threads with 100 KB of LDS (one per CU), for `ticks` of the 100 MHz clock.
//   hipcc --offload-arch=gfx950 -O3 -shared -fPIC tools/beside/beside_kernels.hip -o tools/beside/beside.so
#include <hip/hip_runtime.h>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ void __launch_bounds__(256) burn_mfma(unsigned long long ticks, float *sink) {
    extern __shared__ char smem[];
    f16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(threadIdx.x * 0.001f + i); b[i] = (_Float16)(i * 0.5f); }
    f32x4 acc[8];
    for (int k = 0; k < 8; ++k) acc[k] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const unsigned long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) {
#pragma unroll
        for (int r = 0; r < 16; ++r)
#pragma unroll
            for (int k = 0; k < 8; ++k) acc[k] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc[k], 0, 0, 0);
    }
    float s = 0.f;
    for (int k = 0; k < 8; ++k) s += acc[k][0];
    if (s == 12345.678f) sink[0] = s;
}
__global__ void __launch_bounds__(256) burn_valu(unsigned long long ticks, float *sink) {
    extern __shared__ char smem[];
    float v[16];
    for (int i = 0; i < 16; ++i) v[i] = threadIdx.x * 0.001f + i;
    const unsigned long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) {
#pragma unroll
        for (int r = 0; r < 32; ++r)
#pragma unroll
            for (int i = 0; i < 16; ++i) v[i] = fmaf(v[i], 1.0001f, 0.5f);
    }
    float s = 0.f;
    for (int i = 0; i < 16; ++i) s += v[i];
    if (s == 12345.678f) sink[0] = s;
}
__global__ void __launch_bounds__(256) burn_lds(unsigned long long ticks, float *sink) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float4 *l = reinterpret_cast<float4 *>(smem);
    for (int i = threadIdx.x; i < 4096; i += 256) l[i] = make_float4(i, 1.f, 2.f, 3.f);
    __syncthreads();
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    const unsigned long long t0 = wall_clock64();
    unsigned idx = threadIdx.x;
    while (wall_clock64() - t0 < ticks) {
#pragma unroll
        for (int r = 0; r < 32; ++r) {
            const float4 v = l[(idx + r * 256) & 4095];
            acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
        }
        idx += 64;
    }
    if (acc.x + acc.y + acc.z + acc.w == 12345.678f) sink[0] = acc.x;
}
__global__ void __launch_bounds__(256) burn_hbm(unsigned long long ticks, const float4 *buf, size_t n4,
                                                float *sink) {
    extern __shared__ char smem[];
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    const unsigned long long t0 = wall_clock64();
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * 256;
    while (wall_clock64() - t0 < ticks) {
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            const float4 v = buf[i];
            acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
            i += stride;
            if (i >= n4) i -= n4;
        }
    }
    if (acc.x + acc.y + acc.z + acc.w == 12345.678f) sink[0] = acc.x;
}
extern "C" int beside_launch(int kind, int blocks, int busy_us, const void *buf, size_t bytes, float *sink,
                             void *stream) {
    const unsigned long long ticks = (unsigned long long)busy_us * 100ull;
    const size_t lds = 100 * 1024;
    hipStream_t s = (hipStream_t)stream;
    static bool once = false;
    if (!once) {
        hipFuncSetAttribute((const void *)burn_mfma, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipFuncSetAttribute((const void *)burn_valu, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipFuncSetAttribute((const void *)burn_lds, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipFuncSetAttribute((const void *)burn_hbm, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        once = true;
    }
    if (kind == 0) burn_mfma<<<blocks, 256, lds, s>>>(ticks, sink);
    else if (kind == 1) burn_valu<<<blocks, 256, lds, s>>>(ticks, sink);
    else if (kind == 2) burn_lds<<<blocks, 256, lds, s>>>(ticks, sink);
    else burn_hbm<<<blocks, 256, lds, s>>>(ticks, reinterpret_cast<const float4 *>(buf), bytes / 16, sink);
    return (int)hipGetLastError();
}
