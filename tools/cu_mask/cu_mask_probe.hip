// Which XCDs / CUs does a CU-masked HIP stream run workgroups on?  (tools/cu_mask_probe.py)
#include <hip/hip_runtime.h>
#include <stdint.h>

__global__ void where_kernel(int *xcc, int *hwid, int spin) {
    if (threadIdx.x == 0) {
        unsigned x, h;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(h));
        xcc[blockIdx.x] = (int)(x & 0xF);
        hwid[blockIdx.x] = (int)h;
    }
    // hold the CU for a while so that every workgroup of the launch is resident at once
    long long t0 = clock64();
    while (clock64() - t0 < spin) {}
}

extern "C" void *probe_stream_with_mask(const uint32_t *mask, uint32_t words) {
    hipStream_t s = nullptr;
    if (hipExtStreamCreateWithCUMask(&s, words, mask) != hipSuccess) return nullptr;
    return (void *)s;
}

extern "C" int probe_where(void *stream, int blocks, int lds_bytes, int *xcc, int *hwid, int spin) {
    hipFuncSetAttribute(reinterpret_cast<const void *>(where_kernel),
                        hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
    where_kernel<<<blocks, 64, lds_bytes, (hipStream_t)stream>>>(xcc, hwid, spin);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}
