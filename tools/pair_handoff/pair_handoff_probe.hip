// Round 6 probe: what does a 1 KB hand-off between two workgroups cost, per hop, when both sit on
// the same XCD (blockIdx b and b ^ 8) or on different ones (b ^ 1), with the recurrence launch's
// footprint (128 workgroups of 4 waves, one per CU)?  This is the hop the K-pair (2-D) exchange
// of the fp16 backward recurrence adds to a tile's step.
//
// Protocol (the one the kernel would use): each wave hands ITS 64 partial sums (256 B) to the same
// wave of the partner - 64 4-byte stores, drained (vmcnt(0)), then one flag store by lane 0.  The
// receiver polls the flag and reads the 64 values:
//   READ = 0  scalar loads (s_load ... glc: lgkmcnt, NOT the in-order vmcnt queue - the receiver of
//             the real kernel has 39 vector loads of the next phase in flight at this point),
//             64 v_writelane to spread them over the lanes
//   READ = 1  vector loads, sc1 (L2 bypass: works across XCDs)
//   READ = 2  vector loads, sc0 only (L1 bypass, the XCD's own L2)
//   WRITE = 0 plain stores (the XCD's L2 is the point of coherence), 1 = sc1 write-through
// Every variant checks the values it reads (a wrong value = the flag overtook the data).
//
//   hipcc --offload-arch=gfx950 -O3 tools/pair_handoff/pair_handoff_probe.hip -o tools/pair_handoff/probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

typedef int i32x16 __attribute__((ext_vector_type(16)));

struct Slot {                    // per (workgroup, wave): what the partner's wave writes to
    unsigned data[64];
    unsigned flag;
    unsigned pad[63];
};

__device__ __forceinline__ unsigned long long wall() { return wall_clock64(); }

template <int READ, int WRITE>
__global__ void __launch_bounds__(256) pingpong(Slot *slots, int iters, int partner_xor,
                                                unsigned long long *ticks, unsigned *errors,
                                                unsigned *xcc) {
    extern __shared__ char smem[];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int me = blockIdx.x, other = me ^ partner_xor;
    const bool first = me < other;
    Slot *mine = slots + me * 4 + wave, *theirs = slots + other * 4 + wave;
    if (threadIdx.x == 0) {
        unsigned id;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(id));
        xcc[me] = id & 0xF;
    }
    unsigned bad = 0;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(slots, 0, 0x7FFFFFFF, 0x00020000);
    auto send = [&](unsigned tag) {
        const unsigned off = (unsigned)((char *)&theirs->data[lane] - (char *)slots);
        __builtin_amdgcn_raw_buffer_store_b32(tag * 64u + lane, rs, (int)off, 0, WRITE ? 16 : 0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (lane == 0) {
            const unsigned foff = (unsigned)((char *)&theirs->flag - (char *)slots);
            __builtin_amdgcn_raw_buffer_store_b32(tag, rs, (int)foff, 0, WRITE ? 16 : 0);
        }
    };
    auto recv = [&](unsigned tag) {
        unsigned spins = 0;
        if constexpr (READ == 0) {
            const unsigned *fp = &mine->flag;
            for (;;) {
                unsigned f;
                asm volatile("s_load_dword %0, %1, 0x0 glc\n\ts_waitcnt lgkmcnt(0)" : "=s"(f) : "s"(fp) : "memory");
                if (f == tag) break;
                if (++spins > (1u << 14)) { bad |= 0x80000000u; break; }
                __builtin_amdgcn_s_sleep(1);
            }
            const unsigned *dp = mine->data;
            i32x16 a, b, c, d;
            asm volatile("s_load_dwordx16 %0, %4, 0x0 glc\n\ts_load_dwordx16 %1, %4, 0x40 glc\n\t"
                         "s_load_dwordx16 %2, %4, 0x80 glc\n\ts_load_dwordx16 %3, %4, 0xc0 glc\n\t"
                         "s_waitcnt lgkmcnt(0)"
                         : "=&s"(a), "=&s"(b), "=&s"(c), "=&s"(d) : "s"(dp) : "memory");
            unsigned v = 0;
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                asm("v_writelane_b32 %0, %1, %2" : "+v"(v) : "s"(a[i]), "n"(i));
                asm("v_writelane_b32 %0, %1, %2" : "+v"(v) : "s"(b[i]), "n"(i + 16));
                asm("v_writelane_b32 %0, %1, %2" : "+v"(v) : "s"(c[i]), "n"(i + 32));
                asm("v_writelane_b32 %0, %1, %2" : "+v"(v) : "s"(d[i]), "n"(i + 48));
            }
            if (v != tag * 64u + lane) bad += 1;
        } else {
            constexpr int AUX = READ == 1 ? 16 : 1;
            const unsigned foff = (unsigned)((char *)&mine->flag - (char *)slots);
            for (;;) {
                const unsigned f = (unsigned)__builtin_amdgcn_raw_buffer_load_b32(rs, 0, (int)foff, AUX);
                if (__builtin_amdgcn_readfirstlane(f) == tag) break;
                if (++spins > (1u << 14)) { bad |= 0x80000000u; break; }
                __builtin_amdgcn_s_sleep(1);
            }
            const unsigned off = (unsigned)((char *)&mine->data[lane] - (char *)slots);
            const unsigned v = (unsigned)__builtin_amdgcn_raw_buffer_load_b32(rs, (int)off, 0, AUX);
            if (v != tag * 64u + lane) bad += 1;
        }
    };
    const unsigned long long t0 = wall();
    for (int i = 1; i <= iters; ++i) {
        if (first) { send(i); recv(i); } else { recv(i); send(i); }
        if (bad & 0x80000000u) break;       // a time-out: the partner will see one too
    }
    const unsigned long long t1 = wall();
    if (lane == 0) ticks[me * 4 + wave] = t1 - t0;
    if (bad) atomicAdd(errors, bad & 0x80000000u ? 1u << 16 : 1u);
}

template <int READ, int WRITE>
static void run(const char *name, int partner_xor, int iters) {
    const int blocks = 128;
    Slot *slots; unsigned long long *ticks; unsigned *errors, *xcc;
    hipMalloc(&slots, sizeof(Slot) * blocks * 4);
    hipMalloc(&ticks, 8 * blocks * 4);
    hipMalloc(&errors, 4);
    hipMalloc(&xcc, 4 * blocks);
    hipMemset(slots, 0, sizeof(Slot) * blocks * 4);
    hipMemset(errors, 0, 4);
    hipFuncSetAttribute(reinterpret_cast<const void *>(pingpong<READ, WRITE>),
                        hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
    pingpong<READ, WRITE><<<blocks, 256, 100 * 1024>>>(slots, iters, partner_xor, ticks, errors, xcc);
    if (hipDeviceSynchronize() != hipSuccess) { printf("%s: launch failed\n", name); exit(1); }
    std::vector<unsigned long long> h(blocks * 4);
    std::vector<unsigned> hx(blocks);
    unsigned herr = 0;
    hipMemcpy(h.data(), ticks, 8 * blocks * 4, hipMemcpyDeviceToHost);
    hipMemcpy(hx.data(), xcc, 4 * blocks, hipMemcpyDeviceToHost);
    hipMemcpy(&herr, errors, 4, hipMemcpyDeviceToHost);
    double sum = 0, mx = 0;
    for (auto v : h) { sum += (double)v; if ((double)v > mx) mx = (double)v; }
    int same = 0;
    for (int b = 0; b < blocks; ++b) same += hx[b] == hx[b ^ partner_xor];
    // wall_clock64: 100 MHz
    printf("%-44s partner b^%d (same XCD: %3d/128)  hop mean %.3f us  slowest wave %.3f us  wrong values %u  time-outs %u\n",
           name, partner_xor, same, sum / h.size() / 100.0 / (2.0 * iters), mx / 100.0 / (2.0 * iters),
           herr & 0xFFFFu, herr >> 16);
    hipFree(slots); hipFree(ticks); hipFree(errors); hipFree(xcc);
}

int main() {
    const int iters = 2000;
    run<0, 0>("s_load glc, plain stores", 8, iters);
    run<0, 1>("s_load glc, sc1 stores", 8, iters);
    run<2, 0>("vector sc0 loads, plain stores", 8, iters);
    run<2, 1>("vector sc0 loads, sc1 stores", 8, iters);
    run<1, 1>("vector sc1 loads, sc1 stores", 8, iters);
    run<1, 1>("vector sc1 loads, sc1 stores", 1, iters);
    run<0, 1>("s_load glc, sc1 stores", 1, iters);
    run<2, 1>("vector sc0 loads, sc1 stores", 1, iters);
    return 0;
}
