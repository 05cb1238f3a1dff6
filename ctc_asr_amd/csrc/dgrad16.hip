// Data gradient of a recurrent layer's input projection, dx = dxw . W_ih, straight from what the
// fp16-pipe backward recurrence has published (reference: the gradient of the cuDNN input
// projection, asr/model.py:194-215).
//
// prnn_bwd16_kernel (rnn_persistent.hip) leaves every step's dgates - for the LSTM these ARE dxw -
// in its exchange buffer as two fp16 pieces per value, scaled per (row, producer = 64 gate columns)
// by a power of two, in the register layout of the A operand of v_mfma_f32_16x16x32_f16:
//     [step][dir][producer P][half m][piece][k group q][b][8 halves]      (16-byte granules)
// followed by the inverse scales [step][dir][producer][32 rows].  That is a block-scaled GEMM
// operand as it stands: this kernel multiplies it into the fp16 pieces of W_ih (fixed scale, packed
// once per step in the same K order and in B-fragment order) with a FRESH accumulator per K = 32
// stage that enters the fp32 total times the row's inverse scale - the recurrence kernel's own
// arithmetic.  No pass over the fp32 dxw, no row split, no library kernel: both operands go from
// HBM / L2 to LDS by LDS-DMA (global_load_lds_dwordx4: a 1 KB chunk = one MFMA fragment of 64
// lanes, lane-linear in LDS = conflict-free ds_read_b128), and nothing in it waits for another
// workgroup, so it may share the chip with whatever the side stream still runs.
//
// Workgroup = 512 threads (8 waves, 2 x 4), tile 256 rows x 256 columns, wave tile 128 x 64 = 8 x 4
// MFMA tiles (128 accumulator registers).  A "row unit" = 16 batch rows of one time step; a tile
// = 16 units (B = 32: 8 time steps, B <= 16: 16).  K stage = (dir, P, m): 32 gate columns; per
// stage 32 KB of A + 32 KB of B + 1 KB of inverse scales, double buffered (130 KB of LDS).
#include "common.h"

#include <type_traits>

namespace {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
union DgFrag {
    u32x4 u;
    f16x8 h;
};

constexpr int DG_H = 1024, DG_GH = 4 * DG_H;
constexpr int DG_THREADS = 512;
constexpr int DG_UNITS = 16;                    // row units (16 rows each) per tile
constexpr int DG_BN = 256;                      // columns per tile
constexpr int DG_A_BYTES = DG_UNITS * 2 * 1024, DG_B_BYTES = (DG_BN / 16) * 2 * 1024;
constexpr int DG_STAGE_BYTES = DG_A_BYTES + DG_B_BYTES + 1024;
constexpr size_t DG_LDS_BYTES = 2 * (size_t)DG_STAGE_BYTES;
constexpr int DG_SCALE_ROWS = 32;               // PRNN_B16_SCALE_ROWS
constexpr int DG_STAGES_PER_DIR = DG_H / 16 * 2;

struct DgArgs {
    const char *xchg;       // exchange buffer: the all-zero block, then one block per step
    const float *scales;    // inverse scales [step][dir][producer][32]
    const char *wpk;        // packed weights [stage][column tile][piece][lane][16 B]
    float *out;             // [T * B, N]
    int64_t ldc;
    int T, B, N, nt_total;
    int t_lo, t_hi;         // rows: times [t_lo, t_hi)
    int ks_lo, ks_hi;       // stages: dir * 128 + P * 2 + m
    int accumulate;
    float out_scale;
    int tiles_m, tiles_n, units_per_step;
};

// One LDS-DMA of 64 x 16 bytes: lane l's 16 bytes from its own global address to LDS byte
// lds_base + 16 l.  Inline assembly on purpose: with the builtin hipcc (ROCm 7.2) tracks the
// outstanding DMA itself and waits vmcnt(0) in front of the first LDS read it cannot prove
// disjoint - in the middle of the stage that is supposed to hide the transfer.  Here the waits are
// explicit (`dma_wait` in front of the stage barrier).
// (address = 64-bit uniform base in SGPRs + the lane's own 32-bit offset: per DMA one scalar add,
// no vector address arithmetic)
__device__ __forceinline__ void dma16(const char *base, unsigned lane_off, unsigned lds_base) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\t"
                 "global_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(lane_off), "s"(base), "s"(lds_base)
                 : "memory");
}
__device__ __forceinline__ void dma_wait() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

// tuning switches of the stage loop (reported by ctcasr_build_flags when not at their defaults)
#ifndef DG_SCHED
#define DG_SCHED 0              // pin the MFMA / VALU / LDS issue order of a row unit (slower)
#endif
#ifndef DG_YOUNG_PRIO
#define DG_YOUNG_PRIO 0         // s_setprio of waves 4 - 7 (0: none)
#endif
#ifndef DG_LOADERS
#define DG_LOADERS 4            // waves that issue the DMAs (8: every wave its eighth)
#endif
#ifndef DG_PROBE
#define DG_PROBE 0              // timing probes, results wrong on purpose (ctcasr_build_flags):
#endif                          // 1 every stage re-reads stage 0's bytes, 2 no MFMAs, 3 no DMA,
                                // 4 phase clocks of workgroup 0 into the first floats of dx
#ifndef DG_FOLD_VALU
#define DG_FOLD_VALU 2          // VALU instructions of the fold in the shadow of one MFMA
#endif
// total + fresh * inverse scale.  (This file is built with -fno-slp-vectorize, build.py: left alone
// hipcc packs neighbouring folds into v_pk_fma_f32, which beside MFMAs costs more than the two
// v_fma_f32 it replaces - MI355X_MICROARCH.md, fillers.  Not inline assembly: the hazard
// recogniser does not see an MFMA result read by an asm VALU instruction and omits the wait
// states - stale accumulators.)
__device__ __forceinline__ float fold1(float fresh, float inv, float total) {
    return fmaf(fresh, inv, total);
}

__global__ void __launch_bounds__(DG_THREADS) dgrad16_bs_kernel(DgArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;

    // XCD-aware tile order (as split_gemm.hip): consecutive blockIdx go round the 8 XCDs; each XCD
    // walks a contiguous range of tiles, four tile rows per tile column - the 32 workgroups an XCD
    // runs at a time share 4 A panels and all B panels in its L2
    const int tiles = p.tiles_m * p.tiles_n;
    const int per_xcd = (tiles + 7) / 8;
    const int v = (int)(blockIdx.x & 7) * per_xcd + (int)(blockIdx.x >> 3);
    if (v >= tiles) return;
    constexpr int GROUP = 4;
    const int group = v / (GROUP * p.tiles_n), within = v - group * GROUP * p.tiles_n;
    const int rows_here = min(GROUP, p.tiles_m - group * GROUP);
    const int tm = group * GROUP + within % rows_here, tn = within / rows_here;

    const int B = p.B, ups = p.units_per_step;
    const int t0 = p.t_lo + tm * (DG_UNITS / ups);
    const int nt0 = tn * (DG_BN / 16);
    const size_t x_step = (size_t)2 * B * DG_GH * sizeof(float);

    // ---- DMA addressing: uniform 64-bit bases (scalar arithmetic per DMA) + one 32-bit offset per
    // lane.  A chunk (unit u, piece): lane l = k group l >> 4, row l & 15 of the unit's 16 rows -
    // the lane's part depends on the unit's half h only; B chunk: lane-linear
    unsigned a_lane[2];
#pragma unroll
    for (int h = 0; h < 2; ++h)
        a_lane[h] = (unsigned)(((lane >> 4) * B * 4 + min(h * 16 + (lane & 15), B - 1) * 4) *
                               sizeof(float));
    const unsigned b_lane = lane * 16;
    // inverse scales: lane l = unit l >> 2, rows 4 (l & 3) .. + 3
    unsigned s_off[2];
    {
        const int u = lane >> 2;
        const int t = min(t0 + u / ups, p.t_hi - 1), h = u % ups;
#pragma unroll
        for (int d = 0; d < 2; ++d) {
            const int s = d == 0 ? t : p.T - 1 - t;
            s_off[d] = (unsigned)((((size_t)s * 2 + d) * (DG_H / 16) * DG_SCALE_ROWS + h * 16 +
                                   4 * (lane & 3)) * sizeof(float));
        }
    }
    const unsigned lds0 = __builtin_amdgcn_readfirstlane(
        (unsigned)(uintptr_t)(__attribute__((address_space(3))) char *)smem);
    // The 65 DMAs of a stage take the CU's one vector-memory issue path ~16 clocks each (1 KB at
    // 64 B per clock), and a wave sits in that queue while its DMAs wait.  The first DG_LOADERS
    // waves issue them all: they are the older half of the workgroup, win every arbitration and
    // would otherwise wait ~1700 clocks at the stage barrier for the younger half.
    constexpr int PER = 32 / DG_LOADERS;        // chunks of A and of B per loading wave
    // (what does not change from stage to stage, as 32-bit offsets: scalar registers)
    unsigned a_unit[PER / 2][2], b_tile[PER / 2];
#pragma unroll
    for (int c = 0; c < PER / 2; ++c) {
        const int u = (wave % DG_LOADERS) * (PER / 2) + c;
        const int t = min(t0 + u / ups, p.t_hi - 1);
#pragma unroll
        for (int d = 0; d < 2; ++d)
            a_unit[c][d] = __builtin_amdgcn_readfirstlane((unsigned)(
                (size_t)(1 + (d == 0 ? t : p.T - 1 - t)) * x_step + (size_t)d * B * DG_GH * sizeof(float)));
        b_tile[c] = __builtin_amdgcn_readfirstlane(
            (unsigned)(min(nt0 + u, p.nt_total - 1) * 2048));
    }
    // quarter q (0 .. 3) of a loading wave's DMAs for stage ks
    auto issue = [&](int ks, unsigned buf, int q) {
        if (DG_PROBE == 1) ks = 0;
        if ((DG_PROBE == 3 && ks != p.ks_lo) || wave >= DG_LOADERS) return;
        const int d = ks / DG_STAGES_PER_DIR, pm = ks % DG_STAGES_PER_DIR;
        const char *a_base = p.xchg + (size_t)pm * 2 * B * 64;
        const char *b_base = p.wpk + (size_t)ks * p.nt_total * 2048;
#pragma unroll
        for (int c = q * (PER / 4); c < (q + 1) * (PER / 4); ++c) {
            const int chunk = wave * PER + c, u = chunk >> 1, pc = chunk & 1;
            dma16(b_base + b_tile[c >> 1] + (c & 1) * 1024, b_lane,
                  buf + DG_A_BYTES + (wave * PER + c) * 1024);
            dma16(a_base + (d ? a_unit[c >> 1][1] : a_unit[c >> 1][0]) + (size_t)pc * B * 64,
                  (u % ups) ? a_lane[1] : a_lane[0], buf + chunk * 1024);
        }
        if (q == 0 && wave == DG_LOADERS - 1)
            dma16(reinterpret_cast<const char *>(p.scales + (size_t)(pm >> 1) * DG_SCALE_ROWS),
                  d ? s_off[1] : s_off[0], buf + DG_A_BYTES + DG_B_BYTES);
    };

    f32x4 total[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) total[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};

#if DG_YOUNG_PRIO
    // (the younger half of the workgroup loses every arbitration to the older one, which issues the
    // DMAs and still finishes its stage ~800 clocks earlier: a static priority for the second half)
    if (wave >= 4) __builtin_amdgcn_s_setprio(DG_YOUNG_PRIO);
#endif
    for (int q = 0; q < 4; ++q) issue(p.ks_lo, lds0, q);
    unsigned long long pt[4] = {0, 0, 0, 0}, pc0 = 0;
    for (int ks = p.ks_lo; ks < p.ks_hi; ++ks) {
        const int par = (ks - p.ks_lo) & 1;
        const char *cur = smem + par * DG_STAGE_BYTES;
        // this stage has landed (every wave waits for its own chunks, then the barrier) and
        // everybody is done reading the other buffer: refill it, then multiply
        if (DG_PROBE == 4) {
            __builtin_amdgcn_sched_barrier(0);
            const unsigned long long c = __builtin_readcyclecounter();
            if (ks > p.ks_lo) pt[3] += c - pc0;
            pc0 = c;
        }
        dma_wait();
        if (DG_PROBE == 4) {
            const unsigned long long c = __builtin_readcyclecounter();
            pt[0] += c - pc0; pc0 = c;
        }
        __syncthreads();
        if (DG_PROBE == 4) {
            const unsigned long long c = __builtin_readcyclecounter();
            pt[1] += c - pc0; pc0 = c;
        }
        const unsigned nxt = lds0 + (par ^ 1) * DG_STAGE_BYTES;
        const bool more = ks + 1 < p.ks_hi;
        if (more)
            for (int q = 0; q < 4; ++q) issue(ks + 1, nxt, q);
        if (DG_PROBE == 4) {
            __builtin_amdgcn_sched_barrier(0);
            const unsigned long long c = __builtin_readcyclecounter();
            pt[2] += c - pc0; pc0 = c;
        }

        // (one address register per operand: everything that is not a compile-time constant - the
        // buffer's parity, the wave's place in the tile, the lane - sits in the base, the reads
        // carry immediate offsets)
        const char *a_rd = cur + wr * (8 * 2048) + lane * 16;
        const char *b_rd = cur + DG_A_BYTES + wc * (4 * 2048) + lane * 16;
        const char *s_rd = cur + DG_A_BYTES + DG_B_BYTES + wr * (8 * 64) + (lane >> 4) * 16;
        DgFrag w1[4], w2[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            w1[j].u = *reinterpret_cast<const u32x4 *>(b_rd + j * 2048);
            w2[j].u = *reinterpret_cast<const u32x4 *>(b_rd + j * 2048 + 1024);
        }
        // Software pipeline over the wave's 8 row units: the fragments and inverse scales of unit
        // i + 1 are read and unit i - 1's fresh accumulators are folded into the totals while
        // unit i's 12 MFMAs issue (two sets of fresh accumulators).
        DgFrag d1[2], d2[2];
        float4 iv[3];               // (unit i - 1's are still needed when unit i + 1's arrive)
        f32x4 f[2][4];
        auto fetch = [&](int i, int slot) {
            d1[slot].u = *reinterpret_cast<const u32x4 *>(a_rd + i * 2048);
            d2[slot].u = *reinterpret_cast<const u32x4 *>(a_rd + i * 2048 + 1024);
            iv[i % 3] = *reinterpret_cast<const float4 *>(s_rd + i * 64);
        };
        auto fold = [&](int i, int slot) {
            const float4 s4 = iv[i % 3];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                total[i][j][0] = fold1(f[slot][j][0], s4.x, total[i][j][0]);
                total[i][j][1] = fold1(f[slot][j][1], s4.y, total[i][j][1]);
                total[i][j][2] = fold1(f[slot][j][2], s4.z, total[i][j][2]);
                total[i][j][3] = fold1(f[slot][j][3], s4.w, total[i][j][3]);
            }
        };
        fetch(0, 0);
#pragma unroll
        for (int i = 0; i < (DG_PROBE == 2 ? 0 : 8); ++i) {
            const int sl = i & 1;
            if (i + 1 < 8) fetch(i + 1, sl ^ 1);
#pragma unroll
            for (int j = 0; j < 4; ++j)
                f[sl][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(d1[sl].h, w1[j].h, zero, 0, 0, 0);
#pragma unroll
            for (int j = 0; j < 4; ++j)
                f[sl][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(d1[sl].h, w2[j].h, f[sl][j], 0, 0, 0);
#pragma unroll
            for (int j = 0; j < 4; ++j)
                f[sl][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(d2[sl].h, w1[j].h, f[sl][j], 0, 0, 0);
            if (i > 0) fold(i - 1, sl ^ 1);
#if DG_SCHED
            // issue order of a unit: its three LDS reads first, then every MFMA with the fold's
            // VALU work of the unit before in its shadow
            if (i + 1 < 8) __builtin_amdgcn_sched_group_barrier(0x100, 3, 0);
#pragma unroll
            for (int g = 0; g < 12; ++g) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                if (i > 0) __builtin_amdgcn_sched_group_barrier(0x002, DG_FOLD_VALU, 0);
            }
#endif
        }
        if (DG_PROBE != 2) fold(7, 1);
    }

    // C / D map of the 16 x 16 MFMA: column lane & 15, rows 4 (lane >> 4) + r
    auto store_tiles = [&](auto accumulate) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int u = 8 * wr + i;
            const int t = t0 + u / ups, h = u % ups;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int b = h * 16 + 4 * (lane >> 4) + r;
                float *row = p.out + ((int64_t)t * B + b) * p.ldc;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int col = (nt0 + 4 * wc + j) * 16 + (lane & 15);
                    if (t < p.t_hi && b < B && col < p.N) {
                        const float val = total[i][j][r] * p.out_scale;
                        row[col] = decltype(accumulate)::value ? row[col] + val : val;
                    }
                }
            }
        }
    };
    if (p.accumulate)
        store_tiles(std::true_type{});
    else
        store_tiles(std::false_type{});
    if (DG_PROBE == 4 && blockIdx.x == 0 && lane == 0) {
        // [wave][dma wait, barrier, issue, multiply] in clocks per stage
        __syncthreads();
        for (int k = 0; k < 4; ++k)
            p.out[(size_t)p.ldc * 300 + wave * 4 + k] = (float)pt[k] / (float)(p.ks_hi - p.ks_lo);
    }
}

// W_ih [2 * 4H, N] (row = dir * 4H + gate * H + unit) -> fp16 pieces of w * scale in the K order of
// the exchange buffer and in B-fragment order: stage (dir, P, m), lane l (k group q = l >> 4,
// column l & 15), element e: unit 16 P + 8 m + 2 q + (e >> 2), gate e & 3.
__global__ void __launch_bounds__(256) dgrad16_pack_kernel(const float *w, int64_t ldw, char *out,
                                                           int N, int nt_total, float scale) {
    const int ks = blockIdx.x, lane = threadIdx.x & 63;
    const int nt = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (nt >= nt_total) return;
    const int d = ks / DG_STAGES_PER_DIR, pm = ks % DG_STAGES_PER_DIR, P = pm >> 1, m = pm & 1;
    const int n = nt * 16 + (lane & 15), q = lane >> 4;
    unsigned pc[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int unit = 16 * P + 8 * m + 2 * q + (e >> 2), gate = e & 3;
        float v = n < N ? w[((int64_t)d * DG_GH + gate * DG_H + unit) * ldw + n] * scale : 0.f;
        v = fminf(fmaxf(v, -60000.f), 60000.f);         // saturate, never inf
        const _Float16 h1 = (_Float16)v;
        const _Float16 h2 = (_Float16)(v - (float)h1);
        pc[e] = (unsigned)__builtin_bit_cast(unsigned short, h1) |
                ((unsigned)__builtin_bit_cast(unsigned short, h2) << 16);
    }
    const u32x4 first = {(pc[0] & 0xFFFFu) | (pc[1] << 16), (pc[2] & 0xFFFFu) | (pc[3] << 16),
                         (pc[4] & 0xFFFFu) | (pc[5] << 16), (pc[6] & 0xFFFFu) | (pc[7] << 16)};
    const u32x4 second = {(pc[0] >> 16) | (pc[1] & 0xFFFF0000u), (pc[2] >> 16) | (pc[3] & 0xFFFF0000u),
                          (pc[4] >> 16) | (pc[5] & 0xFFFF0000u), (pc[6] >> 16) | (pc[7] & 0xFFFF0000u)};
    char *dst = out + ((size_t)ks * nt_total + nt) * 2048 + lane * 16;
    *reinterpret_cast<u32x4 *>(dst) = first;
    *reinterpret_cast<u32x4 *>(dst + 1024) = second;
}

}  // namespace

// (rnn_persistent.hip) where the fp16 backward recurrence keeps its exchange blocks and inverse
// scales inside a recurrence workspace
int prnn_b16_published(void *sync, int T, int B, int H, const char **xchg, const float **scales);
// (rnn_step.hip) the barrier words of row block 0 of a recurrence workspace
void *rnn_workspace_sync_block0(void *workspace, int B, int H);

// this file's share of ctcasr_build_flags() (rnn_persistent.hip)
unsigned dgrad16_build_flags() {
    unsigned flags = 0;
    if (DG_PROBE != 0) flags |= CTCASR_BUILD_PROBE_WRONG_RESULTS;
    if (DG_SCHED != 0 || DG_LOADERS != 4 || DG_YOUNG_PRIO != 0 || DG_FOLD_VALU != 2 || 0)
        flags |= CTCASR_BUILD_NONDEFAULT_TUNING;
    return flags;
}

extern "C" size_t ctcasr_dgrad16_packed_bytes(int n) {
    return n > 0 ? (size_t)2 * DG_STAGES_PER_DIR * ((n + 15) / 16) * 2048 : 0;
}

extern "C" int ctcasr_dgrad16_pack_weights(const float *w_ih, int64_t ld_w, int hidden, int n,
                                           float scale, void *packed, ctcasr_stream_t stream) {
    if (!w_ih || !packed || hidden != DG_H || n <= 0 || ld_w < n || !(scale > 0.f))
        return CTCASR_ERR_BAD_ARGUMENT;
    const int nt_total = (n + 15) / 16;
    dgrad16_pack_kernel<<<dim3(2 * DG_STAGES_PER_DIR, (nt_total + 3) / 4), 256, 0,
                          (hipStream_t)stream>>>(w_ih, ld_w, reinterpret_cast<char *>(packed), n,
                                                 nt_total, scale);
    return ctcasr_launch_status();
}

extern "C" int ctcasr_dgrad16_published_offsets(int T, int B, int hidden, size_t *exchange,
                                                size_t *inverse_scales) {
    if (!exchange || !inverse_scales) return CTCASR_ERR_BAD_ARGUMENT;
    char *base = reinterpret_cast<char *>(uintptr_t(4096));
    const char *x;
    const float *sc;
    if (T < 1 || B < 1 || hidden < 1 ||
        prnn_b16_published(rnn_workspace_sync_block0(base, B, hidden), T, B, hidden, &x, &sc) !=
            CTCASR_OK)
        return CTCASR_ERR_BAD_ARGUMENT;
    *exchange = (size_t)(x - base);
    *inverse_scales = (size_t)(reinterpret_cast<const char *>(sc) - base);
    return CTCASR_OK;
}

extern "C" int ctcasr_dgrad16_supported(int cell, int T, int B, int hidden) {
    // (the per-step block offsets inside the exchange buffer are 32-bit, like the buffer descriptor
    // of the recurrence kernel that fills it: ctcasr_rnn_persistent_supported has the same bound)
    return cell == CTCASR_CELL_LSTM && hidden == DG_H && T >= 1 && B >= 1 && B <= 32 &&
           (size_t)(T + 1) * 2 * B * 4 * DG_H * sizeof(float) < (1ull << 31);
}

extern "C" int ctcasr_dgrad16_blockscaled(void *workspace, int T, int B, int hidden,
                                          const void *packed, float scale, int n, float *dx,
                                          int64_t ld_dx, int t_lo, int t_hi, int dir_lo, int dir_hi,
                                          int accumulate, ctcasr_stream_t stream) {
    if (!workspace || !packed || !dx || !ctcasr_dgrad16_supported(CTCASR_CELL_LSTM, T, B, hidden) ||
        n <= 0 || ld_dx < n || t_lo < 0 || t_hi > T || t_lo >= t_hi || dir_lo < 0 || dir_hi > 2 ||
        dir_lo >= dir_hi || !(scale > 0.f))
        return CTCASR_ERR_BAD_ARGUMENT;
    static bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(dgrad16_bs_kernel),
                                hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)DG_LDS_BYTES) != hipSuccess)
            return CTCASR_ERR_LAUNCH;
        attr_set = true;
    }
    DgArgs a = {};
    if (prnn_b16_published(rnn_workspace_sync_block0(workspace, B, hidden), T, B, hidden, &a.xchg,
                           &a.scales) != CTCASR_OK)
        return CTCASR_ERR_BAD_ARGUMENT;
    a.wpk = reinterpret_cast<const char *>(packed);
    a.out = dx; a.ldc = ld_dx;
    a.T = T; a.B = B; a.N = n; a.nt_total = (n + 15) / 16;
    a.t_lo = t_lo; a.t_hi = t_hi;
    a.ks_lo = dir_lo * DG_STAGES_PER_DIR; a.ks_hi = dir_hi * DG_STAGES_PER_DIR;
    a.accumulate = accumulate; a.out_scale = 1.0f / scale;
    a.units_per_step = (B + 15) / 16;
    const int steps_per_tile = DG_UNITS / a.units_per_step;
    a.tiles_m = (t_hi - t_lo + steps_per_tile - 1) / steps_per_tile;
    a.tiles_n = (n + DG_BN - 1) / DG_BN;
    const int tiles = a.tiles_m * a.tiles_n;
    dgrad16_bs_kernel<<<8 * ((tiles + 7) / 8), DG_THREADS, DG_LDS_BYTES, (hipStream_t)stream>>>(a);
    return ctcasr_launch_status();
}
