// C[M, N] (+)= A[M, K] . B[N, K]^T in fp32 accuracy on the bf16 matrix pipe, splitting in registers.
//
// Same arithmetic as split.hip + a bf16 library GEMM over K-concatenated pieces:
// a = a1 + a2 + a3 (three bf16 pieces, exact), six piece products of order <= 2, fp32 accumulate.
// Here the fp32 tiles are loaded once, split on their way into LDS, and every pair of fragments
// read from LDS feeds the MFMAs of all the products it takes part in: no split pass over HBM, 4
// instead of 12 operand bytes per element from L2, half the LDS reads per MFMA of the
// K-concatenated form, and no inter-workgroup waits (the library's stream-K kernels have them).
//
// Workgroup = 256 threads (4 waves, one per SIMD, 2 x 2), tile 256 x 256, wave tile 128 x 128 =
// 4 x 4 MFMA tiles of v_mfma_f32_32x32x16_bf16 (256 accumulator registers per lane); K step 16
// fp32 columns: 2 x 16 KB of fp32 from global memory -> 2 x 3 x 8 KB of pieces in LDS, double
// buffered (96 KB), 96 MFMAs per wave and step.
#include "common.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int SG_BM = 256, SG_BN = 256, SG_BK = 16, SG_THREADS = 256;
// issue pattern of one K step (see k_step): the first SG_QUIET MFMAs run without split work (the
// global loads issued one step earlier get that long to land), then SG_VALU VALU instructions
// per MFMA, and after every SG_UNIT MFMAs the LDS writes of one unit and its refilling load
#ifndef SG_QUIET
#define SG_QUIET 24
#endif
#ifndef SG_VALU
#define SG_VALU 4
#endif
#ifndef SG_UNIT
#define SG_UNIT 9
#endif
constexpr int SG_PIECE = SG_BM * SG_BK;                 // bf16 elements of one piece of one tile
constexpr int SG_STAGE = 2 * 3 * SG_PIECE;              // A pieces then B pieces
constexpr size_t SG_LDS_BYTES = 2ull * SG_STAGE * sizeof(__bf16);     // 98304

struct SgArgs {
    const float *a, *b;
    float *c;
    int64_t lda, ldb, ldc;
    int m, n, k, accumulate, tiles_m, tiles_n;
};

__device__ __forceinline__ void split4(const float4 v, bf16x4 &p1, bf16x4 &p2, bf16x4 &p3) {
    const float x[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const __bf16 b1 = (__bf16)x[e];
        const float r1 = x[e] - (float)b1;
        const __bf16 b2 = (__bf16)r1;
        const float r2 = r1 - (float)b2;
        p1[e] = b1;
        p2[e] = b2;
        p3[e] = (__bf16)r2;
    }
}

// TN = false: A [M, K], B [N, K] (K is the column axis of both: projections, data gradient)
// TN = true:  A [K, M], B [K, N] (K is the row axis of both: weight gradients); any K, rows past
//             it count as zeros
template <bool TN>
__global__ void __launch_bounds__(SG_THREADS) split_gemm_kernel(SgArgs p) {
    // Two separate arrays, not one dynamic block: the compiler then knows that the LDS writes of
    // the next stage cannot alias the fragment reads of this one and is free to move them (and
    // the VALU work that feeds them) in between the MFMAs.
    __shared__ __attribute__((aligned(16))) __bf16 stage0[SG_STAGE];
    __shared__ __attribute__((aligned(16))) __bf16 stage1[SG_STAGE];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1;

    // XCD-aware tile order: consecutive blockIdx go round the 8 XCDs; each XCD walks a contiguous
    // range of tiles, four tile rows for one tile column, then the next column - the 32
    // workgroups an XCD runs at a time share 4 A panels and 8 B panels in its L2.
    const int tiles = p.tiles_m * p.tiles_n;
    const int per_xcd = (tiles + 7) / 8;
    const int v = (int)(blockIdx.x & 7) * per_xcd + (int)(blockIdx.x >> 3);
    if (v >= tiles) return;
    constexpr int GROUP = 4;
    const int group = v / (GROUP * p.tiles_n), within = v - group * GROUP * p.tiles_n;
    const int rows_here = min(GROUP, p.tiles_m - group * GROUP);
    const int tm = group * GROUP + within % rows_here, tn = within / rows_here;
    const int m0 = tm * SG_BM, n0 = tn * SG_BN;

    // global -> register staging.  NT: thread t, unit i: row (t >> 2) + 64 i of the tile, columns
    // 4 (t & 3) .. of the K step (16-byte loads).  TN: thread t owns COLUMN t of both tiles and
    // loads its 16 K rows one by one (a wave reads 256 contiguous bytes per row).
    // (Rows / columns past the matrix are clamped to its last one: they only feed rows / columns of
    // C that are never stored, so those loads need no predicate.)
    const int lrow = tid >> 2, lkq = tid & 3;
    const float *ga[4], *gb[4];
    float4 ra[4], rb[4];
    if (TN) {
        ga[0] = p.a + min(m0 + tid, p.m - 1);
        gb[0] = p.b + min(n0 + tid, p.n - 1);
    } else {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            ga[i] = p.a + (int64_t)min(m0 + lrow + 64 * i, p.m - 1) * p.lda + 4 * lkq;
            gb[i] = p.b + (int64_t)min(n0 + lrow + 64 * i, p.n - 1) * p.ldb + 4 * lkq;
        }
    }
    auto load_stage = [&](int k0) {
        if (TN) {
            float *fa = reinterpret_cast<float *>(ra), *fbv = reinterpret_cast<float *>(rb);
#pragma unroll
            for (int kk = 0; kk < SG_BK; ++kk) {
                const int row = k0 + kk;
                const float keep = row < p.k ? 1.f : 0.f;        // (wave-uniform)
                const int64_t r = min(row, p.k - 1);
                fa[kk] = keep * ga[0][r * p.lda];
                fbv[kk] = keep * gb[0][r * p.ldb];
            }
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                ra[i] = *reinterpret_cast<const float4 *>(ga[i] + k0);
                rb[i] = *reinterpret_cast<const float4 *>(gb[i] + k0);
            }
        }
    };
    // piece layout in LDS: [k group of 8][row][8] - the 32 rows a half-wave reads are 512
    // contiguous bytes (row-major [row][16] makes lanes r and r + 8 share banks)
    auto store_stage = [&](__bf16 *base) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            // NT: unit i = 4 columns (k) of row lrow + 64 i;  TN: unit i = k rows 4 i .. 4 i + 3
            // of column tid
            const int off = TN ? (i >> 1) * (SG_BM * 8) + tid * 8 + 4 * (i & 1)
                               : (lkq >> 1) * (SG_BM * 8) + (lrow + 64 * i) * 8 + 4 * (lkq & 1);
            bf16x4 p1, p2, p3;
            split4(ra[i], p1, p2, p3);
            *reinterpret_cast<bf16x4 *>(base + 0 * SG_PIECE + off) = p1;
            *reinterpret_cast<bf16x4 *>(base + 1 * SG_PIECE + off) = p2;
            *reinterpret_cast<bf16x4 *>(base + 2 * SG_PIECE + off) = p3;
            split4(rb[i], p1, p2, p3);
            *reinterpret_cast<bf16x4 *>(base + 3 * SG_PIECE + off) = p1;
            *reinterpret_cast<bf16x4 *>(base + 4 * SG_PIECE + off) = p2;
            *reinterpret_cast<bf16x4 *>(base + 5 * SG_PIECE + off) = p3;
        }
    };

    f32x16 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int steps = (p.k + SG_BK - 1) / SG_BK;
    load_stage(0);
    store_stage(stage0);
    load_stage(min(1, steps - 1) * SG_BK);
    __syncthreads();

    // fragment addresses: row (lane & 31) of an MFMA tile, k group (lane >> 5) of 8
    const int frag = (lane >> 5) * (SG_BM * 8) + (lane & 31) * 8;
    const int fa_off = (wr * 128) * 8 + frag, fb_off = 3 * SG_PIECE + (wc * 128) * 8 + frag;

    // One K step: MFMAs on the fragments of `cur`; meanwhile the registers (stage s + 1) are split
    // into `nxt` and refilled with stage s + 2.  No conditions inside (one basic block: the
    // split's VALU work is scheduled into the MFMAs' shadow): the last steps store / load a stage
    // nobody reads.
    auto k_step = [&](const __bf16 *cur, __bf16 *nxt, int s) {
        const __bf16 *sa = cur + fa_off, *sb = cur + fb_off;
        bf16x8 fb[3][4];
#pragma unroll
        for (int q = 0; q < 3; ++q)
#pragma unroll
            for (int j = 0; j < 4; ++j)
                fb[q][j] = *reinterpret_cast<const bf16x8 *>(sb + q * SG_PIECE + j * 32 * 8);
        store_stage(nxt);
        load_stage(min(s + 2, steps - 1) * SG_BK);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            bf16x8 fa[3];
#pragma unroll
            for (int q = 0; q < 3; ++q)
                fa[q] = *reinterpret_cast<const bf16x8 *>(sa + q * SG_PIECE + i * 32 * 8);
            // (consecutive MFMAs go to different accumulators: a dependent one would wait for
            // its predecessor's last pass)
#define SG_ROW(pa, pb)                                                                       \
    _Pragma("unroll") for (int j = 0; j < 4; ++j)                                           \
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[pa], fb[pb][j], acc[i][j], 0, 0, 0);
            SG_ROW(2, 0) SG_ROW(1, 1) SG_ROW(0, 2) SG_ROW(1, 0) SG_ROW(0, 1) SG_ROW(0, 0)
#undef SG_ROW
        }
        // Issue order of the block (left alone the scheduler runs the ~240 VALU instructions of
        // the split, then the LDS traffic, then the 96 MFMAs): B fragments and the first row of A
        // fragments, then every MFMA with three VALU instructions in its shadow; after each
        // twelfth MFMA one of the eight 16-byte units of stage s + 1 has been split - its LDS
        // writes and the global load that refills its registers follow; the next row of A
        // fragments is read ten MFMAs before it is used.
#define SG_GROUP(mask, count) __builtin_amdgcn_sched_group_barrier(mask, count, 0)
        SG_GROUP(0x100, 15);
#pragma unroll
        for (int g = 0; g < 96; ++g) {
            SG_GROUP(0x008, 1);
            if (g >= SG_QUIET) {
                SG_GROUP(0x002, SG_VALU);
                if ((g - SG_QUIET) % SG_UNIT == SG_UNIT - 1) {
                    SG_GROUP(0x200, 2);
                    SG_GROUP(0x020, TN ? 4 : 1);
                }
            }
            if (g == 14 || g == 38 || g == 62) SG_GROUP(0x100, 3);
        }
#undef SG_GROUP
        __syncthreads();
    };
    int s = 0;
    for (; s + 1 < steps; s += 2) {
        k_step(stage0, stage1, s);
        k_step(stage1, stage0, s + 1);
    }
    if (s < steps) k_step(stage0, stage1, s);

    // C / D map of the 32 x 32 MFMA: col = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5)
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int col = n0 + wc * 128 + j * 32 + (lane & 31);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + wr * 128 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                if (row < p.m && col < p.n) {
                    float *dst = p.c + (int64_t)row * p.ldc + col;
                    *dst = p.accumulate ? *dst + acc[i][j][r] : acc[i][j][r];
                }
            }
        }
}

}  // namespace

static int launch_split_gemm(bool tn, const float *a, int64_t lda, const float *b, int64_t ldb,
                             float *c, int64_t ldc, int m, int n, int k, int accumulate,
                             hipStream_t stream) {
    SgArgs args{a, b, c, lda, ldb, ldc, m, n, k, accumulate,
                (m + SG_BM - 1) / SG_BM, (n + SG_BN - 1) / SG_BN};
    const int tiles = args.tiles_m * args.tiles_n;
    const int grid = 8 * ((tiles + 7) / 8);
    if (tn)
        split_gemm_kernel<true><<<grid, SG_THREADS, 0, stream>>>(args);
    else
        split_gemm_kernel<false><<<grid, SG_THREADS, 0, stream>>>(args);
    return ctcasr_launch_status();
}

extern "C" int ctcasr_gemm_split_nt(const float *a, int64_t lda, const float *b, int64_t ldb,
                                    float *c, int64_t ldc, int m, int n, int k, int accumulate,
                                    ctcasr_stream_t stream) {
    if (!a || !b || !c || m <= 0 || n <= 0 || k <= 0 || k % SG_BK != 0 || lda < k || ldb < k ||
        ldc < n || lda % 4 != 0 || ldb % 4 != 0)
        return CTCASR_ERR_BAD_ARGUMENT;
    if ((reinterpret_cast<uintptr_t>(a) | reinterpret_cast<uintptr_t>(b)) % 16 != 0)
        return CTCASR_ERR_BAD_ARGUMENT;
    return launch_split_gemm(false, a, lda, b, ldb, c, ldc, m, n, k, accumulate,
                             (hipStream_t)stream);
}

extern "C" int ctcasr_gemm_split_tn(const float *a, int64_t lda, const float *b, int64_t ldb,
                                    float *c, int64_t ldc, int m, int n, int k, int accumulate,
                                    ctcasr_stream_t stream) {
    if (!a || !b || !c || m <= 0 || n <= 0 || k <= 0 || lda < m || ldb < n || ldc < n)
        return CTCASR_ERR_BAD_ARGUMENT;
    return launch_split_gemm(true, a, lda, b, ldb, c, ldc, m, n, k, accumulate,
                             (hipStream_t)stream);
}
