// Weight gradients of a recurrent layer, dW_ih = dxw^T x and dW_hh = dxw^T h(t -/+ 1) (the gradients
// of the cuDNN layer's two matrices, asr/model.py:194-215), as ONE own kernel per direction and
// step range instead of two library GEMMs + a column split + two rescales.
//
// The products sum over ROWS (time step, utterance): K is the row axis of both operands, and an
// MFMA lane holds 8 consecutive k of one column - 8 values 32 KB apart in the row-major tensors.  A
// pack pass therefore writes both operands transposed, as fp16 pieces in MFMA fragment order
// (HBM-bound, what the column split of the library form cost):
//     packed [stage = 32 rows][column tile of 16][piece][lane][16 B],
//     lane l = (rows 8 (l >> 4) .. + 7 of the stage, column l & 15)
// dxw with a power of two per column (its column is the product's output row: the scale leaves
// through it; from the backward recurrence kernel's column maxima), the layer input x and the layer
// output y - bounded - under their fixed scales; y is packed SHIFTED by one time step (rows before
// the first / behind the last step read zeros: h(-1) = 0), so that stage s of dxw meets stage s of
// every second operand.  The GEMM kernel is csrc/dgrad16.hip's without the block scales: tile 256
// gate columns x 256 input columns, 8 waves, both operands by LDS-DMA (a 1 KB chunk lands as the
// fragment), 96 MFMAs per wave and stage straight into the 128 accumulator registers, the first
// four waves issue the DMAs.  One launch covers both second operands (192 tiles: 128 of W_ih's,
// 64 of W_hh's) - W_hh's 64 tiles alone leave half of the CUs beside a recurrence launch idle.
#include "common.h"

namespace {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
union WgFrag {
    u32x4 u;
    f16x8 h;
};

constexpr int WG_THREADS = 512;
constexpr int WG_TILE = 256;                       // output rows (gate columns) and columns per tile
constexpr int WG_A_BYTES = (WG_TILE / 16) * 2 * 1024, WG_B_BYTES = WG_A_BYTES;
constexpr int WG_STAGE_BYTES = WG_A_BYTES + WG_B_BYTES;
constexpr size_t WG_LDS_BYTES = 2 * (size_t)WG_STAGE_BYTES;
constexpr int WG_LOADERS = 4;

struct WgArgs {
    const char *a;          // packed dxw of the range: [stages][m_tiles][2][1 KB]
    const char *b[2];       // packed second operands: [stages of all rows][n_tiles[k]][2][1 KB]
    float *out[2];          // [m, n[k]] row-major, accumulated into
    int64_t ld[2];
    const float *inv;       // [m] inverse column scales of dxw
    float alpha[2];         // 1 / scale of the second operands
    int m, n[2], stage0[2]; // first stage of the range inside b[k]
    int stages, m_tiles, n_tiles[2], tiles_n[2], tiles_m;
    int parts;              // row ranges one tile's sum is cut into (one workgroup each)
    int *sync;              // [1 + tiles]: time-out word, then the part whose turn it is to add
    long spin_limit;        // polls of a turn word before a part gives up
};

// polls before a part gives up waiting for its turn (~5 s); "wgrad16_spin_limit" of
// ctcasr_set_option shortens it for the test of the give-up path
long g_wgrad_spin_limit = 1L << 24;

__device__ __forceinline__ void wg_dma16(const char *base, unsigned lane_off, unsigned lds_base) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\t"
                 "global_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(lane_off), "s"(base), "s"(lds_base)
                 : "memory");
}

__global__ void __launch_bounds__(WG_THREADS) wgrad16_kernel(WgArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    __shared__ int gave_up;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;
    // A time-out word that is already set (sticky until the host has looked: hip.wgrad16_check):
    // nothing of this launch is added - the step's update is dropped by the guard word anyway, and
    // turn words a part abandoned would make every later part spin to its limit
    if (p.parts > 1 &&
        __hip_atomic_load(p.sync, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0)
        return;
    if (tid == 0) gave_up = 0;
    // tile list: operand 0's tiles_m x tiles_n[0], then operand 1's; within an operand the tiles
    // of one row of tiles are neighbours (they share the A panel in L2)
    const int tiles_all = p.tiles_m * (p.tiles_n[0] + p.tiles_n[1]);
    const int part = blockIdx.x / tiles_all, tile_id = blockIdx.x % tiles_all;
    const int s_lo = (int)((int64_t)p.stages * part / p.parts);
    const int s_hi = (int)((int64_t)p.stages * (part + 1) / p.parts);
    int v = tile_id, which = 0;
    if (v >= p.tiles_m * p.tiles_n[0]) {
        v -= p.tiles_m * p.tiles_n[0];
        which = 1;
    }
    // Workgroup b runs on XCD b % 8, which has its own L2: an XCD takes a compact block of the
    // operand's tiles (4 x 4 where the tile counts allow: 8 operand panels per stage for 16 tiles
    // instead of 17), not every eighth tile of the row-major list.
    int tm, tn;
    {
        const int tiles_n = p.tiles_n[which], count = p.tiles_m * tiles_n;
        if (p.tiles_m % 4 == 0 && tiles_n % 4 == 0 && count % 8 == 0 &&
            (p.tiles_m * p.tiles_n[0]) % 8 == 0) {
            const int pos = (v % 8) * (count / 8) + v / 8;     // position in block-major order
            const int block = pos / 16, in_block = pos % 16, blocks_n = tiles_n / 4;
            tm = (block / blocks_n) * 4 + in_block / 4;
            tn = (block % blocks_n) * 4 + in_block % 4;
        } else {
            tm = v / tiles_n;
            tn = v % tiles_n;
        }
    }
    const int nt_total = p.n_tiles[which];
    const int mt0 = tm * (WG_TILE / 16), nt0 = tn * (WG_TILE / 16);
    const char *b_all = p.b[which] + (size_t)p.stage0[which] * nt_total * 2048;
    const unsigned lds0 = __builtin_amdgcn_readfirstlane(
        (unsigned)(uintptr_t)(__attribute__((address_space(3))) char *)smem);
    const unsigned lane16 = lane * 16;
    // chunk c (0 .. 31) of a stage's A / B panel: column tile c >> 1, piece c & 1 (clamped to the
    // last column tile: rows / columns past the matrix are computed and never stored)
    unsigned a_chunk[8], b_chunk[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        const int chunk = (wave % WG_LOADERS) * 8 + c;
        a_chunk[c] = __builtin_amdgcn_readfirstlane(
            (unsigned)(min(mt0 + (chunk >> 1), p.m_tiles - 1) * 2048 + (chunk & 1) * 1024));
        b_chunk[c] = __builtin_amdgcn_readfirstlane(
            (unsigned)(min(nt0 + (chunk >> 1), nt_total - 1) * 2048 + (chunk & 1) * 1024));
    }
    auto issue = [&](int s, unsigned buf) {
        if (wave >= WG_LOADERS) return;
        const char *a_base = p.a + (size_t)s * p.m_tiles * 2048;
        const char *b_base = b_all + (size_t)s * nt_total * 2048;
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            wg_dma16(b_base + b_chunk[c], lane16, buf + WG_A_BYTES + (wave * 8 + c) * 1024);
            wg_dma16(a_base + a_chunk[c], lane16, buf + (wave * 8 + c) * 1024);
        }
    };

    f32x4 total[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) total[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    issue(s_lo, lds0);
    for (int s = s_lo; s < s_hi; ++s) {
        const int par = (s - s_lo) & 1;
        const char *cur = smem + par * WG_STAGE_BYTES;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (s + 1 < s_hi) issue(s + 1, lds0 + (par ^ 1) * WG_STAGE_BYTES);
        const char *a_rd = cur + wr * (8 * 2048) + lane * 16;
        const char *b_rd = cur + WG_A_BYTES + wc * (4 * 2048) + lane * 16;
        WgFrag x1[4], x2[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            x1[j].u = *reinterpret_cast<const u32x4 *>(b_rd + j * 2048);
            x2[j].u = *reinterpret_cast<const u32x4 *>(b_rd + j * 2048 + 1024);
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            WgFrag d1, d2;
            d1.u = *reinterpret_cast<const u32x4 *>(a_rd + i * 2048);
            d2.u = *reinterpret_cast<const u32x4 *>(a_rd + i * 2048 + 1024);
#pragma unroll
            for (int j = 0; j < 4; ++j)
                total[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(d1.h, x1[j].h, total[i][j], 0, 0, 0);
#pragma unroll
            for (int j = 0; j < 4; ++j)
                total[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(d1.h, x2[j].h, total[i][j], 0, 0, 0);
#pragma unroll
            for (int j = 0; j < 4; ++j)
                total[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(d2.h, x1[j].h, total[i][j], 0, 0, 0);
        }
    }

    // C / D map of the 16 x 16 MFMA: column (= input column n) lane & 15, rows (= gate column m)
    // 4 (lane >> 4) + r
    float *out = p.out[which];
    const int64_t ld = p.ld[which];
    const float alpha = p.alpha[which];
    const int n_cols = p.n[which];
    if (p.parts == 1) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = (mt0 + 8 * wr + i) * 16 + 4 * (lane >> 4) + r;
                if (m >= p.m) continue;
                const float scale = p.inv[m] * alpha;
                float *row = out + (int64_t)m * ld;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int n = (nt0 + 4 * wc + j) * 16 + (lane & 15);
                    if (n < n_cols) row[n] += total[i][j][r] * scale;
                }
            }
        }
        return;
    }
    // The parts of a tile add in order (the same sums on every run): part q waits for the tile's
    // word to read q - part 0 too: the word is 0 exactly when no launch is adding to this tile, so
    // launches of two streams that share the words take turns instead of releasing each other's
    // parts.  Part q - 1 has a lower workgroup id, was dispatched before and waits only on lower ids
    // itself.  dW and the word are read and written at agent scope access by access (sc1 loads /
    // write-through stores): another XCD's L2 never holds a stale or a dirty line of them, and no
    // workgroup has to write back or invalidate a whole L2 - which would cost every other
    // workgroup of its XCD the operand panels they share there.
    // A part that gives up (spin limit, or another part's time-out word) raises the sticky word
    // and leaves WITHOUT adding and without passing the turn on: dW stays a sum of whole parts in
    // order, CTCModel.step_guard drops the step's update, the host raises and zeroes the words.
    int *turn = p.sync + 1 + tile_id;
    if (tid == 0) {
        long spins = 0;
        while (__hip_atomic_load(turn, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != part) {
            __builtin_amdgcn_s_sleep(8);
            if (++spins > p.spin_limit ||
                ((spins & 255) == 0 &&
                 __hip_atomic_load(p.sync, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0)) {
                __hip_atomic_store(p.sync, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                gave_up = 1;
                break;
            }
        }
    }
    __syncthreads();
    if (gave_up) return;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        float have[4][4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int m = (mt0 + 8 * wr + i) * 16 + 4 * (lane >> 4) + r;
            float *row = out + (int64_t)m * ld;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int n = (nt0 + 4 * wc + j) * 16 + (lane & 15);
                have[r][j] = (m < p.m && n < n_cols)
                    ? __hip_atomic_load(row + n, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.f;
            }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int m = (mt0 + 8 * wr + i) * 16 + 4 * (lane >> 4) + r;
            if (m >= p.m) continue;
            const float scale = p.inv[m] * alpha;
            float *row = out + (int64_t)m * ld;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int n = (nt0 + 4 * wc + j) * 16 + (lane & 15);
                if (n < n_cols)
                    __hip_atomic_store(row + n, have[r][j] + total[i][j][r] * scale,
                                       __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    }
    // every store of this workgroup acknowledged at agent scope, then the word
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0)
        __hip_atomic_store(turn, part + 1 == p.parts ? 0 : part + 1, __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_AGENT);
}

// rows [row0, row0 + 32 * stages) of x [rows_total, ld] (zeros outside [0, rows_total)), columns
// [0, cols), as fp16 pieces of x * col_scale[c] * scale in fragment order.  Block = 64 columns x 4
// k groups of one stage.
__global__ void __launch_bounds__(256) wgrad16_pack_kernel(const float *x, int64_t ld,
                                                           int64_t rows_total, int cols,
                                                           int64_t row0, const float *col_scale,
                                                           float scale, char *out, int col_tiles) {
    const int stage = blockIdx.x;
    const int c = blockIdx.y * 64 + (threadIdx.x & 63), kg = threadIdx.x >> 6;
    if (c >= col_tiles * 16) return;
    const float cs = (c < cols ? (col_scale ? col_scale[c] : 1.f) : 0.f) * scale;
    const int64_t r = row0 + (int64_t)stage * 32 + 8 * kg;
    unsigned pc[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int64_t row = r + e;
        float v = (c < cols && row >= 0 && row < rows_total) ? x[row * ld + c] * cs : 0.f;
        v = fminf(fmaxf(v, -60000.f), 60000.f);
        const _Float16 h1 = (_Float16)v;
        const _Float16 h2 = (_Float16)(v - (float)h1);
        pc[e] = (unsigned)__builtin_bit_cast(unsigned short, h1) |
                ((unsigned)__builtin_bit_cast(unsigned short, h2) << 16);
    }
    const u32x4 first = {(pc[0] & 0xFFFFu) | (pc[1] << 16), (pc[2] & 0xFFFFu) | (pc[3] << 16),
                         (pc[4] & 0xFFFFu) | (pc[5] << 16), (pc[6] & 0xFFFFu) | (pc[7] << 16)};
    const u32x4 second = {(pc[0] >> 16) | (pc[1] & 0xFFFF0000u), (pc[2] >> 16) | (pc[3] & 0xFFFF0000u),
                          (pc[4] >> 16) | (pc[5] & 0xFFFF0000u), (pc[6] >> 16) | (pc[7] & 0xFFFF0000u)};
    char *dst = out + (((size_t)stage * col_tiles + (c >> 4)) * 2) * 1024 + (kg * 16 + (c & 15)) * 16;
    *reinterpret_cast<u32x4 *>(dst) = first;
    *reinterpret_cast<u32x4 *>(dst + 1024) = second;
}

}  // namespace

void wgrad16_set_spin_limit(long polls) { g_wgrad_spin_limit = polls > 0 ? polls : 1L << 24; }

extern "C" size_t ctcasr_wgrad16_packed_bytes(int stages, int cols) {
    return stages > 0 && cols > 0 ? (size_t)stages * ((cols + 15) / 16) * 2048 : 0;
}

extern "C" size_t ctcasr_wgrad16_sync_ints(int m, int nx, int ny) {
    const int tm = (m + WG_TILE - 1) / WG_TILE;
    return 1 + (size_t)tm * ((nx + WG_TILE - 1) / WG_TILE + (ny > 0 ? (ny + WG_TILE - 1) / WG_TILE : 0));
}

extern "C" int ctcasr_wgrad16_pack(const float *x, int64_t ld_x, int64_t rows_total, int cols,
                                   int64_t row0, int stages, const float *col_scale, float scale,
                                   void *packed, ctcasr_stream_t stream) {
    if (!x || !packed || rows_total <= 0 || cols <= 0 || ld_x < cols || stages <= 0 ||
        !(scale > 0.f))
        return CTCASR_ERR_BAD_ARGUMENT;
    const int col_tiles = (cols + 15) / 16;
    wgrad16_pack_kernel<<<dim3(stages, (col_tiles * 16 + 63) / 64), 256, 0, (hipStream_t)stream>>>(
        x, ld_x, rows_total, cols, row0, col_scale, scale, reinterpret_cast<char *>(packed),
        col_tiles);
    return ctcasr_launch_status();
}

extern "C" int ctcasr_wgrad16_gemm(const void *d_packed, int m, int stages, const float *inv_scale,
                                   const void *x_packed, int x_stage0, int nx, float x_scale,
                                   float *dw_x, int64_t ld_x, const void *y_packed, int y_stage0,
                                   int ny, float y_scale, float *dw_y, int64_t ld_y,
                                   int parts, int32_t *sync, ctcasr_stream_t stream) {
    if (!d_packed || !inv_scale || !x_packed || !dw_x || m <= 0 || stages <= 0 || nx <= 0 ||
        ld_x < nx || x_stage0 < 0 || !(x_scale > 0.f) ||
        parts < 1 || parts > stages || (parts > 1 && !sync) ||
        (y_packed && (!dw_y || ny <= 0 || ld_y < ny || y_stage0 < 0 || !(y_scale > 0.f))))
        return CTCASR_ERR_BAD_ARGUMENT;
    static bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(wgrad16_kernel),
                                hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)WG_LDS_BYTES) != hipSuccess)
            return CTCASR_ERR_LAUNCH;
        attr_set = true;
    }
    WgArgs a = {};
    a.a = reinterpret_cast<const char *>(d_packed);
    a.m = m; a.stages = stages; a.inv = inv_scale;
    a.m_tiles = (m + 15) / 16;
    a.tiles_m = (m + WG_TILE - 1) / WG_TILE;
    a.b[0] = reinterpret_cast<const char *>(x_packed); a.stage0[0] = x_stage0; a.n[0] = nx;
    a.alpha[0] = 1.0f / x_scale; a.out[0] = dw_x; a.ld[0] = ld_x;
    a.n_tiles[0] = (nx + 15) / 16; a.tiles_n[0] = (nx + WG_TILE - 1) / WG_TILE;
    if (y_packed) {
        a.b[1] = reinterpret_cast<const char *>(y_packed); a.stage0[1] = y_stage0; a.n[1] = ny;
        a.alpha[1] = 1.0f / y_scale; a.out[1] = dw_y; a.ld[1] = ld_y;
        a.n_tiles[1] = (ny + 15) / 16; a.tiles_n[1] = (ny + WG_TILE - 1) / WG_TILE;
    }
    a.parts = parts; a.sync = sync; a.spin_limit = g_wgrad_spin_limit;
    const int tiles = a.tiles_m * (a.tiles_n[0] + a.tiles_n[1]) * parts;
    wgrad16_kernel<<<tiles, WG_THREADS, WG_LDS_BYTES, (hipStream_t)stream>>>(a);
    return ctcasr_launch_status();
}
