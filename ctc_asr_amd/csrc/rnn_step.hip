// K4/K5, streaming variant: one launch per time step, recurrent weights re-read from L2/MALL.
//
// This is the generic path of ctcasr_rnn_fwd/bwd (any H that is a multiple of 64, any B); the
// LDS-resident persistent variant in rnn_persistent.hip takes over when the per-CU weight slice
// fits.  Per step and direction the work is the skinny GEMM  [B, H] x [H, G*H]  (forward) or
// [B, G*H] x [G*H, H] (backward data) on v_mfma_f32_16x16x4_f32, K split over the 4 waves of a
// workgroup, with the gate non-linearities / cell update fused behind the reduction so the
// pre-activations never touch HBM.
//
// Fragment trick used throughout: for a 16-float K chunk, lane l loads ONE float4 at
// k = 16q + 4*(l>>4) for both operands; MFMA r (r = 0..3) then multiplies element r of those
// float4s, i.e. lane group g covers k = 16q + 4g + r.  Every k is used exactly once, and both
// operands are fetched with 16-byte loads instead of the 4-byte loads the textbook layout needs.
#include "common.h"

#define RNN_THREADS 256

namespace {

struct StepArgs {
    const float *xw;       // [T, B, 2, G*H]
    const float *w;        // fwd: w_hh [2, G*H, H]; bwd: w_hh_t [2, H, G*H]
    const int *seq_len;    // [B] or nullptr
    float *y;              // [T, B, 2H]
    const float *dy;       // bwd
    float *dxw;            // bwd out [T, B, 2, G*H]
    float *gates;          // reserve: LSTM [T, B, 2, 4H] post-activation
    float *cells;          // reserve: LSTM [T, B, 2, H]
    float *hbuf;           // [2, 2, B, H] state ping-pong (fwd)
    float *cbuf;           // [2, B, H] LSTM cell state (fwd) / dc or GRU dh*z carry (bwd)
    const float *b_hh;     // GRU: recurrent bias [2, 3H] (candidate-gate third is read)
    const float *xw_bias;  // forward: [2, G*H] added to xw (NULL: none)
    float *drec;           // GRU bwd: gradient w.r.t. the recurrent pre-activations [T,B,2,3H]
    int T, B, H, step;
};

__device__ __forceinline__ float4 ldg4(const float *p) {
    return *reinterpret_cast<const float4 *>(p);
}

__device__ __forceinline__ void mma4(f32x4 &acc, const float4 &a, const float4 &b) {
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, b.x, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, b.y, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, b.z, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, b.w, acc, 0, 0, 0);
}

// number of steps row b runs, and the time index it touches at step s
__device__ __forceinline__ int row_steps(const int *seq_len, int b, int T) {
    return seq_len ? min(max(seq_len[b], 0), T) : T;
}
__device__ __forceinline__ int row_time(int dir, int s, int steps) {
    return dir == 0 ? s : steps - 1 - s;
}

// ---------------------------------------------------------------------------------------------
// forward step.  G gates, UPB = 32 / G units per workgroup -> 32 gate columns = 2 MFMA N tiles.
// grid = (H / UPB, 2 directions, ceil(B / 16))
// ---------------------------------------------------------------------------------------------
template <int CELL> struct CellTraits {
    static constexpr int G = CELL == CTCASR_CELL_LSTM ? 4 : (CELL == CTCASR_CELL_GRU ? 3 : 1);
    static constexpr int UPB = CELL == CTCASR_CELL_LSTM ? 8 : (CELL == CTCASR_CELL_GRU ? 16 : 32);
    static constexpr int NT = G * UPB / 16;     // MFMA N tiles per workgroup (2, 3, 2)
};

template <int CELL>
__global__ void __launch_bounds__(RNN_THREADS) rnn_fwd_step_kernel(StepArgs p) {
    constexpr int G = CellTraits<CELL>::G;
    constexpr int UPB = CellTraits<CELL>::UPB;
    constexpr int NT = CellTraits<CELL>::NT;
    __shared__ float red[4][NT][16][17];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int dir = blockIdx.y, mt = blockIdx.z;
    const int u0 = blockIdx.x * UPB;
    const int H = p.H, B = p.B;
    const int pp = p.step & 1;
    const float *hprev = p.hbuf + ((size_t)(pp * 2 + dir) * B) * H;
    float *hnext = p.hbuf + ((size_t)((pp ^ 1) * 2 + dir) * B) * H;

    f32x4 acc[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) acc[nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (p.step > 0) {
        const int row = mt * 16 + (lane & 15);
        const bool a_ok = row < B;
        const int kq = 4 * (lane >> 4);
        const float *arow = hprev + (size_t)(a_ok ? row : 0) * H + kq;
        const float *brow[NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            int c = nt * 16 + (lane & 15);
            int wrow = (c / UPB) * H + u0 + (c % UPB);
            brow[nt] = p.w + ((size_t)dir * G * H + wrow) * H + kq;
        }
        const int kbeg = wave * (H / 4), kend = kbeg + H / 4;
#pragma unroll 4
        for (int k = kbeg; k < kend; k += 16) {
            float4 a = a_ok ? ldg4(arow + k) : make_float4(0.f, 0.f, 0.f, 0.f);
            float4 bfr[NT];
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) bfr[nt] = ldg4(brow[nt] + k);
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) mma4(acc[nt], a, bfr[nt]);
        }
    }
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int r = 0; r < 4; ++r) red[wave][nt][4 * (lane >> 4) + r][lane & 15] = acc[nt][r];
    __syncthreads();

    for (int item = tid; item < 16 * UPB; item += RNN_THREADS) {
        const int bl = item / UPB, u = item % UPB;
        const int b = mt * 16 + bl;
        if (b >= B) continue;
        const int unit = u0 + u;
        const int steps = row_steps(p.seq_len, b, p.T);
        const size_t hoff = (size_t)b * H + unit;
        if (p.step >= steps) {   // beyond this row's length: carry the state, emit nothing
            hnext[hoff] = p.step > 0 ? hprev[hoff] : 0.f;
            continue;
        }
        const int t = row_time(dir, p.step, steps);
        float rec[G];
#pragma unroll
        for (int g = 0; g < G; ++g) {
            int c = g * UPB + u;
            rec[g] = red[0][c >> 4][bl][c & 15] + red[1][c >> 4][bl][c & 15] +
                     red[2][c >> 4][bl][c & 15] + red[3][c >> 4][bl][c & 15];
        }
        // input projection (+ its bias when the caller did not fold it into the GEMM)
        float xv[G];
        {
            const float *xw = p.xw + (((size_t)t * B + b) * 2 + dir) * G * H + unit;
#pragma unroll
            for (int g = 0; g < G; ++g) xv[g] = xw[(size_t)g * H];
            if (p.xw_bias) {
                const float *xb = p.xw_bias + (size_t)dir * G * H + unit;
#pragma unroll
                for (int g = 0; g < G; ++g) xv[g] += xb[(size_t)g * H];
            }
        }
        float h;
        if (CELL == CTCASR_CELL_LSTM) {
            float gi = sigmoidf_(xv[0] + rec[0]);
            float gf = sigmoidf_(xv[G > 1 ? 1 : 0] + rec[G > 1 ? 1 : 0]);
            float gg = tanhf_(xv[G > 2 ? 2 : 0] + rec[G > 2 ? 2 : 0]);
            float go = sigmoidf_(xv[G > 3 ? 3 : 0] + rec[G > 3 ? 3 : 0]);
            float *cst = p.cbuf + ((size_t)dir * B + b) * H + unit;
            float cprev = p.step > 0 ? *cst : 0.f;
            float c = gf * cprev + gi * gg;
            *cst = c;
            h = go * tanhf_(c);
            float *gr = p.gates + (((size_t)t * B + b) * 2 + dir) * 4 * H + unit;
            gr[0] = gi; gr[H] = gf; gr[2 * H] = gg; gr[3 * H] = go;
            p.cells[(((size_t)t * B + b) * 2 + dir) * H + unit] = c;
        } else if (CELL == CTCASR_CELL_GRU) {
            // cuDNN GRU: n = tanh(W_n x + b_Wn + r * (R_n h + b_Rn)); h = (1 - z) n + z h_prev
            const float hp = p.step > 0 ? hprev[hoff] : 0.f;
            const float gr_ = sigmoidf_(xv[0] + rec[0]);
            const float gz = sigmoidf_(xv[G > 1 ? 1 : 0] + rec[G > 1 ? 1 : 0]);
            const float q = rec[G > 2 ? 2 : 0] + p.b_hh[(size_t)dir * 3 * H + 2 * H + unit];
            const float gn = tanhf_(xv[G > 2 ? 2 : 0] + gr_ * q);
            h = (1.f - gz) * gn + gz * hp;
            float *rs = p.gates + (((size_t)t * B + b) * 2 + dir) * 4 * H + unit;
            rs[0] = gr_; rs[H] = gz; rs[2 * H] = gn; rs[3 * H] = q;
        } else {
            float pre = xv[0] + rec[0];
            h = CELL == CTCASR_CELL_RNN_RELU ? fmaxf(pre, 0.f) : tanhf_(pre);
        }
        hnext[hoff] = h;
        p.y[((size_t)t * B + b) * 2 * H + dir * H + unit] = h;
    }
}

// ---------------------------------------------------------------------------------------------
// backward step s: dh_rec = dgates(step s+1) x W_hh  for 16 units, fused with the cell
// derivative of step s for those units.  grid = (H / 16, 2, ceil(B / 16))
// ---------------------------------------------------------------------------------------------
template <int CELL>
__global__ void __launch_bounds__(RNN_THREADS) rnn_bwd_step_kernel(StepArgs p) {
    constexpr int G = CellTraits<CELL>::G;
    __shared__ float red[4][16][17];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int dir = blockIdx.y, mt = blockIdx.z;
    const int u0 = blockIdx.x * 16;
    const int H = p.H, B = p.B, GH = G * p.H;
    const int s = p.step;

    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    {
        const int row = mt * 16 + (lane & 15);
        bool a_ok = row < B;
        int steps = a_ok ? row_steps(p.seq_len, row, p.T) : 0;
        a_ok = a_ok && (s + 1 < steps);
        const int kq = 4 * (lane >> 4);
        const int t_next = a_ok ? row_time(dir, s + 1, steps) : 0;
        // the recurrent path back-propagates d(pre-activation of R h): dxw itself, except for
        // the GRU whose candidate gate is scaled by r there (kept in a buffer of its own)
        const float *dsrc = CELL == CTCASR_CELL_GRU ? p.drec : p.dxw;
        const float *arow = dsrc + (((size_t)t_next * B + (a_ok ? row : 0)) * 2 + dir) * GH + kq;
        const float *brow = p.w + ((size_t)dir * H + u0 + (lane & 15)) * GH + kq;
        const int kbeg = wave * (GH / 4), kend = kbeg + GH / 4;
        // wave-uniform early out is not possible (rows differ), but a fully dead tile is cheap:
        if (__any(a_ok)) {
#pragma unroll 4
            for (int k = kbeg; k < kend; k += 16) {
                float4 a = a_ok ? ldg4(arow + k) : make_float4(0.f, 0.f, 0.f, 0.f);
                float4 b = ldg4(brow + k);
                mma4(acc, a, b);
            }
        }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) red[wave][4 * (lane >> 4) + r][lane & 15] = acc[r];
    __syncthreads();

    const int bl = tid >> 4, u = tid & 15;
    const int b = mt * 16 + bl;
    if (b >= B) return;
    const int steps = row_steps(p.seq_len, b, p.T);
    if (s >= steps) return;
    const int unit = u0 + u;
    const int t = row_time(dir, s, steps);
    const float dh = p.dy[((size_t)t * B + b) * 2 * H + dir * H + unit] +
                     red[0][bl][u] + red[1][bl][u] + red[2][bl][u] + red[3][bl][u];
    float *dx = p.dxw + (((size_t)t * B + b) * 2 + dir) * GH + unit;
    if (CELL == CTCASR_CELL_LSTM) {
        const float *gr = p.gates + (((size_t)t * B + b) * 2 + dir) * 4 * H + unit;
        const float gi = gr[0], gf = gr[H], gg = gr[2 * H], go = gr[3 * H];
        const float c = p.cells[(((size_t)t * B + b) * 2 + dir) * H + unit];
        float cprev = 0.f;
        if (s > 0) {
            const int tp = row_time(dir, s - 1, steps);
            cprev = p.cells[(((size_t)tp * B + b) * 2 + dir) * H + unit];
        }
        float *dcst = p.cbuf + ((size_t)dir * B + b) * H + unit;
        const float dc_in = (s + 1 < steps) ? *dcst : 0.f;
        const float tc = tanhf_(c);
        const float dc = dc_in + dh * go * (1.f - tc * tc);
        dx[0] = dc * gg * gi * (1.f - gi);
        dx[H] = dc * cprev * gf * (1.f - gf);
        dx[2 * H] = dc * gi * (1.f - gg * gg);
        dx[3 * H] = dh * tc * go * (1.f - go);
        *dcst = dc * gf;
    } else if (CELL == CTCASR_CELL_GRU) {
        const float *rs = p.gates + (((size_t)t * B + b) * 2 + dir) * 4 * H + unit;
        const float gr_ = rs[0], gz = rs[H], gn = rs[2 * H], q = rs[3 * H];
        float hp = 0.f;
        if (s > 0) {
            const int tp = row_time(dir, s - 1, steps);
            hp = p.y[((size_t)tp * B + b) * 2 * H + dir * H + unit];
        }
        float *carry = p.cbuf + ((size_t)dir * B + b) * H + unit;    // dh_{s+1} * z_{s+1}
        const float dht = dh + ((s + 1 < steps) ? *carry : 0.f);
        const float dpre_n = dht * (1.f - gz) * (1.f - gn * gn);
        const float dpre_z = dht * (hp - gn) * gz * (1.f - gz);
        const float dpre_r = dpre_n * q * gr_ * (1.f - gr_);
        dx[0] = dpre_r; dx[H] = dpre_z; dx[2 * H] = dpre_n;
        float *dr = p.drec + (((size_t)t * B + b) * 2 + dir) * GH + unit;
        dr[0] = dpre_r; dr[H] = dpre_z; dr[2 * H] = dpre_n * gr_;
        *carry = dht * gz;
    } else {
        const float h = p.y[((size_t)t * B + b) * 2 * H + dir * H + unit];
        dx[0] = CELL == CTCASR_CELL_RNN_RELU ? (h > 0.f ? dh : 0.f) : dh * (1.f - h * h);
    }
}

int cell_gates(int cell) {
    switch (cell) {
        case CTCASR_CELL_LSTM: return 4;
        case CTCASR_CELL_GRU: return 3;
        case CTCASR_CELL_RNN_RELU:
        case CTCASR_CELL_RNN_TANH: return 1;
        default: return 0;
    }
}

}  // namespace

// persistent (LDS-resident weights) variant, rnn_persistent.hip
extern "C" int ctcasr_rnn_persistent_supported(int cell, int T, int B, int H);
size_t prnn_sync_bytes();
size_t prnn_error_offset();
void prnn_kp_region(int T, int B, int H, int G, size_t *offset, size_t *bytes);
int prnn_resident_gate(void *sync, unsigned ticket, int max_wait_us, hipStream_t s);
size_t prnn_exchange_bytes(int T, int B, int H, int G);
int prnn_fwd(int cell, const float *xw, const float *xw_bias, const float *w_hh,
             const float *b_hh_n, const int32_t *seq_len, int T, int B, int BS, int H, float *y,
             void *y16, float *gates, float *cells, void *sync, float *carry, int step_begin,
             int step_end, int flags, hipStream_t s);
int prnn_bwd(int cell, const float *dy, const float *y, const float *w_hh_t,
             const int32_t *seq_len, int T, int B, int BS, int H, const float *gates,
             const float *cells, float *dxw, float *drec, float *dbias, unsigned *colmax,
             void *sync, float *carry, int step_begin, int step_end, int flags, hipStream_t s);

// The persistent kernels cover at most 32 rows (two 16-row tiles) per launch: a bigger batch
// (33..64) runs as consecutive launches over blocks of rows, each block with its own barrier
// words, exchange buffer and carry in the workspace, the tensors addressed with the full batch
// as stride.
#define PRNN_BLOCK_ROWS 32
static int prnn_blocks(int B) { return (B + PRNN_BLOCK_ROWS - 1) / PRNN_BLOCK_ROWS; }
static int prnn_block_rows(int B, int block) {
    const int left = B - block * PRNN_BLOCK_ROWS;
    return left < PRNN_BLOCK_ROWS ? left : PRNN_BLOCK_ROWS;
}
// bytes of one block's region: barrier words, then its exchange buffer
static size_t prnn_block_bytes(int T, int B, int H, int G) {
    const int rows = B < PRNN_BLOCK_ROWS ? B : PRNN_BLOCK_ROWS;
    return ctcasr_align_up(prnn_sync_bytes(), 256) + prnn_exchange_bytes(T, rows, H, G);
}

static size_t rnn_state_bytes(int B, int H) {
    return ctcasr_align_up((size_t)6 * B * H * sizeof(float), 256);
}

// (dgrad16.hip) the barrier words of row block 0 inside a recurrence workspace: the exchange
// buffer of a pass over B <= 32 rows follows them
void *rnn_workspace_sync_block0(void *workspace, int B, int H) {
    return reinterpret_cast<char *>(workspace) + rnn_state_bytes(B, H);
}

extern "C" size_t ctcasr_rnn_reserve_bytes(int cell, int T, int B, int H) {
    if (T <= 0 || B <= 0 || H <= 0) return 0;
    if (cell == CTCASR_CELL_LSTM) return (size_t)T * B * 2 * 5 * H * sizeof(float);
    // GRU: r, z, n, q = R_n h + b_Rn  [T,B,2,4H], then (backward) drec [T,B,2,3H]
    if (cell == CTCASR_CELL_GRU) return (size_t)T * B * 2 * 7 * H * sizeof(float);
    return 256;   // plain RNN cells recompute their derivative from y
}

extern "C" size_t ctcasr_rnn_workspace_bytes(int cell, int T, int B, int H) {
    if (T <= 0 || B <= 0 || H <= 0 || cell_gates(cell) == 0) return 0;
    // state ping-pong [2,2,B,H] + cell / dc carry [2,B,H]
    // persistent variant: + the per-step exchange buffer (h forward, dgates backward)
    return rnn_state_bytes(B, H) +
           (ctcasr_rnn_persistent_supported(cell, T, B, H)
                ? prnn_blocks(B) * prnn_block_bytes(T, B, H, cell_gates(cell))
                : ctcasr_align_up(prnn_sync_bytes(), 256));
}

static int rnn_check(int cell, int T, int B, int H) {
    if (T <= 0 || B <= 0 || H <= 0) return CTCASR_ERR_BAD_ARGUMENT;
    if (cell_gates(cell) == 0) return CTCASR_ERR_BAD_ARGUMENT;
    if (H % 64 != 0) return CTCASR_ERR_UNSUPPORTED;
    return CTCASR_OK;
}

// Steps [step_begin, step_end) of the forward recurrence.  A whole pass is (0, T); a pass may be
// cut into launches that cover 0..T in ascending order with the same workspace and reserve - the
// state between them (h through the exchange buffer / state ping-pong, c in the carry) stays in
// the workspace.  After a launch, y of the steps it covered is final: their share of the NEXT
// layer's input projection can run on another stream beside the following launch.
// Whether a forward call with these flags runs the fp16-pipe kernel (and so can write `y_pieces`).
extern "C" int ctcasr_rnn_fwd_f16_supported(int cell, int T, int B, int H, int flags) {
    const int rows = B < PRNN_BLOCK_ROWS ? B : PRNN_BLOCK_ROWS;
    return (flags & CTCASR_RNN_F16) && (cell == CTCASR_CELL_LSTM || cell == CTCASR_CELL_GRU) &&
           !(rows > 16 && (flags & CTCASR_RNN_ONE_BARRIER)) &&
           ctcasr_rnn_persistent_supported(cell, T, B, H) ? 1 : 0;
}

extern "C" int ctcasr_rnn_fwd_steps(int cell, const float *xw, const float *xw_bias,
                                    const float *w_hh, const float *b_hh_n,
                                    const int32_t *seq_len, int T, int B,
                                    int H, float *y, void *y_pieces, void *reserve,
                                    void *workspace, size_t workspace_bytes, int step_begin,
                                    int step_end, int flags, ctcasr_stream_t stream) {
    int rc = rnn_check(cell, T, B, H);
    if (rc != CTCASR_OK) return rc;
    // the pieces of y come out of the fp16-pipe kernel only, and only where every row runs all
    // T steps (rows past their length would have to be zero-filled)
    if (y_pieces && (seq_len || !ctcasr_rnn_fwd_f16_supported(cell, T, B, H, flags)))
        return CTCASR_ERR_UNSUPPORTED;
    if (flags & 0xFF & ~(CTCASR_RNN_HALF_CHIP | CTCASR_RNN_WHOLE_CHIP | CTCASR_RNN_ONE_BARRIER |
                         CTCASR_RNN_REDUCE_SCATTER | CTCASR_RNN_F16 | CTCASR_RNN_XCD_SPLIT |
                         CTCASR_RNN_STAGGER | CTCASR_RNN_KPAIR))
        return CTCASR_ERR_BAD_ARGUMENT;
    if (!xw || !w_hh || !y || !reserve) return CTCASR_ERR_BAD_ARGUMENT;
    if (step_begin < 0 || step_end > T || step_begin >= step_end) return CTCASR_ERR_BAD_ARGUMENT;
    if (cell == CTCASR_CELL_GRU && !b_hh_n) return CTCASR_ERR_BAD_ARGUMENT;
    if (!workspace || workspace_bytes < ctcasr_rnn_workspace_bytes(cell, T, B, H))
        return CTCASR_ERR_WORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    StepArgs p = {};
    p.xw = xw; p.w = w_hh; p.seq_len = seq_len; p.y = y;
    p.gates = reinterpret_cast<float *>(reserve);
    p.cells = p.gates + (size_t)T * B * 2 * 4 * H;
    p.hbuf = reinterpret_cast<float *>(workspace);
    p.cbuf = p.hbuf + (size_t)4 * B * H;
    p.T = T; p.B = B; p.H = H; p.b_hh = b_hh_n; p.xw_bias = xw_bias;
    // (rows past their length keep zeros in y)
    if (seq_len && step_begin == 0 &&
        hipMemsetAsync(y, 0, (size_t)T * B * 2 * H * sizeof(float), s) != hipSuccess)
        return CTCASR_ERR_LAUNCH;
    if (ctcasr_rnn_persistent_supported(cell, T, B, H)) {
        const int G = cell_gates(cell);
        for (int blk = 0; blk < prnn_blocks(B); ++blk) {
            const size_t b0 = (size_t)blk * PRNN_BLOCK_ROWS;
            rc = prnn_fwd(cell, xw + b0 * 2 * G * H, xw_bias, w_hh, b_hh_n,
                          seq_len ? seq_len + b0 : nullptr, T, prnn_block_rows(B, blk), B, H,
                          y + b0 * 2 * H,
                          y_pieces ? reinterpret_cast<char *>(y_pieces) + b0 * 3 * 2 * H * 2
                                   : nullptr,
                          p.gates + b0 * 2 * 4 * H, p.cells + b0 * 2 * H,
                          reinterpret_cast<char *>(workspace) + rnn_state_bytes(B, H) +
                              blk * prnn_block_bytes(T, B, H, G),
                          p.cbuf + b0 * 2 * H, step_begin, step_end, flags, s);
            if (rc != CTCASR_OK) return rc;
        }
        return CTCASR_OK;
    }
    const int upb = cell == CTCASR_CELL_LSTM ? 8 : (cell == CTCASR_CELL_GRU ? 16 : 32);
    dim3 grid(H / upb, 2, (B + 15) / 16);
    for (int step = step_begin; step < step_end; ++step) {
        p.step = step;
        if (cell == CTCASR_CELL_LSTM)
            rnn_fwd_step_kernel<CTCASR_CELL_LSTM><<<grid, RNN_THREADS, 0, s>>>(p);
        else if (cell == CTCASR_CELL_GRU)
            rnn_fwd_step_kernel<CTCASR_CELL_GRU><<<grid, RNN_THREADS, 0, s>>>(p);
        else if (cell == CTCASR_CELL_RNN_RELU)
            rnn_fwd_step_kernel<CTCASR_CELL_RNN_RELU><<<grid, RNN_THREADS, 0, s>>>(p);
        else
            rnn_fwd_step_kernel<CTCASR_CELL_RNN_TANH><<<grid, RNN_THREADS, 0, s>>>(p);
    }
    return ctcasr_launch_status();
}

extern "C" int ctcasr_rnn_fwd(int cell, const float *xw, const float *xw_bias, const float *w_hh,
                              const float *b_hh_n, const int32_t *seq_len, int T, int B, int H,
                              float *y, void *reserve, void *workspace, size_t workspace_bytes,
                              ctcasr_stream_t stream) {
    return ctcasr_rnn_fwd_steps(cell, xw, xw_bias, w_hh, b_hh_n, seq_len, T, B, H, y, nullptr,
                                reserve, workspace,
                                workspace_bytes, 0, T, CTCASR_RNN_DEFAULT, stream);
}

// Whether a backward call with these flags runs the fp16-pipe kernel (and so can fill `colmax`).
extern "C" int ctcasr_rnn_bwd_f16_supported(int cell, int T, int B, int H, int flags) {
    const int rows = B < PRNN_BLOCK_ROWS ? B : PRNN_BLOCK_ROWS;
    return (flags & CTCASR_RNN_F16) &&
           ((cell == CTCASR_CELL_LSTM && (H == 1024 || H == 2048)) ||
            (cell == CTCASR_CELL_RNN_RELU && H == 2048 && rows <= 16 && B <= 16)) &&
           ctcasr_rnn_persistent_supported(cell, T, B, H) ? 1 : 0;
}

// Whether a recurrence call runs an fp16-pipe kernel at all (ABI v6; for labels and flags: the
// forward ReLU-2048 kernel writes no `y_pieces`, so ctcasr_rnn_fwd_f16_supported stays false for
// it).  ``ragged``: the call passes per-row lengths.
extern "C" int ctcasr_rnn_f16_recurrence(int cell, int T, int B, int H, int flags, int backward,
                                         int ragged) {
    if (cell == CTCASR_CELL_RNN_RELU)
        return !ragged && ctcasr_rnn_bwd_f16_supported(cell, T, B, H, flags);
    return backward ? ctcasr_rnn_bwd_f16_supported(cell, T, B, H, flags)
                    : ctcasr_rnn_fwd_f16_supported(cell, T, B, H, flags);
}

// Steps [step_begin, step_end) of the backward recurrence, walked downwards.  A whole pass is
// (0, T); a pass may be cut into launches that cover T..0 in descending order with the same
// workspace - the state between them (dh through the exchange buffer / state ping-pong, dc in the
// carry) stays in the workspace.  Between two launches the caller can hand the time steps that
// are already final to weight-gradient GEMMs on another stream.
extern "C" int ctcasr_rnn_bwd_steps(int cell, const float *dy, const float *y,
                                    const float *w_hh_t, const float *b_hh_n,
                                    const int32_t *seq_len, int T, int B, int H,
                                    const void *reserve, float *dxw, float *dbias,
                                    uint32_t *colmax, void *workspace, size_t workspace_bytes,
                                    int step_begin, int step_end, int flags,
                                    ctcasr_stream_t stream) {
    (void)b_hh_n;
    if (colmax && (!ctcasr_rnn_bwd_f16_supported(cell, T, B, H, flags) ||
                   (cell == CTCASR_CELL_RNN_RELU && seq_len)))
        return CTCASR_ERR_UNSUPPORTED;
    int rc = rnn_check(cell, T, B, H);
    if (rc != CTCASR_OK) return rc;
    if (flags & 0xFF & ~(CTCASR_RNN_HALF_CHIP | CTCASR_RNN_WHOLE_CHIP | CTCASR_RNN_ONE_BARRIER |
                         CTCASR_RNN_REDUCE_SCATTER | CTCASR_RNN_F16 | CTCASR_RNN_XCD_SPLIT |
                         CTCASR_RNN_STAGGER | CTCASR_RNN_KPAIR))
        return CTCASR_ERR_BAD_ARGUMENT;
    if (!dy || !y || !w_hh_t || !reserve || !dxw) return CTCASR_ERR_BAD_ARGUMENT;
    if (step_begin < 0 || step_end > T || step_begin >= step_end) return CTCASR_ERR_BAD_ARGUMENT;
    if (!workspace || workspace_bytes < ctcasr_rnn_workspace_bytes(cell, T, B, H))
        return CTCASR_ERR_WORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    const int G = cell_gates(cell);
    StepArgs p = {};
    p.w = w_hh_t; p.seq_len = seq_len; p.y = const_cast<float *>(y); p.dy = dy; p.dxw = dxw;
    p.gates = reinterpret_cast<float *>(const_cast<void *>(reserve));
    p.cells = p.gates + (size_t)T * B * 2 * 4 * H;
    p.hbuf = reinterpret_cast<float *>(workspace);
    p.cbuf = p.hbuf + (size_t)4 * B * H;
    p.T = T; p.B = B; p.H = H;
    p.drec = p.gates + (size_t)T * B * 2 * 4 * H;      // GRU: behind r, z, n, q in the reserve
    // (steps past a row's length carry no gradient)
    if (seq_len && step_end == T &&
        hipMemsetAsync(dxw, 0, (size_t)T * B * 2 * G * H * sizeof(float), s) != hipSuccess)
        return CTCASR_ERR_LAUNCH;
    if (cell == CTCASR_CELL_GRU && seq_len && step_end == T &&
        hipMemsetAsync(p.drec, 0, (size_t)T * B * 2 * G * H * sizeof(float), s) != hipSuccess)
        return CTCASR_ERR_LAUNCH;
    if (ctcasr_rnn_persistent_supported(cell, T, B, H)) {
        for (int blk = 0; blk < prnn_blocks(B); ++blk) {
            const size_t b0 = (size_t)blk * PRNN_BLOCK_ROWS;
            rc = prnn_bwd(cell, dy + b0 * 2 * H, y + b0 * 2 * H, w_hh_t,
                          seq_len ? seq_len + b0 : nullptr, T, prnn_block_rows(B, blk), B, H,
                          p.gates + b0 * 2 * 4 * H, p.cells + b0 * 2 * H, dxw + b0 * 2 * G * H,
                          p.drec + b0 * 2 * G * H, dbias, colmax,
                          reinterpret_cast<char *>(workspace) + rnn_state_bytes(B, H) +
                              blk * prnn_block_bytes(T, B, H, G),
                          p.cbuf + b0 * 2 * H, step_begin, step_end, flags, s);
            if (rc != CTCASR_OK) return rc;
        }
        return CTCASR_OK;
    }
    dim3 grid(H / 16, 2, (B + 15) / 16);
    for (int step = step_end - 1; step >= step_begin; --step) {
        p.step = step;
        if (cell == CTCASR_CELL_LSTM)
            rnn_bwd_step_kernel<CTCASR_CELL_LSTM><<<grid, RNN_THREADS, 0, s>>>(p);
        else if (cell == CTCASR_CELL_GRU)
            rnn_bwd_step_kernel<CTCASR_CELL_GRU><<<grid, RNN_THREADS, 0, s>>>(p);
        else if (cell == CTCASR_CELL_RNN_RELU)
            rnn_bwd_step_kernel<CTCASR_CELL_RNN_RELU><<<grid, RNN_THREADS, 0, s>>>(p);
        else
            rnn_bwd_step_kernel<CTCASR_CELL_RNN_TANH><<<grid, RNN_THREADS, 0, s>>>(p);
    }
    if (ctcasr_launch_status() != CTCASR_OK) return CTCASR_ERR_LAUNCH;
    // bias gradients (streaming path): column sums over the whole pass once its last range -
    // the one that ends at step 0 - is through
    if (dbias && step_begin == 0) {
        rc = ctcasr_colsum_accumulate(dxw, dbias, (int64_t)T * B, 2 * G * H, stream);
        if (rc == CTCASR_OK && cell == CTCASR_CELL_GRU)
            rc = ctcasr_colsum_accumulate(p.drec, dbias + (size_t)2 * G * H, (int64_t)T * B,
                                          2 * G * H, stream);
        return rc;
    }
    return CTCASR_OK;
}

extern "C" int ctcasr_rnn_bwd(int cell, const float *dy, const float *y, const float *w_hh_t,
                              const float *b_hh_n, const int32_t *seq_len, int T, int B, int H,
                              const void *reserve, float *dxw, float *dbias, void *workspace,
                              size_t workspace_bytes, ctcasr_stream_t stream) {
    return ctcasr_rnn_bwd_steps(cell, dy, y, w_hh_t, b_hh_n, seq_len, T, B, H, reserve, dxw,
                                dbias, nullptr, workspace, workspace_bytes, 0, T,
                                CTCASR_RNN_DEFAULT, stream);
}

extern "C" size_t ctcasr_rnn_timeout_word_offset(int cell, int T, int B, int H, int block) {
    if (rnn_check(cell, T, B, H) != CTCASR_OK || !ctcasr_rnn_persistent_supported(cell, T, B, H) ||
        block < 0 || block >= prnn_blocks(B))
        return (size_t)-1;
    return rnn_state_bytes(B, H) + (size_t)block * prnn_block_bytes(T, B, H, cell_gates(cell)) +
           prnn_error_offset();
}

// Synchronises `stream` and reports whether ANY persistent launch that used `workspace` since
// the last poll gave up at a grid barrier (CTCASR_ERR_TIMEOUT); the word is sticky across
// launches and cleared by this call.  Streaming launches never set it.
extern "C" int ctcasr_rnn_poll_error(void *workspace, size_t workspace_bytes, int cell,
                                     int T, int B, int H, ctcasr_stream_t stream) {
    if (!workspace || workspace_bytes < ctcasr_rnn_workspace_bytes(cell, T, B, H))
        return CTCASR_ERR_WORKSPACE;
    if (hipStreamSynchronize((hipStream_t)stream) != hipSuccess) return CTCASR_ERR_LAUNCH;
    if (!ctcasr_rnn_persistent_supported(cell, T, B, H)) return CTCASR_OK;
    bool timed_out = false;
    for (int blk = 0; blk < prnn_blocks(B); ++blk) {
        unsigned err = 0;
        char *sync = reinterpret_cast<char *>(workspace) + rnn_state_bytes(B, H) +
                     blk * prnn_block_bytes(T, B, H, cell_gates(cell));
        if (hipMemcpy(&err, sync + prnn_error_offset(), sizeof(err), hipMemcpyDeviceToHost) !=
            hipSuccess)
            return CTCASR_ERR_LAUNCH;
        // after a time-out the arrival counters are in an undefined state: start over with
        // clean barrier words (this also clears the time-out word)
        if (err && hipMemset(sync, 0, prnn_sync_bytes()) != hipSuccess) return CTCASR_ERR_LAUNCH;
        // ... and so are the write counts of the K-pair hand-off slots (prnn_bwd16k_kernel)
        size_t kp_off = 0, kp_bytes = 0;
        prnn_kp_region(T, prnn_block_rows(B, blk), H, cell_gates(cell), &kp_off, &kp_bytes);
        if (err && kp_bytes && hipMemset(sync + kp_off, 0, kp_bytes) != hipSuccess)
            return CTCASR_ERR_LAUNCH;
        timed_out = timed_out || err != 0;
    }
    return timed_out ? CTCASR_ERR_TIMEOUT : CTCASR_OK;
}

// Enqueue a one-lane gate on `stream` that returns once the persistent launch carrying `ticket`
// (CTCASR_RNN_TICKET(ticket) in its `flags`) on this workspace has ALL its workgroups running,
// or after `max_wait_us`.  Shapes that take the streaming kernels: no-op.
extern "C" int ctcasr_rnn_resident_gate(void *workspace, size_t workspace_bytes, int cell, int T,
                                        int B, int H, unsigned ticket, int max_wait_us,
                                        ctcasr_stream_t stream) {
    if (max_wait_us < 0 || max_wait_us > 100000 || ticket == 0 || ticket > 0xFFFFFFu)
        return CTCASR_ERR_BAD_ARGUMENT;
    if (!workspace || workspace_bytes < ctcasr_rnn_workspace_bytes(cell, T, B, H))
        return CTCASR_ERR_WORKSPACE;
    if (!ctcasr_rnn_persistent_supported(cell, T, B, H) || max_wait_us == 0) return CTCASR_OK;
    // (B = 33 .. 64: a pass runs as one launch per block of <= 32 rows, all carrying the ticket;
    // the gate watches the FIRST block's words - work behind it overlaps every block, and may take
    // CUs the later blocks need at their start: a late start, never a wrong result)
    return prnn_resident_gate(reinterpret_cast<char *>(workspace) + rnn_state_bytes(B, H), ticket,
                              max_wait_us, (hipStream_t)stream);
}

// GRU only: byte offset inside `reserve` of drec [T, B, 2, 3H] - the gradient w.r.t. the
// recurrent pre-activations (dxw with the candidate gate scaled by r) that ctcasr_rnn_bwd leaves
// there; dW_hh = sum_t drec_t^T h_{t-1} and db_hh = column sums of drec are GEMMs of it.
extern "C" size_t ctcasr_rnn_gru_drec_offset(int T, int B, int H) {
    return (size_t)T * B * 2 * 4 * H * sizeof(float);
}
