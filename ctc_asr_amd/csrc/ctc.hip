// K8 / K9 / K10(greedy): log-softmax, CTC loss + gradient, greedy decode for gfx950.
//
// CTC alpha / beta are dependency chains of T' steps per utterance with almost no bytes to move
// (SURVEY.md 8d).  Two launches: ctc_sweep_kernel, grid (B, 2) - workgroup (b, 0) runs the alpha
// recursion of utterance b, workgroup (b, 1) its beta recursion, concurrently; each keeps the
// per-utterance log-softmax table (T' x C floats, 58 KB at T'=500) and the extended-label
// lattice in LDS, so a step costs one s_barrier and a handful of LDS reads, and writes its
// lattice (fp64) to HBM once - and ctc_grad_kernel, one wave per (t, b), which combines alpha
// and beta into the per-class occupancies and the gradient row.
#include "common.h"

// ------------------------------------------------------------------------------------------
// log-softmax over C <= 64 classes: one wave per row.
// ------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) log_softmax_fwd_kernel(const float *__restrict__ x,
                                                               float *__restrict__ y, int rows,
                                                               int C) {
    const int lane = threadIdx.x & 63;
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int nwaves = (gridDim.x * blockDim.x) >> 6;
    for (int r = wave; r < rows; r += nwaves) {
        float v = lane < C ? x[(size_t)r * C + lane] : -INFINITY;
        float mx = wave_max(v);
        float e = lane < C ? expf(v - mx) : 0.f;
        float lz = mx + logf(wave_sum(e));
        if (lane < C) y[(size_t)r * C + lane] = v - lz;
    }
}

__global__ void __launch_bounds__(256) log_softmax_bwd_kernel(const float *__restrict__ y,
                                                               const float *__restrict__ dy,
                                                               float *__restrict__ dx, int rows,
                                                               int C) {
    const int lane = threadIdx.x & 63;
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int nwaves = (gridDim.x * blockDim.x) >> 6;
    for (int r = wave; r < rows; r += nwaves) {
        float g = lane < C ? dy[(size_t)r * C + lane] : 0.f;
        float s = wave_sum(g);
        if (lane < C) dx[(size_t)r * C + lane] = g - expf(y[(size_t)r * C + lane]) * s;
    }
}

extern "C" int ctcasr_log_softmax_fwd(const float *x, float *y, int rows, int C,
                                      ctcasr_stream_t stream) {
    if (!x || !y || rows < 0 || C <= 0) return CTCASR_ERR_BAD_ARGUMENT;
    if (C > 64) return CTCASR_ERR_UNSUPPORTED;
    if (rows == 0) return CTCASR_OK;
    int blocks = (rows + 3) / 4;
    if (blocks > 4096) blocks = 4096;
    log_softmax_fwd_kernel<<<blocks, 256, 0, (hipStream_t)stream>>>(x, y, rows, C);
    return ctcasr_launch_status();
}

extern "C" int ctcasr_log_softmax_bwd(const float *y, const float *dy, float *dx, int rows, int C,
                                      ctcasr_stream_t stream) {
    if (!y || !dy || !dx || rows < 0 || C <= 0) return CTCASR_ERR_BAD_ARGUMENT;
    if (C > 64) return CTCASR_ERR_UNSUPPORTED;
    if (rows == 0) return CTCASR_OK;
    int blocks = (rows + 3) / 4;
    if (blocks > 4096) blocks = 4096;
    log_softmax_bwd_kernel<<<blocks, 256, 0, (hipStream_t)stream>>>(y, dy, dx, rows, C);
    return ctcasr_launch_status();
}

// ------------------------------------------------------------------------------------------
// CTC loss + gradient
// ------------------------------------------------------------------------------------------
#define CTC_THREADS 384
#define CTC_MAX_PER_THREAD 3   // extended labels per thread: S = 2L+1 <= 1152

// log(exp(a) + exp(b) + exp(c)); the state is kept in double so that |alpha| ~ 1e3 at T'=500
// does not cost absolute precision, the exp/log of the O(1) differences run in float.
__device__ __forceinline__ double lse3(double a, double b, double c) {
    double m = fmax(a, fmax(b, c));
    if (m == -INFINITY) return -INFINITY;
    // hardware exp2/log2 (v_exp_f32 / v_log_f32, ~1 ulp): the arguments are O(1) differences and
    // the sum is in [1, 3], so the absolute error per step stays ~1e-7
    float s = __expf((float)(a - m)) + __expf((float)(b - m)) + __expf((float)(c - m));
    return m + (double)__logf(s);
}

// One launch, grid (B, 2): workgroup (b, 0) runs the alpha sweep of utterance b, workgroup (b, 1)
// its beta sweep - the two recursions are independent, each is a chain of `len` barrier-separated
// steps (~0.8 us per step), and B workgroups leave the chip empty anyway.  Both lattices go to
// HBM (fp64 [T, S]); ctc_grad_kernel then forms the per-class occupancies and the gradient for
// every (t, b) in parallel.
template <bool LOGP_IN_LDS>
__global__ void __launch_bounds__(CTC_THREADS)
ctc_sweep_kernel(const float *__restrict__ logits, const int *__restrict__ labels,
                 const int *__restrict__ label_offsets, const int *__restrict__ seq_len, int T,
                 int B, int C, int blank, int s_pad, float *__restrict__ loss,
                 float *__restrict__ grad, int *__restrict__ status,
                 double *__restrict__ alpha_ws, double *__restrict__ beta_ws,
                 double *__restrict__ logpzx_ws, float *__restrict__ logp_ws) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int b = blockIdx.x;
    const bool beta_sweep = blockIdx.y == 1;
    const int tid = threadIdx.x;
    const int L = label_offsets[b + 1] - label_offsets[b];
    const int S = 2 * L + 1;
    const int len = seq_len[b];
    const int *lab = labels + label_offsets[b];

    double *lat0 = reinterpret_cast<double *>(smem);
    double *lat1 = lat0 + s_pad;
    int *ext = reinterpret_cast<int *>(lat1 + s_pad);
    int *flags = reinterpret_cast<int *>(ext + s_pad);   // [0] bad label, [1] adjacent repeats
    float *logp_s = reinterpret_cast<float *>(flags + 4);
    float *logp = LOGP_IN_LDS ? logp_s : logp_ws + ((size_t)blockIdx.y * B + b) * T * C;

    if (tid < 4) flags[tid] = 0;
    __syncthreads();
    for (int u = tid; u < S; u += CTC_THREADS) {
        int sym = blank;
        if (u & 1) {
            sym = lab[u >> 1];
            if (sym < 0 || sym >= C || sym == blank) atomicOr(&flags[0], 1);
            if (u >= 3 && lab[(u >> 1) - 1] == sym) atomicAdd(&flags[1], 1);
        }
        ext[u] = sym;
    }
    __syncthreads();
    int st = 0;
    if (flags[0] || len > T || len < 0) st = 2;
    else if (len < L + flags[1]) st = 1;
    const int live = st == 0 ? len : 0;

    if (!beta_sweep) {
        // gradient rows that carry no signal (beyond seq_len, or the whole utterance on error)
        for (int i = tid; i < (T - live) * C; i += CTC_THREADS) {
            int t = live + i / C, c = i % C;
            grad[((size_t)t * B + b) * C + c] = 0.f;
        }
        if (tid == 0 && st != 0) { status[b] = st; loss[b] = INFINITY; }
        if (tid == 0 && st == 0 && len == 0) { status[b] = 0; loss[b] = 0.f; }
    }
    if (st != 0 || len == 0) return;

    // ---- per-utterance log-softmax table -----------------------------------------------------
    for (int t = tid; t < len; t += CTC_THREADS) {
        const float *row = logits + ((size_t)t * B + b) * C;
        float mx = row[0];
        for (int c = 1; c < C; ++c) mx = fmaxf(mx, row[c]);
        float sum = 0.f;
        for (int c = 0; c < C; ++c) sum += expf(row[c] - mx);
        float lz = mx + logf(sum);
        // a NaN / inf logit anywhere in the utterance makes its loss NaN, as tf.nn.ctc_loss's
        // log-softmax would (asr/model.py:259 -> NanTensorHook, :368): the lattice's max-based
        // log-sum-exp below would otherwise drop NaN paths and report a finite loss while the
        // gradient carries the NaN into the parameters
        if (!(fabsf(lz) <= 3.0e38f)) atomicOr(&flags[2], 1);
        for (int c = 0; c < C; ++c) logp[t * C + c] = row[c] - lz;
    }
    __syncthreads();
    const bool poisoned = flags[2] != 0;

    int my_ext[CTC_MAX_PER_THREAD];
    bool skip_ok[CTC_MAX_PER_THREAD];   // may take the u-2 -> u (alpha) transition
    bool skip_fw[CTC_MAX_PER_THREAD];   // may take the u -> u+2 (beta) transition
#pragma unroll
    for (int i = 0; i < CTC_MAX_PER_THREAD; ++i) {
        int u = tid + i * CTC_THREADS;
        my_ext[i] = u < S ? ext[u] : blank;
        skip_ok[i] = u < S && u >= 2 && ext[u] != blank && ext[u] != ext[u - 2];
        skip_fw[i] = u + 2 < S && ext[u + 2] != blank && ext[u + 2] != ext[u];
    }

    if (!beta_sweep) {
        // ---- alpha ---------------------------------------------------------------------------
        double *aw = alpha_ws + (size_t)b * T * s_pad;
#pragma unroll
        for (int i = 0; i < CTC_MAX_PER_THREAD; ++i) {
            int u = tid + i * CTC_THREADS;
            if (u < S) {
                double v = u == 0 ? (double)logp[blank]
                                  : (u == 1 ? (double)logp[my_ext[i]] : -INFINITY);
                lat0[u] = v;
                aw[u] = v;
            }
        }
        __syncthreads();
        for (int t = 1; t < len; ++t) {
            double *cur = (t & 1) ? lat1 : lat0;
            const double *prev = (t & 1) ? lat0 : lat1;
            const float *lp = logp + t * C;
#pragma unroll
            for (int i = 0; i < CTC_MAX_PER_THREAD; ++i) {
                int u = tid + i * CTC_THREADS;
                if (u < S) {
                    double a0 = prev[u];
                    double a1 = u >= 1 ? prev[u - 1] : -INFINITY;
                    double a2 = skip_ok[i] ? prev[u - 2] : -INFINITY;
                    double v = lse3(a0, a1, a2);
                    if (v != -INFINITY) v += (double)lp[my_ext[i]];
                    cur[u] = v;
                    aw[(size_t)t * s_pad + u] = v;
                }
            }
            __syncthreads();
        }
        const double *fin = ((len - 1) & 1) ? lat1 : lat0;
        const double log_pzx = lse3(fin[S - 1], S > 1 ? fin[S - 2] : -INFINITY, -INFINITY);
        if (tid == 0) {
            status[b] = 0;
            loss[b] = poisoned ? __uint_as_float(0x7FC00000u) : (float)(-log_pzx);
            logpzx_ws[b] = log_pzx;
        }
        return;
    }

    // ---- beta: beta(t, u) excludes the emission at t, so alpha + beta is the joint log-prob ----
    double *bw = beta_ws + (size_t)b * T * s_pad;
#pragma unroll
    for (int i = 0; i < CTC_MAX_PER_THREAD; ++i) {
        int u = tid + i * CTC_THREADS;
        if (u < S) {
            const double v = (u >= S - 2) ? 0.0 : -INFINITY;
            lat0[u] = v;
            bw[(size_t)(len - 1) * s_pad + u] = v;
        }
    }
    __syncthreads();
    int phase = 0;
    for (int t = len - 1; t > 0; --t, phase ^= 1) {
        const double *cur = phase ? lat1 : lat0;
        double *nxt = phase ? lat0 : lat1;
        const float *lp = logp + t * C;
#pragma unroll
        for (int i = 0; i < CTC_MAX_PER_THREAD; ++i) {
            int u = tid + i * CTC_THREADS;
            if (u < S) {
                double b0 = cur[u] + (double)lp[my_ext[i]];
                double b1 = u + 1 < S ? cur[u + 1] + (double)lp[ext[u + 1]] : -INFINITY;
                double b2 = skip_fw[i] ? cur[u + 2] + (double)lp[ext[u + 2]] : -INFINITY;
                const double v = lse3(b0, b1, b2);
                nxt[u] = v;
                bw[(size_t)(t - 1) * s_pad + u] = v;
            }
        }
        __syncthreads();
    }
}

// Gradient w.r.t. the logits: one wave per (t, b).  occ[k] = sum over extended labels u with
// l'[u] = k of exp(alpha(t,u) + beta(t,u) - ln p(l|x)); grad = (softmax(t,k) - occ[k]) * scale.
#define CTC_GRAD_WAVES 4
__global__ void __launch_bounds__(64 * CTC_GRAD_WAVES)
ctc_grad_kernel(const float *__restrict__ logits, const int *__restrict__ labels,
                const int *__restrict__ label_offsets, const int *__restrict__ seq_len, int T,
                int B, int C, int blank, int s_pad, float grad_scale,
                const int *__restrict__ status, const double *__restrict__ alpha_ws,
                const double *__restrict__ beta_ws, const double *__restrict__ logpzx_ws,
                float *__restrict__ grad) {
    __shared__ float occ_s[CTC_GRAD_WAVES][64];
    const int b = blockIdx.y, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int t = blockIdx.x * CTC_GRAD_WAVES + wave;
    if (status[b] != 0 || t >= seq_len[b]) return;     // those rows were zeroed by the sweep
    float *occ = occ_s[wave];
    occ[lane] = 0.f;
    const int L = label_offsets[b + 1] - label_offsets[b];
    const int S = 2 * L + 1;
    const int *lab = labels + label_offsets[b];
    const double log_pzx = logpzx_ws[b];
    const double *aw = alpha_ws + ((size_t)b * T + t) * s_pad;
    const double *bw = beta_ws + ((size_t)b * T + t) * s_pad;
    if (log_pzx != -INFINITY) {
        for (int u = lane; u < S; u += 64) {
            const double joint = aw[u] + bw[u];
            if (joint != -INFINITY)
                atomicAdd(&occ[(u & 1) ? lab[u >> 1] : blank], expf((float)(joint - log_pzx)));
        }
    }
    // softmax of the frame, as the table of the sweeps has it: exp(x - (max + log(sum)))
    const float *row = logits + ((size_t)t * B + b) * C;
    const float x = lane < C ? row[lane] : -INFINITY;
    float mx = x;
    for (int o = 32; o; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
    float sum = lane < C ? expf(x - mx) : 0.f;
    for (int o = 32; o; o >>= 1) sum += __shfl_xor(sum, o, 64);
    const float lz = mx + logf(sum);
    // the wave's LDS atomics are ordered before this read (one in-order LDS queue per wave)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const float occupied = *reinterpret_cast<volatile float *>(&occ[lane]);
    if (lane < C)
        grad[((size_t)t * B + b) * C + lane] = (expf(x - lz) - occupied) * grad_scale;
}

static size_t ctc_lattice_bytes(int T, int B, int max_label_len) {
    return ctcasr_align_up((size_t)B * T * (2 * (size_t)max_label_len + 1) * sizeof(double), 256);
}

extern "C" size_t ctcasr_ctc_loss_workspace_bytes(int T, int B, int C, int max_label_len) {
    if (T <= 0 || B <= 0 || C <= 0 || max_label_len < 0) return 0;
    // alpha, beta lattices; ln p(l|x) per utterance; log-softmax tables of the two sweeps
    return 2 * ctc_lattice_bytes(T, B, max_label_len) +
           ctcasr_align_up((size_t)B * sizeof(double), 256) +
           ctcasr_align_up((size_t)2 * B * T * C * sizeof(float), 256);
}

extern "C" int ctcasr_ctc_loss_fwd_bwd(const float *logits, const int32_t *labels,
                                       const int32_t *label_offsets, const int32_t *seq_len, int T,
                                       int B, int C, int blank, int max_label_len,
                                       float grad_scale, float *loss, float *grad_logits,
                                       int32_t *status, void *workspace, size_t workspace_bytes,
                                       ctcasr_stream_t stream) {
    if (!logits || !labels || !label_offsets || !seq_len || !loss || !grad_logits || !status)
        return CTCASR_ERR_BAD_ARGUMENT;
    if (T <= 0 || B <= 0 || C <= 1 || blank < 0 || blank >= C || max_label_len < 0)
        return CTCASR_ERR_BAD_ARGUMENT;
    if (C > 64) return CTCASR_ERR_UNSUPPORTED;
    const int s_pad = 2 * max_label_len + 1;
    if (s_pad > CTC_THREADS * CTC_MAX_PER_THREAD) return CTCASR_ERR_UNSUPPORTED;
    if (!workspace || workspace_bytes < ctcasr_ctc_loss_workspace_bytes(T, B, C, max_label_len))
        return CTCASR_ERR_WORKSPACE;
    char *ws = reinterpret_cast<char *>(workspace);
    const size_t lattice = ctc_lattice_bytes(T, B, max_label_len);
    double *alpha_ws = reinterpret_cast<double *>(ws);
    double *beta_ws = reinterpret_cast<double *>(ws + lattice);
    double *logpzx_ws = reinterpret_cast<double *>(ws + 2 * lattice);
    float *logp_ws = reinterpret_cast<float *>(ws + 2 * lattice +
                                               ctcasr_align_up((size_t)B * sizeof(double), 256));
    const size_t fixed = (size_t)s_pad * (2 * sizeof(double) + sizeof(int)) + 4 * sizeof(int);
    const size_t table = (size_t)T * C * sizeof(float);
    hipStream_t s = (hipStream_t)stream;
    const dim3 sweeps(B, 2);
    if (fixed + table <= 150 * 1024) {
        size_t lds = fixed + table;
        if (lds > 64 * 1024) {
            hipError_t e = hipFuncSetAttribute(
                reinterpret_cast<const void *>(&ctc_sweep_kernel<true>),
                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e != hipSuccess) return CTCASR_ERR_LAUNCH;
        }
        ctc_sweep_kernel<true><<<sweeps, CTC_THREADS, lds, s>>>(
            logits, labels, label_offsets, seq_len, T, B, C, blank, s_pad, loss, grad_logits,
            status, alpha_ws, beta_ws, logpzx_ws, logp_ws);
    } else {
        ctc_sweep_kernel<false><<<sweeps, CTC_THREADS, fixed, s>>>(
            logits, labels, label_offsets, seq_len, T, B, C, blank, s_pad, loss, grad_logits,
            status, alpha_ws, beta_ws, logpzx_ws, logp_ws);
    }
    const dim3 rows((T + CTC_GRAD_WAVES - 1) / CTC_GRAD_WAVES, B);
    ctc_grad_kernel<<<rows, 64 * CTC_GRAD_WAVES, 0, s>>>(
        logits, labels, label_offsets, seq_len, T, B, C, blank, s_pad, grad_scale, status,
        alpha_ws, beta_ws, logpzx_ws, grad_logits);
    return ctcasr_launch_status();
}

// ------------------------------------------------------------------------------------------
// Greedy decode: one workgroup per utterance; argmax per frame, keep flags, block scan, scatter.
// ------------------------------------------------------------------------------------------
#define GREEDY_THREADS 256

__global__ void __launch_bounds__(GREEDY_THREADS)
greedy_decode_kernel(const float *__restrict__ logits, const int *__restrict__ seq_len, int T,
                     int B, int C, int blank, int *__restrict__ out, int *__restrict__ out_len) {
    __shared__ int carry_sym;      // argmax of the frame before the current chunk
    __shared__ int base;           // symbols emitted so far
    __shared__ int scan[GREEDY_THREADS];
    __shared__ int syms[GREEDY_THREADS];
    const int b = blockIdx.x, tid = threadIdx.x;
    int len = seq_len[b];
    if (len > T) len = T;
    if (len < 0) len = 0;
    if (tid == 0) { carry_sym = -1; base = 0; }
    for (int i = tid; i < T; i += GREEDY_THREADS) out[(size_t)b * T + i] = 0;
    __syncthreads();
    for (int t0 = 0; t0 < len; t0 += GREEDY_THREADS) {
        const int t = t0 + tid;
        int best = -1;
        if (t < len) {
            const float *row = logits + ((size_t)t * B + b) * C;
            float mx = row[0];
            best = 0;
            for (int c = 1; c < C; ++c) {
                float v = row[c];
                if (v > mx) { mx = v; best = c; }
            }
        }
        syms[tid] = best;
        __syncthreads();
        int prev = tid == 0 ? carry_sym : syms[tid - 1];
        int keep = (t < len && best != blank && best != prev) ? 1 : 0;
        scan[tid] = keep;
        __syncthreads();
        for (int off = 1; off < GREEDY_THREADS; off <<= 1) {
            int add = tid >= off ? scan[tid - off] : 0;
            __syncthreads();
            scan[tid] += add;
            __syncthreads();
        }
        if (keep) out[(size_t)b * T + base + scan[tid] - 1] = best;
        __syncthreads();
        if (tid == GREEDY_THREADS - 1) {
            base += scan[tid];
            int last = len - 1 - t0;
            carry_sym = syms[last < GREEDY_THREADS - 1 ? last : GREEDY_THREADS - 1];
        }
        __syncthreads();
    }
    if (tid == 0) out_len[b] = base;
}

extern "C" int ctcasr_ctc_greedy_decode(const float *logits, const int32_t *seq_len, int T, int B,
                                        int C, int blank, int32_t *out, int32_t *out_len,
                                        ctcasr_stream_t stream) {
    if (!logits || !seq_len || !out || !out_len || T <= 0 || B <= 0 || C <= 0)
        return CTCASR_ERR_BAD_ARGUMENT;
    greedy_decode_kernel<<<B, GREEDY_THREADS, 0, (hipStream_t)stream>>>(logits, seq_len, T, B, C,
                                                                       blank, out, out_len);
    return ctcasr_launch_status();
}
