// K1: log-mel / MFCC(+delta) feature extraction and per-utterance normalisation on the GPU.
//
// Replaces python_speech_features.logfbank / mfcc / delta as called by load_sample
// (asr/input_functions.py:156-335): pre-emphasis 0.97, 400-sample frames every 160 samples with
// a rectangular window, 1024-point power spectrum / 1024, 80 triangular mel filters 64 Hz-8 kHz,
// log; for MFCC a 40-coefficient ortho DCT-II, lifter 22, c0 <- log frame energy and deltas over
// +-2 frames; float32 cast, optional drop of every second frame, 'none' | 'local' |
// 'local_scalar' normalisation, zero padding to the batch width.
//
// The reference does this arithmetic in float64 (numpy) and casts to float32 at the end; the
// kernels keep float64 for the FFT / filterbank / DCT so the float32 output differs from the
// oracle only by the final rounding.  One workgroup per frame: the 1024-point FFT lives in LDS
// (16 KB), twiddles / filterbank / DCT come from a small constant table (L2-resident).
#include "common.h"

#define FEAT_THREADS 256
#define FEAT_NFFT 1024
#define FEAT_BINS 513
#define FEAT_NFILT 80
#define FEAT_NCEP 40
#define FEAT_FRAME 400
#define FEAT_STEP 160
#define FEAT_MAX_W 4096   // upper bound on the number of non-zero filterbank weights

namespace {

struct FeatTables {
    double tw_re[FEAT_NFFT / 2], tw_im[FEAT_NFFT / 2];
    int fb_start[FEAT_NFILT], fb_count[FEAT_NFILT], fb_offset[FEAT_NFILT];
    double fb_weight[FEAT_MAX_W];
    double dct[FEAT_NCEP][FEAT_NFILT];   // ortho DCT-II rows, lifter folded in
};

__device__ __forceinline__ unsigned bitrev10(unsigned v) { return __brev(v) >> 22; }

// grid (Tmax, B).  raw32: [B, Tmax, 80] float (mel: final pre-normalisation values; mfcc: unused)
// cep64: [B, Tmax, 40] double (mfcc only)
__global__ void __launch_bounds__(FEAT_THREADS)
frame_features_kernel(const int16_t *__restrict__ pcm, const int *__restrict__ num_samples,
                      int n_max, int t_max, int mfcc, const FeatTables *__restrict__ tab,
                      float *__restrict__ raw32, double *__restrict__ cep64) {
    // The frame is real: its 1024-point spectrum comes out of ONE 512-point complex FFT of
    // z[n] = x[2n] + i x[2n+1] (9 stages of 256 butterflies - one per thread - instead of 10 of
    // 512) and an unpacking pass  X[k] = E[k] + W^k O[k]  with  E = (Z[k] + conj Z[512-k]) / 2,
    // O = (Z[k] - conj Z[512-k]) / 2i.
    __shared__ double re[FEAT_NFFT / 2 + 1], im[FEAT_NFFT / 2 + 1];
    __shared__ double pw[FEAT_BINS];
    __shared__ double logmel[FEAT_NFILT];
    __shared__ double wsum[FEAT_THREADS / 64];
    const int t = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
    const int n = num_samples[b];
    const int frames = n <= FEAT_FRAME ? 1 : 1 + (n - FEAT_FRAME + FEAT_STEP - 1) / FEAT_STEP;
    if (t >= frames) return;
    const int16_t *x = pcm + (size_t)b * n_max;
    auto sample = [&](int i) -> double {          // pre-emphasised sample i of the frame
        const int g = t * FEAT_STEP + i;
        if (i >= FEAT_FRAME || g >= n) return 0.0;
        return g == 0 ? (double)x[0] : (double)x[g] - 0.97 * (double)x[g - 1];
    };
    for (int i = tid; i < FEAT_NFFT / 2; i += FEAT_THREADS) {
        const unsigned r = __brev((unsigned)i) >> 23;           // 9-bit reversal
        re[r] = sample(2 * i);
        im[r] = sample(2 * i + 1);
    }
    __syncthreads();
    for (int s = 1; s <= 9; ++s) {
        const int half = 1 << (s - 1);
        const int idx = tid;                                     // 256 butterflies per stage
        const int k = idx & (half - 1);
        const int i0 = ((idx >> (s - 1)) << s) + k, i1 = i0 + half;
        const int tw = k << (10 - s);                            // W_512^(k 2^(9-s)) = W_1024^tw
        const double wr = tab->tw_re[tw], wi = tab->tw_im[tw];
        const double xr = re[i1] * wr - im[i1] * wi, xi = re[i1] * wi + im[i1] * wr;
        const double ar = re[i0], ai = im[i0];
        re[i0] = ar + xr; im[i0] = ai + xi;
        re[i1] = ar - xr; im[i1] = ai - xi;
        __syncthreads();
    }
    // unpack into the power spectrum pw[0..512]; frame energy
    double part = 0.0;
    for (int k = tid; k < FEAT_BINS; k += FEAT_THREADS) {
        const int kz = k & (FEAT_NFFT / 2 - 1), kc = (FEAT_NFFT / 2 - k) & (FEAT_NFFT / 2 - 1);
        const double zr = re[kz], zi = im[kz], cr = re[kc], ci = -im[kc];   // Z[k], conj Z[512-k]
        const double er = 0.5 * (zr + cr), ei = 0.5 * (zi + ci);
        const double dr = 0.5 * (zr - cr), di = 0.5 * (zi - ci);
        const double o_r = di, o_i = -dr;                                    // (d) / i
        double wr, wi;
        if (k < FEAT_NFFT / 2) { wr = tab->tw_re[k]; wi = tab->tw_im[k]; }
        else { wr = -1.0; wi = 0.0; }                                        // W^512
        const double xr = er + o_r * wr - o_i * wi, xi = ei + o_r * wi + o_i * wr;
        const double p = (xr * xr + xi * xi) / (double)FEAT_NFFT;
        part += p;
        pw[k] = p;
    }
    for (int off = 32; off > 0; off >>= 1) part += __shfl_xor(part, off, 64);
    if ((tid & 63) == 0) wsum[tid >> 6] = part;
    __syncthreads();
    if (tid < FEAT_NFILT) {
        double acc = 0.0;
        const int start = tab->fb_start[tid], count = tab->fb_count[tid];
        const double *w = tab->fb_weight + tab->fb_offset[tid];
        for (int i = 0; i < count; ++i) acc += pw[start + i] * w[i];
        if (acc == 0.0) acc = 2.220446049250313e-16;      // numpy.finfo(float).eps
        const double lm = log(acc);
        logmel[tid] = lm;
        if (!mfcc) raw32[((size_t)b * t_max + t) * FEAT_NFILT + tid] = (float)lm;
    }
    if (!mfcc) return;
    __syncthreads();
    if (tid < FEAT_NCEP) {
        double c;
        if (tid == 0) {
            double energy = wsum[0] + wsum[1] + wsum[2] + wsum[3];
            if (energy == 0.0) energy = 2.220446049250313e-16;
            c = log(energy);
        } else {
            c = 0.0;
            for (int j = 0; j < FEAT_NFILT; ++j) c += tab->dct[tid][j] * logmel[j];
        }
        cep64[((size_t)b * t_max + t) * FEAT_NCEP + tid] = c;
    }
}

// mfcc: [cepstra || delta] -> raw32.  grid (Tmax, B), 64 threads.
__global__ void __launch_bounds__(64)
mfcc_delta_kernel(const double *__restrict__ cep64, const int *__restrict__ num_samples,
                  int t_max, float *__restrict__ raw32) {
    const int t = blockIdx.x, b = blockIdx.y, c = threadIdx.x;
    const int n = num_samples[b];
    const int frames = n <= FEAT_FRAME ? 1 : 1 + (n - FEAT_FRAME + FEAT_STEP - 1) / FEAT_STEP;
    if (t >= frames || c >= FEAT_NCEP) return;
    const double *base = cep64 + (size_t)b * t_max * FEAT_NCEP + c;
    double d = 0.0;
    for (int o = -2; o <= 2; ++o) {
        int tt = t + o;
        tt = tt < 0 ? 0 : (tt >= frames ? frames - 1 : tt);
        d += (double)o * base[(size_t)tt * FEAT_NCEP];
    }
    float *out = raw32 + ((size_t)b * t_max + t) * FEAT_NFILT;
    out[c] = (float)base[(size_t)t * FEAT_NCEP];
    out[FEAT_NCEP + c] = (float)(d / 10.0);
}

// frame drop + normalisation + zero padding.  grid (B, NORM_SPLIT), 256 threads: every workgroup
// of an utterance forms the (cheap, L2-resident) column statistics itself, in the same order, and
// writes its own band of output rows - B workgroups alone took 160 us at C2, this grid with one
// column per lane 100 us, with 16-byte loads and 12 row groups per workgroup 29 us.
// norm: 0 none, 1 local (per feature column over time), 2 local_scalar (whole matrix).
#define NORM_SPLIT 16
__global__ void __launch_bounds__(FEAT_THREADS)
normalize_kernel(const float *__restrict__ raw32, const int *__restrict__ num_samples, int t_max,
                 int out_t, int drop, int norm, float *__restrict__ out, int *__restrict__ out_len) {
    // statistics: thread = (4 feature columns, one of 12 row groups): 16-byte loads, every lane
    // busy, eight rows in flight per thread (one column per lane and one accumulator took 100 us
    // for a 5 MB matrix: the loop only waits for its loads)
    constexpr int RG = 12, C4 = FEAT_NFILT / 4;              // 12 x 20 = 240 threads
    __shared__ double s1[RG][FEAT_NFILT], s2[RG][FEAT_NFILT];
    __shared__ float mean_s[FEAT_NFILT], inv_s[FEAT_NFILT];
    const int b = blockIdx.x, tid = threadIdx.x;
    const int n = num_samples[b];
    const int frames = n <= FEAT_FRAME ? 1 : 1 + (n - FEAT_FRAME + FEAT_STEP - 1) / FEAT_STEP;
    const int step = drop ? 2 : 1;
    const int kept = (frames + step - 1) / step;
    const float *src = raw32 + (size_t)b * t_max * FEAT_NFILT;
    if (tid == 0 && blockIdx.y == 0) out_len[b] = kept;
    if (norm != 0) {
        if (tid < RG * C4) {
            const int c4 = tid % C4, rg = tid / C4;
            double a[4] = {0.0, 0.0, 0.0, 0.0}, q[4] = {0.0, 0.0, 0.0, 0.0};
            for (int i = rg; i < kept; i += 8 * RG) {
                float4 v[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int row = i + RG * j;
                    v[j] = row < kept
                               ? *reinterpret_cast<const float4 *>(
                                     src + (size_t)row * step * FEAT_NFILT + 4 * c4)
                               : make_float4(0.f, 0.f, 0.f, 0.f);
                }
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    a[0] += (double)v[j].x; q[0] += (double)v[j].x * (double)v[j].x;
                    a[1] += (double)v[j].y; q[1] += (double)v[j].y * (double)v[j].y;
                    a[2] += (double)v[j].z; q[2] += (double)v[j].z * (double)v[j].z;
                    a[3] += (double)v[j].w; q[3] += (double)v[j].w * (double)v[j].w;
                }
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) { s1[rg][4 * c4 + k] = a[k]; s2[rg][4 * c4 + k] = q[k]; }
        }
        __syncthreads();
        if (tid < FEAT_NFILT) {
            double a = 0.0, q = 0.0;
            for (int rg = 0; rg < RG; ++rg) { a += s1[rg][tid]; q += s2[rg][tid]; }
            s1[0][tid] = a; s2[0][tid] = q;
        }
        __syncthreads();
        if (tid < FEAT_NFILT) {
            double a = s1[0][tid], q = s2[0][tid], cnt = (double)kept;
            if (norm == 2) {
                a = 0.0; q = 0.0;
                for (int c = 0; c < FEAT_NFILT; ++c) { a += s1[0][c]; q += s2[0][c]; }
                cnt *= FEAT_NFILT;
            }
            const double mean = a / cnt;
            const double var = fmax(q / cnt - mean * mean, 0.0);
            mean_s[tid] = (float)mean;
            inv_s[tid] = (float)(1.0 / sqrt(var));   // no epsilon, like the reference
        }
        __syncthreads();
    }
    float *dst = out + (size_t)b * out_t * FEAT_NFILT;
    const int band = (out_t + gridDim.y - 1) / gridDim.y;
    const int row_lo = blockIdx.y * band, row_hi = min(out_t, row_lo + band);
    for (int i = row_lo * FEAT_NFILT + tid; i < row_hi * FEAT_NFILT; i += FEAT_THREADS) {
        const int t = i / FEAT_NFILT, c = i % FEAT_NFILT;
        float v = 0.f;
        if (t < kept) {
            v = src[(size_t)t * step * FEAT_NFILT + c];
            if (norm != 0) v = (v - mean_s[c]) * inv_s[c];
        }
        dst[i] = v;
    }
}

void build_tables(FeatTables *t, int sampling_rate) {
    const double pi = 3.14159265358979323846;
    for (int k = 0; k < FEAT_NFFT / 2; ++k) {
        t->tw_re[k] = cos(-2.0 * pi * k / FEAT_NFFT);
        t->tw_im[k] = sin(-2.0 * pi * k / FEAT_NFFT);
    }
    // python_speech_features.get_filterbanks(80, 1024, rate, 64, rate / 2)
    const double low = 64.0, high = sampling_rate / 2.0;
    const double lowmel = 2595.0 * log10(1.0 + low / 700.0);
    const double highmel = 2595.0 * log10(1.0 + high / 700.0);
    double edge[FEAT_NFILT + 2];
    for (int i = 0; i < FEAT_NFILT + 2; ++i) {
        const double mel = lowmel + (highmel - lowmel) * i / (FEAT_NFILT + 1);
        const double hz = 700.0 * (pow(10.0, mel / 2595.0) - 1.0);
        edge[i] = floor((FEAT_NFFT + 1) * hz / sampling_rate);
    }
    int cursor = 0;
    for (int j = 0; j < FEAT_NFILT; ++j) {
        const int lo = (int)edge[j], mid = (int)edge[j + 1], hi = (int)edge[j + 2];
        t->fb_start[j] = lo; t->fb_offset[j] = cursor;
        int count = 0;
        for (int i = lo; i < mid && cursor < FEAT_MAX_W; ++i, ++count)
            t->fb_weight[cursor++] = (i - edge[j]) / (edge[j + 1] - edge[j]);
        for (int i = mid; i < hi && cursor < FEAT_MAX_W; ++i, ++count)
            t->fb_weight[cursor++] = (edge[j + 2] - i) / (edge[j + 2] - edge[j + 1]);
        t->fb_count[j] = count;
    }
    for (int c = 0; c < FEAT_NCEP; ++c) {
        const double scale = c == 0 ? sqrt(1.0 / (4.0 * FEAT_NFILT)) : sqrt(1.0 / (2.0 * FEAT_NFILT));
        const double lift = 1.0 + 11.0 * sin(pi * c / 22.0);
        for (int j = 0; j < FEAT_NFILT; ++j)
            t->dct[c][j] = lift * scale * 2.0 * cos(pi * c * (2 * j + 1) / (2.0 * FEAT_NFILT));
    }
}

int frames_of(int n) { return n <= FEAT_FRAME ? 1 : 1 + (n - FEAT_FRAME + FEAT_STEP - 1) / FEAT_STEP; }

size_t tables_bytes() { return ctcasr_align_up(sizeof(FeatTables), 256); }

}  // namespace

extern "C" int ctcasr_features_num_frames(int num_samples) {
    return num_samples < 1 ? 0 : frames_of(num_samples);
}

extern "C" size_t ctcasr_features_tables_bytes(void) { return tables_bytes(); }

// Fills `tables` (device, ctcasr_features_tables_bytes()) for the given sampling rate: FFT
// twiddles, the psf mel filterbank in sparse form, the liftered DCT matrix.  Call once.
extern "C" int ctcasr_features_init_tables(void *tables, int sampling_rate, ctcasr_stream_t stream) {
    if (!tables || sampling_rate <= 128) return CTCASR_ERR_BAD_ARGUMENT;
    static FeatTables host;     // pageable staging buffer; the copy below is synchronous
    build_tables(&host, sampling_rate);
    if (hipStreamSynchronize((hipStream_t)stream) != hipSuccess) return CTCASR_ERR_LAUNCH;
    if (hipMemcpy(tables, &host, sizeof(FeatTables), hipMemcpyHostToDevice) != hipSuccess)
        return CTCASR_ERR_LAUNCH;
    return CTCASR_OK;
}

extern "C" size_t ctcasr_features_workspace_bytes(int B, int max_samples) {
    if (B <= 0 || max_samples <= 0) return 0;
    const size_t t_max = (size_t)frames_of(max_samples);
    return ctcasr_align_up((size_t)B * t_max * FEAT_NFILT * sizeof(float), 256) +
           ctcasr_align_up((size_t)B * t_max * FEAT_NCEP * sizeof(double), 256);
}

extern "C" int ctcasr_features(const int16_t *pcm, const int32_t *num_samples, int B,
                               int max_samples, int feature_type, int normalization,
                               int drop_every_second_frame, const void *tables, float *out,
                               int out_frames, int32_t *out_len, void *workspace,
                               size_t workspace_bytes, ctcasr_stream_t stream) {
    if (!pcm || !num_samples || !tables || !out || !out_len || B <= 0 || max_samples <= 0)
        return CTCASR_ERR_BAD_ARGUMENT;
    if (feature_type < 0 || feature_type > 1 || normalization < 0 || normalization > 2)
        return CTCASR_ERR_BAD_ARGUMENT;
    const int t_max = frames_of(max_samples);
    const int kept_max = drop_every_second_frame ? (t_max + 1) / 2 : t_max;
    if (out_frames < kept_max) return CTCASR_ERR_BAD_ARGUMENT;
    if (!workspace || workspace_bytes < ctcasr_features_workspace_bytes(B, max_samples))
        return CTCASR_ERR_WORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    float *raw32 = reinterpret_cast<float *>(workspace);
    double *cep64 = reinterpret_cast<double *>(
        reinterpret_cast<char *>(workspace) +
        ctcasr_align_up((size_t)B * t_max * FEAT_NFILT * sizeof(float), 256));
    const FeatTables *tab = reinterpret_cast<const FeatTables *>(tables);
    dim3 grid(t_max, B);
    frame_features_kernel<<<grid, FEAT_THREADS, 0, s>>>(pcm, num_samples, max_samples, t_max,
                                                        feature_type, tab, raw32, cep64);
    if (feature_type == 1)
        mfcc_delta_kernel<<<grid, 64, 0, s>>>(cep64, num_samples, t_max, raw32);
    normalize_kernel<<<dim3(B, NORM_SPLIT), FEAT_THREADS, 0, s>>>(raw32, num_samples, t_max, out_frames,
                                                drop_every_second_frame, normalization, out,
                                                out_len);
    return ctcasr_launch_status();
}
