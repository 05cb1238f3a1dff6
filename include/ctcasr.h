/* ctcasr.h — C ABI of the MI355X-native CTC acoustic-model hot path (libctcasr.so).
 *
 * The reference (mdangschat/ctc-asr) has no FFI of its own: its hot path is a chain of
 * TensorFlow / cuDNN / python_speech_features calls made from Python.  Each entry point below
 * replaces one of those call sites (cited per function; paths relative to the reference root)
 * and is what a binding for that call site would bind.  INTEGRATION.md shows the ctypes stubs.
 *
 * Conventions (all functions):
 *   - plain C types only; every pointer is a DEVICE pointer (HBM) owned by the caller unless the
 *     parameter name ends in `_host`; tensors are contiguous, row-major, float32 unless noted;
 *   - `stream` is a hipStream_t passed as void*; work is enqueued asynchronously on it and the
 *     function returns without synchronising; nothing is allocated behind the caller's back —
 *     scratch memory is sized by the matching *_workspace_bytes() query and passed in;
 *   - return value: CTCASR_OK (0) or a negative CTCASR_ERR_* code for an argument / launch
 *     error; functions never throw and never abort.  Per-utterance data errors (an infeasible
 *     CTC alignment) are reported through a device-side `status` array, mirroring the point at
 *     which TensorFlow raises InvalidArgumentError.
 *   - time-major activations: [T, B, *]; direction index 0 = forward, 1 = backward.
 */
#ifndef CTCASR_H_
#define CTCASR_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CTCASR_ABI_VERSION 7

enum {
    CTCASR_OK = 0,
    CTCASR_ERR_BAD_ARGUMENT = -1,   /* null pointer, non-positive size, unknown enum           */
    CTCASR_ERR_UNSUPPORTED = -2,    /* valid request outside what the kernels cover            */
    CTCASR_ERR_WORKSPACE = -3,      /* workspace pointer null or too small                     */
    CTCASR_ERR_LAUNCH = -4,         /* hipGetLastError() reported a launch failure             */
    CTCASR_ERR_TIMEOUT = -5         /* a bounded in-kernel wait gave up (persistent RNN kernel) */
};

/* RNN cell types; names follow FLAGS.rnn_cell (asr/params.py:47-50, asr/model.py:194-199). */
enum {
    CTCASR_CELL_RNN_RELU = 0,
    CTCASR_CELL_RNN_TANH = 1,
    CTCASR_CELL_LSTM = 2,
    CTCASR_CELL_GRU = 3
};

typedef void *ctcasr_stream_t;

int ctcasr_abi_version(void);
/* Bit mask of the non-default compile-time macros this library was built with.  A/B and probe
 * builds of the recurrence kernels (tools/build_alt.sh) come out of the same sources as the
 * product; a loader must refuse a library whose CTCASR_BUILD_PROBE_WRONG_RESULTS bit is set
 * (a timing probe that skips part of the arithmetic ON PURPOSE) unless it was asked for one -
 * ctc_asr_amd/hip.py does (CTCASR_ALLOW_PROBE_BUILD=1 to override).  0 = the product build. */
#define CTCASR_BUILD_PROBE_WRONG_RESULTS 1u   /* e.g. -DPRNN_PROBE_HALF_LOADS=1               */
#define CTCASR_BUILD_NONDEFAULT_TUNING   2u   /* PRNN_GROUPS / _XCD_AWARE / _CHAIN_* / ...     */
unsigned ctcasr_build_flags(void);
/* Process-wide PROFILING switch (no option changes what a call computes or which kernel variant
 * it runs - that is the per-call `flags` argument of ctcasr_rnn_fwd_steps / _bwd_steps):
 * "rnn_kernel_events" (0/1, default 0): record a HIP event pair on the launch stream around every
 * persistent recurrence kernel; ctcasr_rnn_kernel_events() waits for them, returns launch counts
 * and summed durations ([0] forward, [1] backward) and clears the record (benchmarking). */
int ctcasr_set_option(const char *name, int value);
int ctcasr_rnn_kernel_events(int launches[2], double total_ms[2]);
const char *ctcasr_error_string(int code);

/* ---- K8: (log-)softmax over the class axis ------------------------------------------------
 * Replaces the softmax inside tf.nn.ctc_loss / ctc_beam_search_decoder (asr/model.py:259,292).
 * x, y: [rows, C] with C <= 64.  bwd: dx = dy - exp(y) * sum_c(dy). */
int ctcasr_log_softmax_fwd(const float *x, float *y, int rows, int C, ctcasr_stream_t stream);
int ctcasr_log_softmax_bwd(const float *y, const float *dy, float *dx, int rows, int C,
                           ctcasr_stream_t stream);

/* ---- K9: CTC loss + gradient, log-softmax fused ---------------------------------------------
 * Replaces tf.nn.ctc_loss(labels, inputs=logits, sequence_length, time_major=True,
 * ctc_merge_repeated=True, preprocess_collapse_repeated=False) and its gradient
 * (asr/model.py:259-264; the reduce_mean of :267 is applied through `grad_scale` = 1/B).
 *   logits        [T, B, C] raw activations; blank = C - 1 for the reference (pass it anyway)
 *   labels        int32, concatenated label ids of all B utterances (ids in [0, C) \ {blank})
 *   label_offsets int32 [B + 1], labels of utterance b are labels[label_offsets[b] .. [b+1])
 *   seq_len       int32 [B], frames of utterance b that count (<= T)
 *   loss          [B]   -ln p(label_b | x_b); +inf when status[b] != 0
 *   grad_logits   [T, B, C] d(sum_b grad_scale * loss_b) / d logits; rows t >= seq_len[b] are 0
 *   status        int32 [B]: 0 ok; 1 "not enough time for target transition sequence"
 *                 (TensorFlow raises here); 2 label id out of range / seq_len > T
 * Workspace: ctcasr_ctc_loss_workspace_bytes(T, B, C, max_label_len). */
size_t ctcasr_ctc_loss_workspace_bytes(int T, int B, int C, int max_label_len);
int ctcasr_ctc_loss_fwd_bwd(const float *logits, const int32_t *labels,
                            const int32_t *label_offsets, const int32_t *seq_len, int T, int B,
                            int C, int blank, int max_label_len, float grad_scale, float *loss,
                            float *grad_logits, int32_t *status, void *workspace,
                            size_t workspace_bytes, ctcasr_stream_t stream);

/* ---- K10: decoding --------------------------------------------------------------------------
 * Greedy: argmax per frame (first maximum), merge repeats, drop blank — ctc_greedy_decoder
 * semantics (named in asr/model.py:290,298 and required by BASELINE.json).
 *   out [B, T] int32 (row padded with 0), out_len [B]. */
int ctcasr_ctc_greedy_decode(const float *logits, const int32_t *seq_len, int T, int B, int C,
                             int blank, int32_t *out, int32_t *out_len, ctcasr_stream_t stream);

/* Beam search: tf.nn.ctc_beam_search_decoder(inputs=logits, sequence_length,
 * beam_width, top_paths=1, merge_repeated=False) (asr/model.py:292-296).
 *   norm_mode 0: per-frame max subtraction (TensorFlow 1.12); 1: full log-softmax (TF >= 1.14)
 *   out [B, T] int32, out_len [B], logp [B] (log-probability of the top path; may be NULL)
 * Workspace: ctcasr_ctc_beam_workspace_bytes(T, B, C, beam_width). */
size_t ctcasr_ctc_beam_workspace_bytes(int T, int B, int C, int beam_width);
int ctcasr_ctc_beam_decode(const float *logits, const int32_t *seq_len, int T, int B, int C,
                           int blank, int beam_width, int norm_mode, int32_t *out,
                           int32_t *out_len, float *logp, void *workspace,
                           size_t workspace_bytes, ctcasr_stream_t stream);

/* ---- K4/K5: recurrence of one bidirectional RNN layer ----------------------------------------
 * Replaces the time loop of tfc.cudnn_rnn.Cudnn{LSTM,GRU,RNNRelu,RNNTanh}(direction=
 * 'bidirectional') (asr/model.py:194-215) and of stack_bidirectional_dynamic_rnn over
 * BasicRNNCell (asr/model.py:171-183).  The input projection W x + b_W (+ b_R where it commutes)
 * is a plain GEMM done by the caller; this entry point runs h_t = cell(xw_t, h_{t-1}).
 *   cell     CTCASR_CELL_*; G = gates per unit (LSTM 4: i,f,g,o; GRU 3: r,z,n; RNN 1)
 *   xw       [T, B, 2, G*H] pre-computed input projections, both directions
 *   xw_bias  [2, G*H] or NULL: added to xw inside the kernel (b_W, + b_R where it commutes) - the
 *            caller's GEMM then needs no bias epilogue; NULL when xw already includes the bias
 *   w_hh     [2, G*H, H] recurrent weights (cuDNN / torch layout, gate-major rows)
 *   b_hh_n   GRU only (NULL otherwise): the recurrent bias b_hh [2, 3H]; its candidate-gate third
 *            is applied inside r * (R_n h + b_Rn), the r / z thirds must be folded into xw
 *   seq_len  int32 [B] or NULL.  NULL = cuDNN semantics (all T steps for every row; the backward
 *            direction starts at t = T-1).  Non-NULL = dynamic_rnn semantics (steps
 *            t >= seq_len[b] emit zeros and keep the state; backward direction reversed per row).
 *   y        [T, B, 2H]  = [h_fw || h_bw]
 *   reserve  activations kept for the backward pass, ctcasr_rnn_reserve_bytes()
 *   workspace ctcasr_rnn_workspace_bytes() (state ping-pong, grid-barrier words, exchange
 *            buffer).  ZERO-FILL IT ONCE before its first use with a given (B, H): the launches
 *            themselves issue no memset - the arrival counters are reset by the kernel that used
 *            them, the all-zero block of the exchange buffer is never written, and the time-out
 *            word is sticky until ctcasr_rnn_poll_error reads it.
 * bwd: dy [T,B,2H] -> dxw [T,B,2,G*H] (gradient w.r.t. xw, which is also what the weight
 * gradients are GEMMs of), w_hh_t = w_hh transposed to [2, H, G*H] (caller keeps it current;
 * ctcasr_transpose_batched does it).  GRU: the recurrent path differs from dxw in the candidate
 * gate (scaled by r); that tensor, drec [T,B,2,3H], is left in `reserve` at
 * ctcasr_rnn_gru_drec_offset(): dW_hh is a GEMM of it.
 *   dbias    (optional) bias gradients, ACCUMULATED into (caller zeroes them before a pass):
 *            [2, G*H] column sums of dxw over (t, b) = db_ih, then - GRU only - [2, 3H] column sums
 *            of drec = db_hh (for the other cells db_hh = db_ih).  Complete once the launch that
 *            covers step 0 has run (the persistent kernels add each launch's share, the
 *            streaming path sums the whole pass at its end).  Summation order is not fixed
 *            (atomics), like ctcasr_colsum_accumulate's. */
size_t ctcasr_rnn_gru_drec_offset(int T, int B, int H);
size_t ctcasr_rnn_reserve_bytes(int cell, int T, int B, int H);
size_t ctcasr_rnn_workspace_bytes(int cell, int T, int B, int H);
int ctcasr_rnn_fwd(int cell, const float *xw, const float *xw_bias, const float *w_hh,
                   const float *b_hh_n, const int32_t *seq_len, int T, int B, int H, float *y,
                   void *reserve, void *workspace, size_t workspace_bytes, ctcasr_stream_t stream);
/* Steps [step_begin, step_end) of the forward recurrence only; ctcasr_rnn_fwd == (0, T).  A pass may
 * be cut into launches covering 0..T in ascending order on the same workspace and reserve: after
 * a launch, y of the steps it covered is final (time s of the forward direction, time
 * seq_len-1-s of the backward direction), so the next layer's input projection of those steps
 * can run on another stream while the next launch continues the recurrence.
 *
 * `flags` (CTCASR_RNN_*) picks the variant of the persistent kernel for THIS call - there is no
 * process-wide state, so calls from different threads / streams do not interact:
 *   CTCASR_RNN_DEFAULT     forward: all 256 CUs; backward: 128 CUs
 *   CTCASR_RNN_HALF_CHIP   128 CUs (64 workgroups per direction, weights split between LDS and
 *                          registers), so that GEMMs on another stream can run beside it
 *   CTCASR_RNN_WHOLE_CHIP  all 256 CUs
 * Batches of 17..32 rows are two independent 16-row tiles: on the whole chip each tile runs as
 * its own group of half-chip workgroups, on half of the chip as a second chain of 4 waves inside
 * every workgroup - each tile with its own barrier, results identical to one tile at a time.
 * Shapes without the requested variant, and the streaming kernels, ignore it. */
#define CTCASR_RNN_DEFAULT 0
#define CTCASR_RNN_HALF_CHIP 1
#define CTCASR_RNN_WHOLE_CHIP 2
/* batches of 17..32 rows: the round-1 kernels (both 16-row tiles behind ONE barrier per step)
 * instead of the default two independent chains per workgroup - for A/B measurements */
#define CTCASR_RNN_ONE_BARRIER 4
/* backward, LSTM with H = 1024: the reduce-scatter form of the recurrence (every workgroup
 * multiplies the dgates of its OWN units into partial dh tiles for all units, consumers sum 64
 * tiles) instead of the all-gather form (every workgroup reads the dgates of all units) */
#define CTCASR_RNN_REDUCE_SCATTER 8
/* forward, LSTM / GRU on the persistent kernels (ABI v5): the recurrent product h W_hh^T on the
 * fp16 matrix pipe - h (|h| <= 1) and the workgroup's slice of W_hh (scaled by a power of two found
 * inside the kernel from the slice's largest magnitude, so no weight can overflow) as TWO fp16
 * pieces each, three piece products, fp32 accumulation: fp32-grade (22 significand bits, the
 * dropped term <= 2^-22 of a product), not bit-equal to the fp32-MFMA kernel.  Every launch of one
 * pass (step ranges) must carry the same value of this bit: the exchange buffer then holds h as
 * its two pieces.  The backward kernels and everything else are unaffected (y, the reserve and
 * the carry stay fp32).  Other cells / shapes / CTCASR_RNN_ONE_BARRIER ignore the bit. */
#define CTCASR_RNN_F16 16
/* fp16-pipe kernels, both directions in one launch: direction 0's workgroups on XCDs 0 - 3,
 * direction 1's on XCDs 4 - 7 (workgroup b runs on XCD b % 8) - every exchange block crosses the
 * fabric into four L2s instead of eight.  Same results bit for bit (ABI v5). */
#define CTCASR_RNN_XCD_SPLIT 32
/* backward, LSTM with H = 1024 on the fp16 pipe, 24 or 32 rows (17..32 and a multiple of 8: tile 1's
 * rows then start on a cache line of their own in the exchange buffer; ABI v7 - other batches keep
 * the one-barrier kernel) on half of the chip: the two 16-row tiles staggered by half a step (prnn_bwd16s_kernel: each tile with its own arrival counters, one
 * tile's exchange round trip under the other tile's loads) instead of both behind one barrier.
 * Same results bit for bit; every launch of a pass may choose freely.  Calls with per-row lengths
 * and other shapes ignore it.  (ABI v6) */
#define CTCASR_RNN_STAGGER 64
/* backward, same calls as CTCASR_RNN_STAGGER (and taking precedence over it): the K-pair form
 * (prnn_bwd16k_kernel, ABI v7) - pairs of workgroups on one XCD share 32 hidden units, each holds
 * one K half of their weights, reads only that half of the published dgates (128 KB instead of 256
 * per tile and step) and hands its partner a [16 x 16] partial tile through L2.  Publishes exactly
 * what the other fp16 backward kernels publish; results agree with them to rounding (the order of
 * the K sums differs; the hand-off carries a tag in the lowest significand bit of each partial
 * sum), step ranges bit-identical to one launch.  Every launch of a pass may choose freely. */
#define CTCASR_RNN_KPAIR 128
/* Residency ticket (bits 8..31 of `flags`, 0 = none): a persistent launch that carries one posts
 * it in the workspace once ALL of its workgroups are running; ctcasr_rnn_resident_gate() makes
 * another stream wait for exactly that (bounded).  Use: work for the CUs a half-chip launch
 * leaves free is enqueued behind such a gate, so that it cannot occupy the chip first and make
 * the persistent kernel wait for CUs.  Tickets are caller-chosen launch numbers in 1..2^24-1,
 * increasing per workspace, compared modulo 2^24.  (ABI v3; replaces ctcasr_stream_delay.) */
#define CTCASR_RNN_TICKET(ticket) ((int)(((unsigned)(ticket) & 0xFFFFFFu) << 8))
int ctcasr_rnn_fwd_steps(int cell, const float *xw, const float *xw_bias, const float *w_hh,
                         const float *b_hh_n, const int32_t *seq_len, int T, int B, int H,
                         float *y, void *y_pieces, void *reserve, void *workspace,
                         size_t workspace_bytes, int step_begin, int step_end, int flags,
                         ctcasr_stream_t stream);
/* ABI v5.  `y_pieces` (optional; only where ctcasr_rnn_fwd_f16_supported and seq_len == NULL, else
 * CTCASR_ERR_UNSUPPORTED): fp16 [T, B, 3, 2H] - the two fp16 pieces of y * 2^15 in the block
 * layout of ctcasr_split_f16(order 0, 0, 1), written by the fp16-pipe forward kernel itself (it
 * has them in registers: they are what it publishes to the other workgroups): the operand of the
 * next layer's forward projection / dense4 and of this layer's recurrent weight gradient, without
 * a split pass over y. */
int ctcasr_rnn_fwd_f16_supported(int cell, int T, int B, int H, int flags);
/* 1 when the LDS-resident single-launch kernels cover (cell, T, B, H) on this device, else the
 * per-step streaming kernels run.  CTCASR_RNN_MODE=stream in the environment forces the latter. */
int ctcasr_rnn_persistent_supported(int cell, int T, int B, int H);
/* Synchronises the stream and returns CTCASR_ERR_TIMEOUT if ANY persistent launch that used
 * `workspace` since the previous poll abandoned a grid barrier (the results of that pass are then
 * invalid), else CTCASR_OK.  The time-out word is sticky across launches, layers and passes and
 * is cleared by this call (after a time-out the barrier words and - ABI v7 - the write counts of
 * the K-split kernels' hand-off slots are zero-filled with it). */
int ctcasr_rnn_poll_error(void *workspace, size_t workspace_bytes, int cell, int T, int B,
                          int H, ctcasr_stream_t stream);
/* Enqueues a one-lane gate kernel on `stream` that returns as soon as the persistent launch that
 * carries CTCASR_RNN_TICKET(ticket) on this workspace has every workgroup running (or a later
 * ticket has been posted), and after `max_wait_us` (<= 100 ms) at the latest.  No-op for shapes
 * that take the streaming kernels. */
int ctcasr_rnn_resident_gate(void *workspace, size_t workspace_bytes, int cell, int T, int B,
                             int H, unsigned ticket, int max_wait_us, ctcasr_stream_t stream);
int ctcasr_rnn_bwd(int cell, const float *dy, const float *y, const float *w_hh_t,
                   const float *b_hh_n, const int32_t *seq_len, int T, int B, int H,
                   const void *reserve, float *dxw, float *dbias, void *workspace,
                   size_t workspace_bytes, ctcasr_stream_t stream);
/* Steps [step_begin, step_end) of the backward recurrence only (it walks the steps downwards;
 * step s is time s of the forward direction and time seq_len-1-s of the backward direction).
 * ctcasr_rnn_bwd == (0, T).  A pass may be cut into launches covering T..0 in descending order
 * on the same workspace, e.g. (T/2, T) then (0, T/2): after a launch, dxw of the steps it
 * covered is final, so their share of the weight-gradient GEMMs can run on another stream while
 * the next launch continues the recurrence. */
int ctcasr_rnn_bwd_steps(int cell, const float *dy, const float *y, const float *w_hh_t,
                         const float *b_hh_n, const int32_t *seq_len, int T, int B, int H,
                         const void *reserve, float *dxw, float *dbias, uint32_t *colmax,
                         void *workspace, size_t workspace_bytes, int step_begin, int step_end,
                         int flags, ctcasr_stream_t stream);
/* ABI v5.  CTCASR_RNN_F16 in a BACKWARD call's flags (LSTM, H = 1024 on the persistent kernels;
 * ctcasr_rnn_bwd_f16_supported says whether a call qualifies, every other call ignores the bit):
 * dgates x W_hh on the fp16 matrix pipe - each producer workgroup scales every row of its 64 gate
 * columns by its own power of two and publishes two fp16 pieces, the consumer accumulates every
 * producer's block apart and adds it, unscaled, to an fp32 total; W_hh as in the forward kernel.
 * fp32-grade (22 significand bits per (row, 64-column block)), not bit-equal to the fp32 kernel.
 * `colmax` (optional, only with that kernel - else CTCASR_ERR_UNSUPPORTED): u32[2 * 4H], zeroed
 * by the caller; the kernel raises (atomicMax) word [dir][g H + unit] to the bit pattern of the
 * largest |dxw| of that column over the steps of the call - what ctcasr_colmax_scale's pass over
 * the finished rows of dxw would find (same values, no pass). */
int ctcasr_rnn_bwd_f16_supported(int cell, int T, int B, int H, int flags);
/* ABI v6: whether a recurrence call with these flags runs an fp16-pipe kernel at all (`backward`:
 * the backward pass; `ragged`: the call passes per-row lengths).  Differs from the two functions
 * above for the ReLU cell at H = 2048 (B <= 16, no lengths: round 5), whose forward kernel writes
 * no y_pieces. */
int ctcasr_rnn_f16_recurrence(int cell, int T, int B, int H, int flags, int backward, int ragged);

/* ---- fused dense / conv epilogues (tf.layers.dense + ReLU + tf.minimum(., relu_cutoff) +
 * tf.layers.dropout: asr/util/tf_contrib.py:50-61,122-135, asr/model.py:219-225) --------------
 * fwd (in place): y[r, c] = dropout(min(max(y[r, c] + bias[c], 0), cutoff)); inverted dropout
 *   with keep = 1 - rate, counter-based RNG keyed by (seed, element index); rate 0 = no dropout.
 *   `cutoff` <= 0 disables the activation entirely (plain bias add: the logits layer).
 * bwd: dz = dy * [0 < y < cutoff/keep] / keep * [y != 0] from the stored *output* y (no mask
 *   tensor is kept); dbias[c] += sum_r dz[r, c] when dbias != NULL (caller zeroes it).
 *   `cols` is the fastest dimension; for conv activations in NHWC pass rows = B*T*F, cols = C. */
int ctcasr_bias_act_fwd(float *y, const float *bias, int64_t rows, int cols, float cutoff,
                        float dropout_rate, uint64_t seed, ctcasr_stream_t stream);
int ctcasr_bias_act_bwd(const float *y, const float *dy, float *dz, float *dbias, int64_t rows,
                        int cols, float cutoff, float dropout_rate, ctcasr_stream_t stream);
/* Inverted dropout without activation (cuDNN's inter-layer RNN dropout, DropoutWrapper of the
 * BasicRNNCell path: asr/model.py:203, asr/util/tf_contrib.py:190-194): out = in * mask / keep,
 * mask regenerated from (seed, element index) - call it on the gradient with the same seed for
 * the backward pass.  in == out is allowed. */
int ctcasr_dropout(const float *in, float *out, int64_t n, float dropout_rate, uint64_t seed,
                   ctcasr_stream_t stream);
/* dbias[c] += sum_r dz[r, c] on its own (layers without activation). */
int ctcasr_colsum_accumulate(const float *dz, float *dbias, int64_t rows, int cols,
                             ctcasr_stream_t stream);

/* ---- the 11 x 21, stride (1, 2) convolutions over 32 input channels of the DS2 stack (layers 2
 * and 3 of conv_layers, asr/util/tf_contrib.py:64-146; TensorFlow SAME padding): forward pass and
 * data gradient and kernel gradient as implicit GEMMs on the fp32 MFMA units, NHWC, any number of frames, no padded
 * intermediates.  Covered (ctcasr_conv_s12_supported): freq_in = 40, cout = 32 and freq_in = 20,
 * cout = 96.
 *   fwd:       x  [B, T, freq_in, 32]       -> y  [B, T, freq_in/2, cout] = conv(x) + bias|NULL
 *   bwd_data:  dz [B, T, freq_in/2, cout]   -> dx [B, T, freq_in, 32]
 *   wrw:       dz, x                        -> dw [cout, 32, 11, 21]
 * `packed`: fragment-ordered copies of the kernel w [cout, 32, 11, 21] made by
 * ctcasr_conv_s12_pack_weights, 2 * 11*21*32*cout floats (backward order, then forward order). */
int ctcasr_conv_s12_supported(int freq_in, int cout);
int ctcasr_conv_s12_pack_weights(const float *w, float *packed, int cout, ctcasr_stream_t stream);
/* relu_cutoff > 0: the epilogue also applies min(max(., 0), relu_cutoff) (the ReLU + tf.minimum of
 * conv_layers; 0 = convolution + bias only).  y_time_major / dz_time_major != 0: that tensor is
 * laid out [T, B, F, C] instead of [B, T, F, C] - the last layer of the stack hands its output
 * to the recurrent stack (and takes its gradient back) without a transpose. */
/* ABI v5: the forward pass with its products on the fp16 matrix pipe (csrc/conv16.hip): input and
 * weights as two fp16 pieces each, three piece products per tap, fp32 accumulation - for inputs
 * with a known bound (behind the clipped ReLU of the previous layer): bound * x_scale < 65504,
 * x_scale a power of two.  The weights are packed per step by ctcasr_conv_s12_pack_weights16 into
 * `packed16` (ctcasr_conv_s12_pack16_bytes(cout) bytes: the bit pattern of max |w|, found on the
 * device, then the pieces in fragment order under the scale derived from it). */
size_t ctcasr_conv_s12_pack16_bytes(int cout);
int ctcasr_conv_s12_pack_weights16(const float *w, void *packed16, int cout,
                                   ctcasr_stream_t stream);
int ctcasr_conv_s12_fwd16(const float *x, float x_scale, const void *packed16, const float *bias,
                          float *y, int B, int T, int freq_in, int cout, float relu_cutoff,
                          int y_time_major, ctcasr_stream_t stream);
/* ... and the data gradient (arguments of ctcasr_conv_s12_bwd_data): dz has no bound, so every dz
 * frame of a workgroup's patch (all taps of one kt read one frame) gets its own power-of-two
 * scale while it is staged; the weights' pieces are the backward-order half of `packed16`. */
int ctcasr_conv_s12_bwd_data16(const float *dz, const void *packed16, float *dx, int B, int T,
                               int freq_in, int cout, int dz_time_major, const float *act,
                               float relu_cutoff, ctcasr_stream_t stream);
/* ... and the kernel gradient (arguments of ctcasr_conv_s12_wrw plus x_scale as in
 * ctcasr_conv_s12_fwd16): the summation runs over (b, t, fo), so x * x_scale and the masked dz -
 * scaled per OUTPUT CHANNEL by the power of two that the channel's largest magnitude asks for,
 * found on the device - are first written to `workspace` as fp16 pieces with 8 utterances
 * innermost (the one axis no tap shifts), then one launch multiplies (asr/util/tf_contrib.py:64-146
 * under tf.gradients). */
size_t ctcasr_conv_s12_wrw16_workspace_bytes(int B, int T, int freq_in, int cout);
int ctcasr_conv_s12_wrw16(const float *dz, const float *x, float x_scale, float *dw, int B, int T,
                          int freq_in, int cout, int dz_time_major, const float *act,
                          float relu_cutoff, float *dbias, void *workspace,
                          size_t workspace_bytes, ctcasr_stream_t stream);
int ctcasr_conv_s12_fwd(const float *x, const float *packed, const float *bias, float *y, int B,
                        int T, int freq_in, int cout, float relu_cutoff, int y_time_major,
                        ctcasr_stream_t stream);
/* Backward of the fused epilogue (round 3): `act` != NULL is the layer's stored OUTPUT (same
 * layout as dz) and `dz` the gradient w.r.t. that output - the kernels apply the mask
 * [0 < act < relu_cutoff] while they stage dz, so no elementwise pass runs in front of them; the
 * wrw kernels then also ADD the bias gradient (column sums of the masked dz over b, t, f) to
 * dbias[cout] when it is non-NULL (caller zeroes it; atomics: summation order not fixed).
 * `act` NULL: dz is the pre-activation gradient already (relu_cutoff ignored). */
int ctcasr_conv_s12_bwd_data(const float *dz, const float *packed, float *dx, int B, int T,
                             int freq_in, int cout, int dz_time_major, const float *act,
                             float relu_cutoff, ctcasr_stream_t stream);
/*   wrw: dz [B, T, freq_in/2, cout], x [B, T, freq_in, 32] -> dw [cout, 32, 11, 21] (overwritten):
 *        the kernel gradient, split over (b, t) tiles with a deterministic two-stage reduction;
 *        workspace: per-split partial sums, ctcasr_conv_s12_wrw_workspace_bytes(B, T, freq_in, cout) */
size_t ctcasr_conv_s12_wrw_workspace_bytes(int B, int T, int freq_in, int cout);
int ctcasr_conv_s12_wrw(const float *dz, const float *x, float *dw, int B, int T, int freq_in,
                        int cout, int dz_time_major, const float *act, float relu_cutoff,
                        float *dbias, void *workspace, size_t workspace_bytes,
                        ctcasr_stream_t stream);

/* ---- the first DS2 convolution: 1 -> 32 channels, 11 x 41 taps, stride (2, 2), SAME padding ----
 *   fwd: x [B, T, 80] -> y [B, ceil(T/2), 40, 32] (NHWC) = conv(x) + bias|NULL; w [32, 1, 11, 41] */
int ctcasr_conv0_fwd(const float *x, const float *w, const float *bias, float *y, int B, int T,
                     float relu_cutoff, ctcasr_stream_t stream);
/*   wrw: dz [B, ceil(T/2), 40, 32] (NHWC), x [B, T, 80] -> dw [32, 1, 11, 41] (overwritten);
 *        workspace: per-workgroup partial sums, ctcasr_conv0_wrw_workspace_bytes(B, T) */
size_t ctcasr_conv0_wrw_workspace_bytes(int B, int T);
int ctcasr_conv0_wrw(const float *dz, const float *x, float *dw, int B, int T, const float *act,
                     float relu_cutoff, float *dbias, void *workspace, size_t workspace_bytes,
                     ctcasr_stream_t stream);

/* ... the same two on the fp16 matrix pipe (round 4; csrc/conv16.hip).  fwd16: the weights as
 * fp16 pieces in fragment order (`packed16`: ctcasr_conv0_pack16_bytes() bytes, re-packed every
 * step, scale found on the device); the features have no bound, every workgroup scales its own
 * input patch by the power of two its largest magnitude asks for.  wrw16: dz scaled per output
 * channel, x by the tensor's largest magnitude, both written to `workspace` as fp16 pieces with
 * 8 utterances innermost (see ctcasr_conv_s12_wrw16). */
size_t ctcasr_conv0_pack16_bytes(void);
int ctcasr_conv0_pack_weights16(const float *w, void *packed16, ctcasr_stream_t stream);
int ctcasr_conv0_fwd16(const float *x, const void *packed16, const float *bias, float *y, int B,
                       int T, float relu_cutoff, ctcasr_stream_t stream);
size_t ctcasr_conv0_wrw16_workspace_bytes(int B, int T);
int ctcasr_conv0_wrw16(const float *dz, const float *x, float *dw, int B, int T, const float *act,
                       float relu_cutoff, float *dbias, void *workspace, size_t workspace_bytes,
                       ctcasr_stream_t stream);

/* out[n][c][r] = in[n][r][c] for n < batch (weight re-layouts, e.g. w_hh -> w_hh_t). */
int ctcasr_transpose_batched(const float *in, float *out, int batch, int rows, int cols,
                             ctcasr_stream_t stream);

/* ---- fp32 operands of the dense / input-projection GEMMs, split for the bf16 matrix pipe ---------
 * The GEMMs of asr/model.py:166-171 (dense), :203-214 (cuDNN RNN input projections) and their
 * gradients are fp32 x fp32 -> fp32.  gfx950 has no xf32 and its fp32 MFMA runs at 1/16 of the bf16
 * rate, so the host side may feed the library GEMM three-piece bfloat16 splits instead:
 *   x = x1 + x2 + x3,  x1 = rne_bf16(x), x2 = rne_bf16(x - x1), x3 = rne_bf16(x - x1 - x2)
 * (both differences are exact in fp32; 24 mantissa bits in all) and sum the six products of order
 * <= 2 (x1w1; x1w2 + x2w1; x1w3 + x2w2 + x3w1) with fp32 accumulation - every term down to 2^-24
 * relative, i.e. what an fp32 FMA chain keeps.
 *   x     f32 [rows, cols], row stride ld_x elements (cols % 8 == 0, ld_x % 4 == 0, 16-byte aligned)
 *   order host int[blocks], piece index (0, 1, 2) held by each block, blocks <= 6
 *   out   bf16, element (r, b, c) at r * ld_out + b * block_stride + c (both strides in elements,
 *         multiples of 8, 16-byte aligned base).  ld_out = blocks * cols, block_stride = cols is the
 *         K-concatenated form [rows, blocks, cols]: order {0,1,2,0,1,0} against a second operand
 *         split with {2,1,0,1,0,0} makes ONE bf16 GEMM over K = 6 * cols the whole product;
 *         ld_out = cols, block_stride = rows * cols stacks the blocks along the rows instead (for an
 *         operand whose K axis is its row axis); wider strides write a column range of a bigger
 *         matrix. */
#define CTCASR_SPLIT_MAX_BLOCKS 6
int ctcasr_split_bf16(const float *x, int64_t rows, int cols, int64_t ld_x, const int *order,
                      int blocks, void *out, int64_t ld_out, int64_t block_stride,
                      ctcasr_stream_t stream);

/* Forward projections of BOUNDED operands (activations behind a clipped ReLU, |h| <= 1 of an LSTM /
 * GRU / tanh cell, weights) also go in TWO fp16 pieces of x * scale - h1 = rne_f16(x scale),
 * h2 = rne_f16(x scale - h1), 11 + 11 mantissa bits - and three products h1 k1 + h1 k2 + h2 k1
 * (dropped: h2 k2 <= 2^-22): measured error = the fp32 GEMM's, at half the bf16 form's cost.
 *   scale  a power of two with |x| * scale < 65504 for every element (the caller knows the bound);
 *   order  host int[blocks] of 0 (h1) / 1 (h2); out / ld_out / block_stride as above, fp16.
 * The product of two such operands carries scale_x * scale_w: fold 1 / that into the GEMM's alpha. */
int ctcasr_split_f16(const float *x, int64_t rows, int cols, int64_t ld_x, float scale,
                     const int *order, int blocks, void *out, int64_t ld_out,
                     int64_t block_stride, ctcasr_stream_t stream);

/* Operands WITHOUT a bound (the gradients dxw: 1e-13 .. 1e-2 inside one matrix) take the same
 * two-piece form with a power-of-two scale per column or per row - along whichever axis is NOT the
 * product's K axis, so that the scale factors out of the sum: per column of dxw (per gate unit)
 * for the weight gradients dW = dxw^T x, per row (per frame) for the data gradient dx = dxw W.
 * The largest magnitude of a column / row lands in [2^13, 2^14); smaller elements keep 22 bits down
 * to 2^-18 of it, never less than 2^-40 of it in absolute terms - measured on the operands of real
 * training steps the weight gradients come out with the fp32 GEMM's error (1.03e-6 vs 1.07e-6 rms
 * relative, profiles/r03_gemm_bf16_split.md).
 *   ctcasr_colmax_scale   scale[c] = 2^(13 - exponent(max_r |x[r][c]|)) (1 for an all-zero column)
 *                         and inv_scale[c] = 1 / scale[c]; workspace: 4 * cols bytes
 *   ctcasr_split_f16_cols two fp16 pieces of x[r][c] * col_scale[c] * scale (layout as above)
 *   ctcasr_split_f16_rows the same with the scale of each ROW found on the fly (one workgroup per
 *                         row, cols <= 16384); inv_scale[r] = 1 / scale of row r
 *   ctcasr_rescale_rows   out[r][c] (+)= t[r][c] * row_factor[r] * alpha: takes the scales back out
 *                         of a product (cols % 4 == 0, 16-byte aligned) */
int ctcasr_colmax_scale(const float *x, int64_t rows, int cols, int64_t ld_x, void *workspace,
                        float *scale, float *inv_scale, ctcasr_stream_t stream);
/* (ABI v5) the same scales from column maxima already known: max_bits u32[cols] = bit patterns of
 * max_r |x[r][c]| as ctcasr_rnn_bwd_steps(..., colmax, ...) leaves them */
int ctcasr_colscale_from_max(const uint32_t *max_bits, int cols, float *scale, float *inv_scale,
                             ctcasr_stream_t stream);
int ctcasr_split_f16_cols(const float *x, int64_t rows, int cols, int64_t ld_x,
                          const float *col_scale, float scale, const int *order, int blocks,
                          void *out, int64_t ld_out, int64_t block_stride, ctcasr_stream_t stream);
int ctcasr_split_f16_rows(const float *x, int64_t rows, int cols, int64_t ld_x, const int *order,
                          int blocks, void *out, int64_t ld_out, int64_t block_stride,
                          float *inv_scale, ctcasr_stream_t stream);
int ctcasr_rescale_rows(const float *t, int64_t ld_t, const float *row_factor, float alpha,
                        float *out, int64_t ld_out, int64_t rows, int cols, int accumulate,
                        ctcasr_stream_t stream);

/* The same product with the split done in registers by an own kernel (csrc/split_gemm.hip): no
 * split pass, no K-concatenated copies, no inter-workgroup waits.
 *   C[M, N] (+)= A[M, K] . B[N, K]^T    all fp32, row-major with leading dimensions lda / ldb / ldc
 *   (elements); K % 16 == 0, lda % 4 == ldb % 4 == 0, A and B 16-byte aligned; accumulate != 0 adds
 *   to C.  Result: the six bf16 piece products of order <= 2 in fp32 accumulation, as above. */
int ctcasr_gemm_split_nt(const float *a, int64_t lda, const float *b, int64_t ldb, float *c,
                         int64_t ldc, int m, int n, int k, int accumulate, ctcasr_stream_t stream);
/*   C[M, N] (+)= A[K, M]^T . B[K, N]   (K = the ROW axis of both operands: the weight gradients
 *   dW = dxw^T x of asr/model.py:203-214's layers); any K, no alignment requirement. */
int ctcasr_gemm_split_tn(const float *a, int64_t lda, const float *b, int64_t ldb, float *c,
                         int64_t ldc, int m, int n, int k, int accumulate, ctcasr_stream_t stream);

/* ABI v6: the data gradient of a recurrent layer's input projection straight from what the fp16
 * backward recurrence published (csrc/dgrad16.hip) - the gradient of the cuDNN input projection,
 * asr/model.py:194-215, which TensorFlow computes as one cuBLAS GEMM over dxw:
 *   dx[T * B, n] (+)= dxw[T * B, 2 * 4H] . W_ih[2 * 4H, n]
 * where dxw is NOT read as fp32: a pass of ctcasr_rnn_bwd_steps with CTCASR_RNN_F16 (LSTM,
 * H = 1024, B <= 32, every row running all T steps) has left each step's dgates in the workspace's
 * exchange buffer as two fp16 pieces per value, scaled per (row, 64 gate columns) by a power of two,
 * in MFMA operand order, with the inverse scales beside them; the kernel multiplies those into the
 * packed fp16 pieces of W_ih with one fresh accumulator per 32-column stage that enters the fp32
 * total times the row's inverse scale (22 significand bits relative to a (row, 64-column block)
 * maximum; piece product h2 w2 dropped).  No library GEMM, no inter-workgroup waits.
 *   workspace   the recurrence workspace the backward pass ran in (same T, B as that pass);
 *               valid until the next persistent launch in that workspace
 *   packed      ctcasr_dgrad16_packed_bytes(n) bytes from ctcasr_dgrad16_pack_weights(W_ih [2 * 4H,
 *               n] with leading dimension ld_w, scale): fp16 pieces of W_ih * scale (saturating at
 *               +-60000: the caller keeps |w| * scale in range), K order and fragment order of the
 *               exchange buffer
 *   [t_lo, t_hi)    the time steps (rows t * B .. ) to produce;  [dir_lo, dir_hi) the directions
 *               whose gate columns enter the sum (0..2: both) - a finished direction / step range
 *               can be multiplied while the other is still running, the rest added later
 *               (accumulate != 0)
 * ctcasr_dgrad16_supported says whether (cell, T, B, H) qualifies.
 * ctcasr_dgrad16_published_offsets: byte offsets inside the recurrence workspace of the exchange
 * blocks (block 0 all-zero, block 1 + s = step s: [dir][producer][half][piece][k group][b][8 halves],
 * step s = time s of direction 0 and time T - 1 - s of direction 1) and of the inverse scales
 * ([step][dir][producer][32 rows] floats) of a pass over (T, B) - the layout contract between the
 * recurrence kernel and this one, exposed for tests and for other consumers of the pieces. */
size_t ctcasr_dgrad16_packed_bytes(int n);
int ctcasr_dgrad16_published_offsets(int T, int B, int hidden, size_t *exchange,
                                     size_t *inverse_scales);
int ctcasr_dgrad16_pack_weights(const float *w_ih, int64_t ld_w, int hidden, int n, float scale,
                                void *packed, ctcasr_stream_t stream);
int ctcasr_dgrad16_supported(int cell, int T, int B, int hidden);
int ctcasr_dgrad16_blockscaled(void *workspace, int T, int B, int hidden, const void *packed,
                               float scale, int n, float *dx, int64_t ld_dx, int t_lo, int t_hi,
                               int dir_lo, int dir_hi, int accumulate, ctcasr_stream_t stream);

/* ABI v6: the weight gradients of a recurrent layer's two matrices for one direction and step range
 * as one own kernel (csrc/wgrad16.hip; TensorFlow: cuDNN's RNN backward-weights, asr/model.py:194-215):
 *   dW_x[m, nx] += D^T X,  dW_y[m, ny] += D^T Y      (sums over the ROWS of D / X / Y)
 * from operands packed transposed - K = the row axis - as fp16 pieces in MFMA fragment order:
 * ctcasr_wgrad16_pack writes rows [row0, row0 + 32 * stages) of x [rows_total, cols] (leading
 * dimension ld_x; rows outside [0, rows_total) read as zeros - a negative / positive row0 shifts
 * an operand by time steps), each column times col_scale[c] (NULL: 1) times scale, saturating at
 * +-60000, into ctcasr_wgrad16_packed_bytes(stages, cols) bytes.  ctcasr_wgrad16_gemm multiplies
 * `stages` stages of the packed D (m columns, inverse column scales inv_scale[m]) into stages
 * [x_stage0, x_stage0 + stages) of the packed X (nx columns, fixed scale x_scale) and - optional -
 * of the packed Y, and ACCUMULATES into dW_x / dW_y (row-major, leading dimensions ld_*): two fp16
 * pieces per operand, three products, fp32 accumulation, no library GEMM.  Tiles of 256 x 256
 * outputs; `parts` >= 1 cuts every tile's row sum into that many workgroups (for launches whose
 * tiles alone do not fill the chip), which add to dW in part order through the words of `sync`
 * (ctcasr_wgrad16_sync_ints(m, nx, ny) int32, ZERO before the first launch that uses them; every
 * launch leaves them zero).  Part q adds when the tile's word reads q - part 0 too, so launches
 * of different streams that share the words take turns at a tile.  ABI v7: a part that gives up
 * waiting (bounded; ctcasr_set_option("wgrad16_spin_limit", polls) shortens the bound for tests)
 * sets word 0 and leaves WITHOUT adding and without passing the turn on; while word 0 is set
 * every launch on these words returns at once.  The caller folds word 0 into the optimizer's skip
 * flag (ctcasr_step_guard's wgrad_word), raises, and zeroes the words before the next launch.
 * sync may be NULL for parts == 1. */
size_t ctcasr_wgrad16_packed_bytes(int stages, int cols);
size_t ctcasr_wgrad16_sync_ints(int m, int nx, int ny);
int ctcasr_wgrad16_pack(const float *x, int64_t ld_x, int64_t rows_total, int cols, int64_t row0,
                        int stages, const float *col_scale, float scale, void *packed,
                        ctcasr_stream_t stream);
int ctcasr_wgrad16_gemm(const void *d_packed, int m, int stages, const float *inv_scale,
                        const void *x_packed, int x_stage0, int nx, float x_scale, float *dw_x,
                        int64_t ld_x, const void *y_packed, int y_stage0, int ny, float y_scale,
                        float *dw_y, int64_t ld_y, int parts, int32_t *sync, ctcasr_stream_t stream);

/* ---- K12: TensorFlow-form Adam over a flat parameter arena ------------------------------------
 * Replaces tf.train.AdamOptimizer(lr, beta1, beta2, epsilon).minimize (asr/model.py:80-83):
 *   lr_t = lr * sqrt(1 - beta2^step) / (1 - beta1^step);  m, v updated in place;
 *   param -= lr_t * m / (sqrt(v) + epsilon)     (epsilon OUTSIDE the bias correction).
 * `step` counts from 1.  grad is scaled by grad_scale first (1/world_size after a summed
 * all-reduce). */
int ctcasr_adam_step(float *param, const float *grad, float *m, float *v, int64_t n, float lr,
                     float beta1, float beta2, float epsilon, int64_t step, float grad_scale,
                     const int32_t *skip, ctcasr_stream_t stream);
/* ABI v5: `skip` (optional device word): the launch leaves param, m and v untouched when it is
 * non-zero - a step whose gradients are known to be invalid never reaches the parameters, without
 * the host having to look first.  ctcasr_step_guard fills it:
 *   skip[0] = any ctc_status[b] != 0 (tf.nn.ctc_loss would have raised, asr/model.py:259)
 *             | any per_utterance_loss[b] not finite | any persistent-recurrence time-out word
 *             (device pointers to the sticky words of up to two row blocks of the workspace,
 *             ctcasr_rnn_timeout_word_offset; either may be NULL)
 *             | *wgrad_word (ABI v7; may be NULL: word 0 of ctcasr_wgrad16_gemm's `sync`, raised
 *             by a part of a weight-gradient tile that gave up waiting for its turn - it and
 *             every later part then leave without adding);
 *   skip[1] = the time-out words or-ed together, bit 30 for the weight-gradient word (the host
 *             can poll this copy asynchronously - ctcasr_rnn_poll_error synchronises). */
int ctcasr_step_guard(const int32_t *ctc_status, const float *per_utterance_loss, int batch,
                      const uint32_t *timeout_word0, const uint32_t *timeout_word1,
                      const int32_t *wgrad_word, int32_t *skip, ctcasr_stream_t stream);
/* byte offset inside the recurrence workspace of the sticky time-out word of row block `block`
 * (0 .. (B - 1) / 32), or (size_t)-1 when (cell, T, B, H) runs the streaming kernels / there is
 * no such block */
size_t ctcasr_rnn_timeout_word_offset(int cell, int T, int B, int H, int block);
/* DIAGNOSTIC (ABI v5): `workgroups` (1..256) workgroups of 256 threads that hold their CUs for
 * `busy_us` (<= 100 ms) and touch no memory - the CU footprint of a collective's ring kernels, for
 * measuring on ONE GPU what such kernels cost beside the persistent recurrences (NCCL / RCCL
 * refuse two ranks on one device).  Not used by the training path. */
int ctcasr_occupy_cus(int workgroups, int busy_us, ctcasr_stream_t stream);
/* DIAGNOSTIC (ABI v6): the same footprint WITH a ring all-reduce's memory traffic - the workgroups
 * stream 2 x payload_bytes of a[i] += b[i] (16-byte elements, read twice and written once: 6 x
 * payload_bytes over L2 / fabric / HBM) through the scratch pair a, b (scratch_floats each, 16-byte
 * aligned), paced over busy_us: what a reduce-scatter + all-gather of the payload does to the
 * memory system beside the persistent recurrences and the library GEMMs.  Not used by training. */
int ctcasr_collective_traffic(int workgroups, int busy_us, float *a, const float *b,
                              int64_t scratch_floats, int64_t payload_bytes,
                              ctcasr_stream_t stream);
/* max_bits[0] = max(max_bits[0], bit pattern of max |x[i]|) (atomicMax; the caller zeroes it):
 * range guard of operands that go to the fp16 matrix pipe under a FIXED scale (the input
 * projections' weights, split_gemm.py) */
int ctcasr_absmax(const float *x, int64_t n, uint32_t *max_bits, ctcasr_stream_t stream);

/* ---- K1: feature extraction ----------------------------------------------------------------
 * Replaces python_speech_features.logfbank / mfcc / delta + the post-processing of load_sample
 * (asr/input_functions.py:156-335) for a batch of utterances.
 *   pcm            int16 [B, max_samples] raw PCM (NOT rescaled), rows zero padded
 *   num_samples    int32 [B] valid samples per row (>= 401 like the reference's check)
 *   feature_type   0 = 'mel' (80 log-mel), 1 = 'mfcc' (40 cepstra + 40 deltas)
 *   normalization  0 = 'none', 1 = 'local', 2 = 'local_scalar'
 *   tables         device block of ctcasr_features_tables_bytes(), filled once by
 *                  ctcasr_features_init_tables(tables, sampling_rate, stream) (synchronous)
 *   out            float32 [B, out_frames, 80], rows beyond an utterance's length are 0
 *   out_len        int32 [B] frames per utterance (after the optional every-2nd-frame drop)
 * ctcasr_features_num_frames(n) = 1 + ceil((n - 400) / 160) is the frame count of n samples. */
int ctcasr_features_num_frames(int num_samples);
size_t ctcasr_features_tables_bytes(void);
int ctcasr_features_init_tables(void *tables, int sampling_rate, ctcasr_stream_t stream);
size_t ctcasr_features_workspace_bytes(int B, int max_samples);
int ctcasr_features(const int16_t *pcm, const int32_t *num_samples, int B, int max_samples,
                    int feature_type, int normalization, int drop_every_second_frame,
                    const void *tables, float *out, int out_frames, int32_t *out_len,
                    void *workspace, size_t workspace_bytes, ctcasr_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* CTCASR_H_ */
