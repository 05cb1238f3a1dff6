/* ORACLE (test infrastructure, not product code): plain-C restatement of the CTC loss and the
 * CTC beam search of TensorFlow 1.12 as called by the reference:
 *   - tf.nn.ctc_loss(time_major=True, ctc_merge_repeated=True)       asr/model.py:259-264
 *   - tf.nn.ctc_beam_search_decoder(top_paths=1, merge_repeated=False) asr/model.py:292-296
 * Same algorithm as oracle/ctc.py (which is the readable statement and is cross-checked against
 * this file in tests/test_oracle_ctc.py); exists so that full-size cases (T'=500, beam 1024)
 * finish in seconds.  PARITY UNPINNED by the reference (it ships no vectors; TensorFlow cannot
 * run in the build container) - pinned by the recalled TensorFlow KATs in tests/golden.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library.
 * Build: gcc -O2 -shared -fPIC -o oracle/_build/libctc_ref.so oracle/c/ctc_ref.c -lm
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

static double lse2(double a, double b) {
    if (a == -INFINITY) return b;
    if (b == -INFINITY) return a;
    double hi = a > b ? a : b, lo = a > b ? b : a;
    return hi + log1p(exp(lo - hi));
}

/* loss[b] = -ln p(label_b | x_b); grad[T,B,C] w.r.t. logits (rows t >= seq_len[b] are zero).
 * status[b]: 0 ok, 1 = not enough time for the target transition sequence, 2 = bad label.
 * labels are concatenated, label_offsets[B+1]. */
int oracle_ctc_loss(const float *logits, int T, int B, int C, const int *labels,
                    const int *label_offsets, const int *seq_len, int blank, double *loss,
                    double *grad, int *status) {
    int max_s = 1;
    for (int b = 0; b < B; ++b) {
        int s = 2 * (label_offsets[b + 1] - label_offsets[b]) + 1;
        if (s > max_s) max_s = s;
    }
    double *logp = (double *)malloc(sizeof(double) * (size_t)T * C);
    double *alpha = (double *)malloc(sizeof(double) * (size_t)T * max_s);
    double *beta = (double *)malloc(sizeof(double) * (size_t)T * max_s);
    int *ext = (int *)malloc(sizeof(int) * max_s);
    double *occ = (double *)malloc(sizeof(double) * C);
    memset(grad, 0, sizeof(double) * (size_t)T * B * C);
    for (int b = 0; b < B; ++b) {
        const int *lab = labels + label_offsets[b];
        int L = label_offsets[b + 1] - label_offsets[b];
        int S = 2 * L + 1, len = seq_len[b];
        status[b] = 0;
        loss[b] = 0.0;
        int repeats = 0;
        for (int i = 0; i < L; ++i) {
            if (lab[i] < 0 || lab[i] >= C || lab[i] == blank) status[b] = 2;
            if (i > 0 && lab[i] == lab[i - 1]) ++repeats;
        }
        if (len > T || len < 0) status[b] = 2;
        if (status[b] == 0 && len < L + repeats) status[b] = 1;
        if (status[b] != 0) { loss[b] = INFINITY; continue; }
        for (int u = 0; u < S; ++u) ext[u] = (u & 1) ? lab[u >> 1] : blank;
        for (int t = 0; t < len; ++t) {
            const float *row = logits + ((size_t)t * B + b) * C;
            double mx = row[0], sum = 0.0;
            for (int c = 1; c < C; ++c) if (row[c] > mx) mx = row[c];
            for (int c = 0; c < C; ++c) sum += exp((double)row[c] - mx);
            double lz = mx + log(sum);
            for (int c = 0; c < C; ++c) logp[(size_t)t * C + c] = (double)row[c] - lz;
        }
        if (len == 0) {            /* only the empty label is feasible; p = 1 */
            continue;
        }
        for (int u = 0; u < S; ++u) alpha[u] = -INFINITY;
        alpha[0] = logp[blank];
        if (S > 1) alpha[1] = logp[ext[1]];
        for (int t = 1; t < len; ++t) {
            const double *prev = alpha + (size_t)(t - 1) * max_s;
            double *cur = alpha + (size_t)t * max_s;
            for (int u = 0; u < S; ++u) {
                double acc = prev[u];
                if (u >= 1) acc = lse2(acc, prev[u - 1]);
                if (u >= 2 && ext[u] != blank && ext[u] != ext[u - 2]) acc = lse2(acc, prev[u - 2]);
                cur[u] = acc == -INFINITY ? -INFINITY : acc + logp[(size_t)t * C + ext[u]];
            }
        }
        double *last = beta + (size_t)(len - 1) * max_s;
        for (int u = 0; u < S; ++u) last[u] = -INFINITY;
        last[S - 1] = 0.0;
        if (S > 1) last[S - 2] = 0.0;
        for (int t = len - 2; t >= 0; --t) {
            const double *nxt = beta + (size_t)(t + 1) * max_s;
            const double *lp = logp + (size_t)(t + 1) * C;
            double *cur = beta + (size_t)t * max_s;
            for (int u = 0; u < S; ++u) {
                double acc = nxt[u] + lp[ext[u]];
                if (u + 1 < S) acc = lse2(acc, nxt[u + 1] + lp[ext[u + 1]]);
                if (u + 2 < S && ext[u + 2] != blank && ext[u + 2] != ext[u])
                    acc = lse2(acc, nxt[u + 2] + lp[ext[u + 2]]);
                cur[u] = acc;
            }
        }
        const double *fin = alpha + (size_t)(len - 1) * max_s;
        double log_pzx = fin[S - 1];
        if (S > 1) log_pzx = lse2(log_pzx, fin[S - 2]);
        loss[b] = -log_pzx;
        for (int t = 0; t < len; ++t) {
            double *g = grad + ((size_t)t * B + b) * C;
            for (int c = 0; c < C; ++c) { occ[c] = -INFINITY; g[c] = exp(logp[(size_t)t * C + c]); }
            if (log_pzx == -INFINITY) continue;
            for (int u = 0; u < S; ++u)
                occ[ext[u]] = lse2(occ[ext[u]], alpha[(size_t)t * max_s + u] + beta[(size_t)t * max_s + u]);
            for (int c = 0; c < C; ++c) g[c] -= exp(occ[c] - log_pzx);
        }
    }
    free(logp); free(alpha); free(beta); free(ext); free(occ);
    return 0;
}

/* ---------------------------------------------------------------------------------------------
 * Beam search (float32 like TensorFlow).
 * ------------------------------------------------------------------------------------------- */
typedef struct {
    int parent, label;
    int first_child;          /* index of child for label 0, children are allocated as a block */
    float o_total, o_blank, o_label;
    float n_total, n_blank, n_label;
} node_t;

static float lse2f(float a, float b) {
    if (a == -INFINITY) return b;
    if (b == -INFINITY) return a;
    float hi = a > b ? a : b, lo = a > b ? b : a;
    return hi + log1pf(expf(lo - hi));
}

typedef struct { node_t *n; int count, cap; } pool_t;

/* worse(a, b): a ranks below b.  Lower total is worse; equal totals: the younger node (larger
 * index = created later) is worse. */
static int worse(const node_t *pool, int a, int b) {
    if (pool[a].n_total != pool[b].n_total) return pool[a].n_total < pool[b].n_total;
    return a > b;
}

/* binary min-heap of node indices, root = worst leaf */
static void heap_up(const node_t *pool, int *h, int i) {
    while (i > 0) {
        int p = (i - 1) / 2;
        if (!worse(pool, h[i], h[p])) break;
        int tmp = h[i]; h[i] = h[p]; h[p] = tmp; i = p;
    }
}
static void heap_down(const node_t *pool, int *h, int n, int i) {
    for (;;) {
        int l = 2 * i + 1, r = l + 1, m = i;
        if (l < n && worse(pool, h[l], h[m])) m = l;
        if (r < n && worse(pool, h[r], h[m])) m = r;
        if (m == i) break;
        int tmp = h[i]; h[i] = h[m]; h[m] = tmp; i = m;
    }
}

static const node_t *g_sort_pool;
static int cmp_desc(const void *a, const void *b) {
    int ia = *(const int *)a, ib = *(const int *)b;
    if (ia == ib) return 0;
    return worse(g_sort_pool, ia, ib) ? 1 : -1;
}

static int ensure_children(pool_t *p, int idx, int C) {
    if (p->n[idx].first_child >= 0) return 0;
    if (p->count + C > p->cap) {
        int cap = p->cap * 2 + C;
        node_t *nn = (node_t *)realloc(p->n, sizeof(node_t) * (size_t)cap);
        if (!nn) return -1;
        p->n = nn; p->cap = cap;
    }
    p->n[idx].first_child = p->count;
    for (int c = 0; c < C; ++c) {
        node_t *ch = &p->n[p->count + c];
        ch->parent = idx; ch->label = c; ch->first_child = -1;
        ch->o_total = ch->o_blank = ch->o_label = -INFINITY;
        ch->n_total = ch->n_blank = ch->n_label = -INFINITY;
    }
    p->count += C;
    return 0;
}

/* norm_mode 0: subtract the frame maximum (TensorFlow 1.12); 1: full log-softmax (TF >= 1.14).
 * out[B, T] (row padded with 0), out_len[B], logp[B] = total of the winning prefix. */
int oracle_ctc_beam_decode(const float *logits, int T, int B, int C, const int *seq_len,
                           int beam_width, int blank, int norm_mode, int *out, int *out_len,
                           float *logp) {
    float *x = (float *)malloc(sizeof(float) * C);
    int *heap = (int *)malloc(sizeof(int) * (beam_width + 1));
    int *branches = (int *)malloc(sizeof(int) * (beam_width + 1));
    for (int b = 0; b < B; ++b) {
        pool_t pool;
        pool.cap = 1 + C * 64; pool.count = 1;
        pool.n = (node_t *)malloc(sizeof(node_t) * (size_t)pool.cap);
        node_t *root = &pool.n[0];
        root->parent = -1; root->label = -1; root->first_child = -1;
        root->o_total = root->o_blank = root->o_label = -INFINITY;
        root->n_total = 0.f; root->n_blank = 0.f; root->n_label = -INFINITY;
        int nleaves = 1; heap[0] = 0;
        int len = seq_len[b];
        for (int t = 0; t < len; ++t) {
            const float *row = logits + ((size_t)t * B + b) * C;
            float mx = row[0];
            for (int c = 1; c < C; ++c) if (row[c] > mx) mx = row[c];
            float off = mx;
            if (norm_mode == 1) {
                float s = 0.f;
                for (int c = 0; c < C; ++c) s += expf(row[c] - mx);
                off = mx + logf(s);
            }
            for (int c = 0; c < C; ++c) x[c] = row[c] - off;

            int nb = nleaves;
            memcpy(branches, heap, sizeof(int) * nb);
            g_sort_pool = pool.n;
            qsort(branches, nb, sizeof(int), cmp_desc);
            nleaves = 0;
            for (int i = 0; i < nb; ++i) {
                node_t *e = &pool.n[branches[i]];
                e->o_total = e->n_total; e->o_blank = e->n_blank; e->o_label = e->n_label;
            }
            for (int i = 0; i < nb; ++i) {
                node_t *e = &pool.n[branches[i]];
                if (e->parent >= 0) {
                    const node_t *par = &pool.n[e->parent];
                    if (par->n_total != -INFINITY) {
                        float prev = (e->label == par->label) ? par->o_blank : par->o_total;
                        e->n_label = lse2f(e->n_label, prev);
                    }
                    e->n_label += x[e->label];
                }
                e->n_blank = e->o_total + x[blank];
                e->n_total = lse2f(e->n_blank, e->n_label);
                heap[nleaves] = branches[i];
                heap_up(pool.n, heap, nleaves);
                ++nleaves;
            }
            for (int i = 0; i < nb; ++i) {
                int bi = branches[i];
                float ototal = pool.n[bi].o_total;
                if (ototal == -INFINITY) continue;
                if (!(nleaves < beam_width || ototal > pool.n[heap[0]].n_total)) continue;
                if (ensure_children(&pool, bi, C) != 0) return -1;
                for (int c = 0; c < C; ++c) {
                    if (c == blank) continue;
                    node_t *par = &pool.n[bi];
                    int ci = par->first_child + c;
                    node_t *ch = &pool.n[ci];
                    if (ch->n_total != -INFINITY) continue;   /* already in the beam */
                    float prev = (c == par->label) ? par->o_blank : par->o_total;
                    ch->n_blank = -INFINITY;
                    ch->n_label = x[c] + prev;
                    ch->n_total = ch->n_label;
                    int cand = ch->n_total != -INFINITY &&
                               (nleaves < beam_width || ch->n_total > pool.n[heap[0]].n_total);
                    if (cand) {
                        if (nleaves == beam_width) {
                            node_t *bot = &pool.n[heap[0]];
                            bot->n_total = bot->n_blank = bot->n_label = -INFINITY;
                            heap[0] = ci;
                            heap_down(pool.n, heap, nleaves, 0);
                        } else {
                            heap[nleaves] = ci;
                            heap_up(pool.n, heap, nleaves);
                            ++nleaves;
                        }
                    } else {
                        ch->o_total = ch->o_blank = ch->o_label = -INFINITY;
                        ch->n_total = ch->n_blank = ch->n_label = -INFINITY;
                    }
                }
            }
        }
        int best = heap[0];
        for (int i = 1; i < nleaves; ++i) if (worse(pool.n, best, heap[i])) best = heap[i];
        int n = 0;
        for (int i = best; pool.n[i].parent >= 0; i = pool.n[i].parent) ++n;
        out_len[b] = n;
        for (int i = 0; i < T; ++i) out[(size_t)b * T + i] = 0;
        int k = n;
        for (int i = best; pool.n[i].parent >= 0; i = pool.n[i].parent) out[(size_t)b * T + --k] = pool.n[i].label;
        logp[b] = pool.n[best].n_total;
        free(pool.n);
    }
    free(x); free(heap); free(branches);
    return 0;
}
