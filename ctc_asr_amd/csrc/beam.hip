// K10: CTC beam search on the GPU with the semantics of TensorFlow 1.12's CTCBeamSearchDecoder
// (top path, merge_repeated=False) as called by CTCModel.decode_fn (asr/model.py:292-296).
//
// One workgroup per utterance; the beam (<= 1024 leaves) lives in LDS.  Per frame:
//   1. all threads: normalise the frame, copy new -> old, update every leaf from its own and its
//      parent's old probabilities (TensorFlow's first loop is embarrassingly parallel);
//   2. all threads: a bitonic sort of the leaves by OLD total (descending: TensorFlow's branch
//      order);
//   3. wave 0: TensorFlow's second loop, literally - branches in order, children in symbol order,
//      a bounded set of the W best leaves whose bottom is replaced when a child beats it
//      (`beam_expand`: the set's keys live in registers, the bottom is a wave-wide minimum).
// The serial part is kept because TensorFlow's result is order dependent in one corner: a leaf
// that is pushed out of the heap and then re-proposed (and rejected) by its still-active parent
// before its own turn has its old probabilities wiped and does not expand in this frame.  That
// is tracked with per-leaf eviction / expansion event numbers instead of mutable tree state.
// The prefix tree (parent, label, children table, beam slot per node) lives in HBM so that a
// prefix that leaves the beam and re-enters later is the same node, like TensorFlow's BeamEntry.
// Ties in total are broken towards the older tree node (TensorFlow leaves them to gtl::TopN /
// std::sort); node ages can differ from the CPU oracle's, so exact ties are not a parity case.
#include "common.h"

#define BEAM_THREADS 256
#define BEAM_MAX_WIDTH 1024
#define BEAM_SLOTS (2 * BEAM_MAX_WIDTH)   // live leaves + leaves evicted but still due to expand
#define BEAM_MAX_CLASSES 64

namespace {

__device__ __forceinline__ float lse2f(float a, float b) {
    if (a == -INFINITY) return b;
    if (b == -INFINITY) return a;
    const float hi = fmaxf(a, b), lo = fminf(a, b);
    return hi + log1pf(expf(lo - hi));
}
__device__ __forceinline__ unsigned order_key(float v) {
    const unsigned u = __float_as_uint(v);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ int gload(const int *p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ void gstore(int *p, int v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

struct BeamLds {
    // per slot
    int node[BEAM_SLOTS], label[BEAM_SLOTS], parent[BEAM_SLOTS], pslot[BEAM_SLOTS];
    float o_total[BEAM_SLOTS], o_blank[BEAM_SLOTS];
    float n_total[BEAM_SLOTS], n_blank[BEAM_SLOTS], n_label[BEAM_SLOTS];
    unsigned long long kids_in_beam[BEAM_SLOTS];   // bit c: child c of this leaf is in the heap
    short alive[BEAM_SLOTS], was_alive[BEAM_SLOTS], expanded[BEAM_SLOTS];
    int bidx[BEAM_SLOTS], evict_time[BEAM_SLOTS];
    // per beam position
    int heap[BEAM_MAX_WIDTH], branches[BEAM_MAX_WIDTH];
    unsigned long long sort_a[BEAM_MAX_WIDTH];
    int freelist[BEAM_SLOTS];
    float x[BEAM_MAX_CLASSES];
    int misc[8];   // 0 nheap, 1 node count, 2 nfree
};

// a ranks below b in the heap order (lower total; equal totals: the younger node)
__device__ __forceinline__ bool worse(const BeamLds &L, int a, int b) {
    if (L.n_total[a] != L.n_total[b]) return L.n_total[a] < L.n_total[b];
    return L.node[a] > L.node[b];
}

// heap-order key of a slot: ascending = worse first (lower total; equal totals: the younger node)
__device__ __forceinline__ unsigned long long heap_key(float total, int node, int slot) {
    const unsigned nd = min((unsigned)node, 0x1fffffu);
    return ((unsigned long long)order_key(total) << 32) | ((0x1fffffu - nd) << 11) | (unsigned)slot;
}
__device__ __forceinline__ float key_total(unsigned long long key) {
    const unsigned k = (unsigned)(key >> 32);
    return __uint_as_float((k & 0x80000000u) ? (k ^ 0x80000000u) : ~k);
}
// wave-wide minimum of a u32 by DPP: four shifts within the rows of 16 lanes, then the row
// broadcasts (lane 63 ends up with the minimum of all 64 lanes); a few VALU instructions instead
// of six LDS-crossbar shuffles
__device__ __forceinline__ unsigned wave_min_u32(unsigned v) {
#define BEAM_DPP_MIN(ctrl, row_mask)                                                          \
    v = min(v, (unsigned)__builtin_amdgcn_update_dpp((int)v, (int)v, ctrl, row_mask, 0xf, false))
    BEAM_DPP_MIN(0x111, 0xf);    // row_shr:1
    BEAM_DPP_MIN(0x112, 0xf);    // row_shr:2
    BEAM_DPP_MIN(0x114, 0xf);    // row_shr:4
    BEAM_DPP_MIN(0x118, 0xf);    // row_shr:8   -> lane 15 of every row holds the row's minimum
    BEAM_DPP_MIN(0x142, 0xa);    // row_bcast:15 into rows 1 and 3
    BEAM_DPP_MIN(0x143, 0xc);    // row_bcast:31 into rows 2 and 3
#undef BEAM_DPP_MIN
    return (unsigned)__builtin_amdgcn_readlane((int)v, 63);
}
__device__ __forceinline__ unsigned long long wave_min_u64(unsigned long long v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const unsigned hi = __shfl_xor((unsigned)(v >> 32), off, 64);
        const unsigned lo = __shfl_xor((unsigned)v, off, 64);
        const unsigned long long o = ((unsigned long long)hi << 32) | lo;
        v = o < v ? o : v;
    }
    return v;
}

// ascending bitonic sort of n 64-bit keys (n padded to pow2 with ~0ull), all threads
__device__ void bitonic_sort(unsigned long long *keys, int n_pow2, int tid) {
    for (int k = 2; k <= n_pow2; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = tid; i < n_pow2; i += BEAM_THREADS) {
                const int partner = i ^ j;
                if (partner > i) {
                    const unsigned long long a = keys[i], b = keys[partner];
                    const bool up = (i & k) == 0;
                    if ((a > b) == up) { keys[i] = b; keys[partner] = a; }
                }
            }
            __syncthreads();
        }
    }
}

// TensorFlow's second loop for one frame, run by ONE wave: branches in order, children in symbol
// order, a bounded set of the W best leaves whose worst member is replaced when a child beats it.
// TensorFlow keeps that set in a min-heap; only its bottom is ever looked at, so here the set is
// unordered - position p = k * 64 + lane of L.heap[] belongs to `lane`, which keeps the member's
// heap key and what an eviction needs to know about it (parent slot, label, "was in the beam
// when the frame began") in registers k - and the bottom is a wave-wide minimum of the keys,
// recomputed after every insertion.  One wave runs a long dependent instruction stream here, so
// the insertion path avoids LDS round trips: values cross lanes with v_readlane (the lane index
// is a ballot result), the read-modify-writes of the children masks are LDS atomics without
// return, an evicted transient hands its slot straight to the child that pushed it out, the next
// free slot and the next branch's children row (tree nodes, global memory) are fetched one
// step ahead.  A serial sift through an LDS heap cost ~2 us per insertion; this ~0.4.
template <int PER>
__device__ void beam_expand(BeamLds &L, int lane, int W, int C, int blank, int nslots, int nheap0,
                            int *pool_parent, int *pool_label, int *pool_children,
                            int nodes_per_utt) {
    int nheap = nheap0, nfree = 0, nodes = L.misc[1];
    for (int base = 0; base < nslots; base += 64) {      // free slots, ordered
        const bool is_free = base + lane < nslots && !L.alive[base + lane];
        const unsigned long long m = __ballot(is_free);
        if (is_free)
            L.freelist[nfree + __popcll(m & ((1ull << lane) - 1ull))] = base + lane;
        nfree += __popcll(m);
    }
    unsigned long long hk[PER];
    unsigned hm[PER];        // parent slot (0xfff: none) | label << 12 | was_alive << 18
#pragma unroll
    for (int k = 0; k < PER; ++k) {
        const int pos = k * 64 + lane;
        hk[k] = ~0ull;
        hm[k] = 0;
        if (pos < nheap) {
            const int s = L.heap[pos];
            hk[k] = heap_key(L.n_total[s], L.node[s], s);
            hm[k] = ((unsigned)L.pslot[s] & 0xfffu) | (((unsigned)L.label[s] & 0x3fu) << 12) |
                    ((unsigned)(L.was_alive[s] != 0) << 18);
        }
    }
    // the worst leaf of the beam: key, owning lane (valid when nheap > 0)
    unsigned long long bkey = ~0ull;
    int owner = 0;
    auto find_bottom = [&]() {
        unsigned long long m = hk[0];
#pragma unroll
        for (int k = 1; k < PER; ++k) m = hk[k] < m ? hk[k] : m;
        // the total's 32 bits decide almost always; exact ties go the long way
        const unsigned hi = (unsigned)(m >> 32), best = wave_min_u32(hi);
        unsigned long long tied = __ballot(hi == best);
        if (tied & (tied - 1)) {
            bkey = wave_min_u64(m);
            tied = __ballot(m == bkey);
        }
        owner = __ffsll((long long)tied) - 1;
        bkey = ((unsigned long long)best << 32) |
               (unsigned)__builtin_amdgcn_readlane((int)(unsigned)m, owner);
    };
    find_bottom();
    float btot = key_total(bkey);
    int free_top = nfree > 0 ? L.freelist[nfree - 1] : -1;      // next free slot, read ahead

    // the children rows (tree nodes of a branch's children) are read one branch ahead: the
    // global-memory round trip would otherwise sit in front of every branch's first insertion
    int kid_next = -1;
    if (nheap0 > 0 && lane < C)
        kid_next = gload(&pool_children[(size_t)L.node[L.branches[0]] * C + lane]);
    for (int j = 0; j < nheap0; ++j) {
        const int s = L.branches[j];
        const float ot = L.o_total[s];
        const int kidv = kid_next;           // tree nodes of this branch's children (-1: none yet)
        if (j + 1 < nheap0 && lane < C)
            kid_next = gload(&pool_children[(size_t)L.node[L.branches[j + 1]] * C + lane]);
        // branches come in descending old total and the bottom only rises: once a branch cannot
        // beat the bottom of a full beam, no later one can
        if (nheap == W && !(ot > btot)) break;
        if (!L.alive[s]) {
            // pushed out earlier in this frame: wiped if its parent re-proposed it since
            const int ps = L.pslot[s];
            if (ps >= 0 && L.expanded[ps] && L.evict_time[s] < L.bidx[ps] * 64 + L.label[s])
                continue;
        }
        if (!(ot > -INFINITY)) continue;
        if (lane == 0) L.expanded[s] = 1;
        const int slabel = L.label[s];
        const unsigned long long active = L.kids_in_beam[s];
        float v = -INFINITY;
        if (lane < C && lane != blank && !((active >> lane) & 1ull))
            v = L.x[lane] + (lane == slabel ? L.o_blank[s] : ot);
        // children that could enter right now; the bottom only rises while we insert
        unsigned long long cand = __ballot(v > -INFINITY && (nheap < W || v > btot));
        if (!cand) continue;
        const int pnode = L.node[s];
        while (cand) {
            const int c = __ffsll((long long)cand) - 1;
            cand &= cand - 1;
            const float vc = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), c));
            if (!(nheap < W || vc > btot)) continue;
            if (nodes + 1 >= nodes_per_utt) {     // tree pool exhausted
                if (lane == 0) L.misc[3] = 1;
                cand = 0ull;
                continue;
            }
            const bool full = nheap == W;
            int ns = -1;
            if (full) {
                // the bottom leaves the beam; a transient (a child that entered in this frame)
                // hands its slot straight to the child that pushes it out
                unsigned meta = 0;
#pragma unroll
                for (int k = 0; k < PER; ++k) meta = hk[k] == bkey ? hm[k] : meta;
                meta = (unsigned)__builtin_amdgcn_readlane((int)meta, owner);
                const int ev = (int)(bkey & 0x7ffu);
                if (lane == 0) {
                    L.alive[ev] = 0;
                    L.evict_time[ev] = j * 64 + c;
                    if ((meta & 0xfffu) != 0xfffu)
                        atomicAnd(&L.kids_in_beam[meta & 0xfffu], ~(1ull << ((meta >> 12) & 0x3fu)));
                }
                if (!((meta >> 18) & 1u)) ns = ev;
            }
            if (ns < 0) {
                ns = free_top;
                --nfree;
                free_top = nfree > 0 ? L.freelist[nfree - 1] : -1;
            }
            // node of child (s, c): reuse or create
            int kid = __builtin_amdgcn_readlane(kidv, c);
            if (kid < 0) {
                kid = nodes++;
                if (lane == 0) {
                    gstore(&pool_children[(size_t)pnode * C + c], kid);
                    gstore(&pool_parent[kid], pnode);
                    gstore(&pool_label[kid], c);
                }
                if (lane < C) gstore(&pool_children[(size_t)kid * C + lane], -1);
            }
            if (lane == 0) {
                L.node[ns] = kid; L.label[ns] = c; L.parent[ns] = pnode; L.pslot[ns] = s;
                L.n_total[ns] = vc; L.n_label[ns] = vc; L.n_blank[ns] = -INFINITY;
                L.alive[ns] = 1; L.was_alive[ns] = 0; L.expanded[ns] = 0;
                L.kids_in_beam[ns] = 0ull;
                atomicOr(&L.kids_in_beam[s], 1ull << c);
            }
            const unsigned long long nk = heap_key(vc, kid, ns);
            const unsigned nm = ((unsigned)s & 0xfffu) | ((unsigned)c << 12);
            if (full) {
#pragma unroll
                for (int k = 0; k < PER; ++k)
                    if (hk[k] == bkey) { hk[k] = nk; hm[k] = nm; L.heap[k * 64 + lane] = ns; }
            } else {
                if (lane == (nheap & 63)) {
#pragma unroll
                    for (int k = 0; k < PER; ++k)
                        if (k == (nheap >> 6)) { hk[k] = nk; hm[k] = nm; }
                    L.heap[nheap] = ns;
                }
                ++nheap;
            }
            find_bottom();
            btot = key_total(bkey);
        }
    }
    if (lane == 0) { L.misc[0] = nheap; L.misc[1] = nodes; }
}

__global__ void __launch_bounds__(BEAM_THREADS)
beam_decode_kernel(const float *__restrict__ logits, const int *__restrict__ seq_len, int T, int B,
                   int C, int blank, int W, int norm_mode, int *__restrict__ out,
                   int *__restrict__ out_len, float *__restrict__ logp, int *pool_parent,
                   int *pool_label, int *pool_slot, int *pool_children, int nodes_per_utt) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    BeamLds &L = *reinterpret_cast<BeamLds *>(smem);
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
    int len = seq_len[b];
    len = len < 0 ? 0 : (len > T ? T : len);
    pool_parent += (size_t)b * nodes_per_utt;
    pool_label += (size_t)b * nodes_per_utt;
    pool_slot += (size_t)b * nodes_per_utt;
    pool_children += (size_t)b * nodes_per_utt * C;
    const int nslots = 2 * W;

    for (int s = tid; s < nslots; s += BEAM_THREADS) { L.alive[s] = 0; L.was_alive[s] = 0; }
    for (int c = tid; c < C; c += BEAM_THREADS) gstore(&pool_children[c], -1);
    for (int i = tid; i < T; i += BEAM_THREADS) out[(size_t)b * T + i] = 0;
    __syncthreads();
    if (tid == 0) {
        L.node[0] = 0; L.label[0] = -1; L.parent[0] = -1; L.alive[0] = 1;
        L.n_total[0] = 0.f; L.n_blank[0] = 0.f; L.n_label[0] = -INFINITY;
        gstore(&pool_parent[0], -1); gstore(&pool_label[0], -1); gstore(&pool_slot[0], 0);
        L.heap[0] = 0;
        L.misc[0] = 1; L.misc[1] = 1; L.misc[3] = 0;
    }
    __syncthreads();

    for (int t = 0; t < len; ++t) {
        const int nheap0 = L.misc[0];
        // ---- 1. frame, old <- new, parent slots, children-in-beam masks -------------------------
        if (tid < 64) {
            const float *row = logits + ((size_t)t * B + b) * C;
            const float v = tid < C ? row[tid] : -INFINITY;
            const float mx = wave_max(v);
            float off = mx;
            if (norm_mode == 1) off = mx + logf(wave_sum(tid < C ? expf(v - mx) : 0.f));
            if (tid < C) L.x[tid] = v - off;
        }
        for (int s = tid; s < nslots; s += BEAM_THREADS) {
            L.kids_in_beam[s] = 0ull;
            L.was_alive[s] = L.alive[s];
            L.expanded[s] = 0;
            L.evict_time[s] = 0x7fffffff;
        }
        __syncthreads();
        for (int i = tid; i < nheap0; i += BEAM_THREADS) {
            const int s = L.heap[i];
            L.o_total[s] = L.n_total[s];
            L.o_blank[s] = L.n_blank[s];
            int ps = -1;
            if (L.parent[s] >= 0) ps = gload(&pool_slot[L.parent[s]]);
            L.pslot[s] = ps;
            if (ps >= 0) atomicOr(&L.kids_in_beam[ps], 1ull << L.label[s]);
        }
        __syncthreads();
        // ---- update the leaves (TensorFlow's first loop) ------------------------------------------
        for (int i = tid; i < nheap0; i += BEAM_THREADS) {
            const int s = L.heap[i];
            float nl = L.n_label[s];
            const int lab = L.label[s];
            if (lab >= 0) {
                const int ps = L.pslot[s];
                if (ps >= 0) nl = lse2f(nl, lab == L.label[ps] ? L.o_blank[ps] : L.o_total[ps]);
                nl += L.x[lab];
            }
            const float nb = L.o_total[s] + L.x[blank];
            L.n_label[s] = nl;
            L.n_blank[s] = nb;
            L.n_total[s] = lse2f(nb, nl);
        }
        __syncthreads();
        // ---- 2. branch order (old total desc, older node first) -------------------------------
        int n_pow2 = 1;
        while (n_pow2 < nheap0) n_pow2 <<= 1;
        for (int i = tid; i < n_pow2; i += BEAM_THREADS) {
            if (i < nheap0) {
                const int s = L.heap[i];
                const unsigned nd = min((unsigned)L.node[s], 0x1fffffu);
                // descending old total == ascending ~key; ties: older (smaller id) node first
                L.sort_a[i] = ((unsigned long long)(~order_key(L.o_total[s])) << 32) |
                              (nd << 11) | (unsigned)s;
            } else {
                L.sort_a[i] = ~0ull;
            }
        }
        __syncthreads();
        bitonic_sort(L.sort_a, n_pow2, tid);
        for (int i = tid; i < nheap0; i += BEAM_THREADS) {
            const int sa = (int)(L.sort_a[i] & 0x7ffu);
            L.branches[i] = sa;
            L.bidx[sa] = i;
        }
        __syncthreads();

        // ---- 3. TensorFlow's second loop, serial over branches, on wave 0 ------------------------
        if (tid < 64) {
            const int per = (W + 63) / 64;
            if (per <= 1)
                beam_expand<1>(L, lane, W, C, blank, nslots, nheap0, pool_parent, pool_label,
                               pool_children, nodes_per_utt);
            else if (per <= 4)
                beam_expand<4>(L, lane, W, C, blank, nslots, nheap0, pool_parent, pool_label,
                               pool_children, nodes_per_utt);
            else
                beam_expand<16>(L, lane, W, C, blank, nslots, nheap0, pool_parent, pool_label,
                                pool_children, nodes_per_utt);
        }
        __syncthreads();
        // ---- beam slots of the tree nodes for the next frame -----------------------------------
        for (int s = tid; s < nslots; s += BEAM_THREADS) {
            if (L.alive[s]) gstore(&pool_slot[L.node[s]], s);
            else if (L.was_alive[s]) gstore(&pool_slot[L.node[s]], -1);
        }
        __syncthreads();
    }

    // ---- best leaf and its label path --------------------------------------------------------------
    if (tid == 0) {
        const int nheap = L.misc[0];
        int best = L.heap[0];
        for (int i = 1; i < nheap; ++i)
            if (worse(L, best, L.heap[i])) best = L.heap[i];
        int n = 0;
        for (int node = L.node[best]; gload(&pool_parent[node]) >= 0; node = gload(&pool_parent[node]))
            ++n;
        out_len[b] = L.misc[3] ? -1 : n;      // -1: the prefix-tree pool overflowed
        if (logp) logp[b] = L.n_total[best];
        int k = n;
        for (int node = L.node[best]; gload(&pool_parent[node]) >= 0; node = gload(&pool_parent[node]))
            out[(size_t)b * T + --k] = gload(&pool_label[node]);
    }
}

// Tree nodes per utterance.  Every insertion of a child that has no node yet creates one, and
// a frame can insert up to W * (C - 1) children (each branch proposes all its symbols, each
// pushing the previous bottom out): T * W * (C - 1) + 1 is the true bound - 14 M nodes of 128 B
// at width 1024, T' = 500.  Measured on noise logits, the worst case for churn, a frame inserts
// 0.8-1.5 W children; the pool holds 4 W per frame plus 64 K (never more than the true bound, so
// small searches cannot exhaust it, and at most 2^21, the id field of the sort key).  Running
// out is reported (out_len = -1 -> CtcAsrError), never silent.
size_t nodes_per_utt(int T, int W, int C) {
    size_t n = (size_t)4 * W * T + 65536;
    const size_t bound = (size_t)T * W * (C > 1 ? C - 1 : 1) + 2;
    n = n < bound ? n : bound;
    return n < (1u << 21) ? n : (1u << 21);
}

}  // namespace

extern "C" size_t ctcasr_ctc_beam_workspace_bytes(int T, int B, int C, int beam_width) {
    if (T <= 0 || B <= 0 || C <= 0 || beam_width <= 0) return 0;
    return (size_t)B * nodes_per_utt(T, beam_width, C) * (3 + (size_t)C) * sizeof(int) + 256;
}

extern "C" int ctcasr_ctc_beam_decode(const float *logits, const int32_t *seq_len, int T, int B,
                                      int C, int blank, int beam_width, int norm_mode,
                                      int32_t *out, int32_t *out_len, float *logp, void *workspace,
                                      size_t workspace_bytes, ctcasr_stream_t stream) {
    if (!logits || !seq_len || !out || !out_len || T <= 0 || B <= 0 || C <= 1 || blank < 0 ||
        blank >= C || beam_width <= 0 || norm_mode < 0 || norm_mode > 1)
        return CTCASR_ERR_BAD_ARGUMENT;
    if (beam_width > BEAM_MAX_WIDTH || C > BEAM_MAX_CLASSES) return CTCASR_ERR_UNSUPPORTED;
    if (!workspace || workspace_bytes < ctcasr_ctc_beam_workspace_bytes(T, B, C, beam_width))
        return CTCASR_ERR_WORKSPACE;
    const size_t n = nodes_per_utt(T, beam_width, C);
    int *pool_parent = reinterpret_cast<int *>(workspace);
    int *pool_label = pool_parent + (size_t)B * n;
    int *pool_slot = pool_label + (size_t)B * n;
    int *pool_children = pool_slot + (size_t)B * n;
    if (hipFuncSetAttribute(reinterpret_cast<const void *>(&beam_decode_kernel),
                            hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)sizeof(BeamLds)) != hipSuccess)
        return CTCASR_ERR_LAUNCH;
    beam_decode_kernel<<<B, BEAM_THREADS, sizeof(BeamLds), (hipStream_t)stream>>>(
        logits, seq_len, T, B, C, blank, beam_width, norm_mode, out, out_len, logp, pool_parent,
        pool_label, pool_slot, pool_children, (int)n);
    return ctcasr_launch_status();
}
