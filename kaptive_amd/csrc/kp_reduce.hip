// kp_reduce.hip -- the per-assembly reduction on the device: hit table finalisation, locus scoring, overlap cull,
// clustering / pieces, gene extraction + translation from the packed stream, and gene states.
//
// These kernels replace, for a whole batch at once, the Python/numba steps the reference runs per genome after its
// aligner returns (src/kaptive/serotyping/core.py:157-396; numba kernels src/kaptive/core/interval.py:595-751,
// src/kaptive/core/seq.py:612-741).  The per-assembly logic is in kp_reduce_core.h (shared with the test harness);
// this file decides who runs what: one 64-lane wave per assembly, lanes in parallel over hits for the quadratic parts
// (rank sorts) and over the kept list / codons for the rest, lane 0 for the short sequential tails.  Assemblies are
// independent, so a batch of N assemblies keeps N waves busy; no step needs more than one wave's worth of LDS.
#include <algorithm>

#include "kp_internal.h"
#include "kp_reduce_core.h"

namespace {

constexpr int KEPT_LDS = 2048;  // upper bound of kept hits per assembly the cull scratch can hold
constexpr int SORT_LDS = 4096;  // sort keys staged in LDS per assembly (more hits than this fall back to global reads)

// ---- 1. band-task results -> per-assembly raw hit lists ----------------------------------------------------------------
__global__ __launch_bounds__(256) void kp_hit_compact_kernel(KpBatchView b, const int32_t *__restrict__ gene_len,
                                                             const KpTask *__restrict__ tasks,
                                                             const KpSwResult *__restrict__ results, const uint8_t *__restrict__ task_drop,
                                                             const uint32_t *__restrict__ task_count, uint32_t task_cap,
                                                             kp_hit *__restrict__ raw, uint32_t *__restrict__ n_raw,
                                                             uint32_t hit_cap, unsigned long long *__restrict__ cells) {
    const int cls = blockIdx.y;
    uint32_t n = task_count[cls];
    if (n > task_cap) n = task_cap;
    unsigned long long my_cells = 0;
    const int lane = threadIdx.x & 63;
    // whole waves iterate together: appends are aggregated per assembly within the wave (tasks of an assembly are
    // neighbours in the list, a per-thread atomic would hit one counter 64 times in a row)
    for (uint32_t i0 = blockIdx.x * blockDim.x + (threadIdx.x & ~63u); i0 < n; i0 += gridDim.x * blockDim.x) {
        const uint32_t i = i0 + lane;
        KpTask t;
        t.asm_id = -1;
        KpSwResult r;
        r.score = 0;
        int qlen = 0;
        if (i < n) {
            t = tasks[(size_t)cls * task_cap + i];
            r = results[(size_t)cls * task_cap + i];
            qlen = gene_len[t.gs >> 1];
            if (t.n_anchors) my_cells += (unsigned long long)qlen * (unsigned)t.width;  // (0: rejected by the chaining)
        }
        const bool hit = i < n && r.score >= KP_MIN_DP_SCORE && !task_drop[(size_t)cls * task_cap + i];  // (dropped: a chain consumed it, kp_join.hip)
        uint32_t slot = 0;
        unsigned long long todo = __ballot(hit);
        while (todo) {
            const int leader = __builtin_ctzll(todo);
            const int asm_l = __shfl(t.asm_id, leader);
            const unsigned long long same = __ballot(hit && t.asm_id == asm_l) & todo;
            uint32_t base = 0;
            if (lane == leader) base = atomicAdd(&n_raw[asm_l], (uint32_t)__builtin_popcountll(same));
            base = __shfl(base, leader);
            if ((same >> lane) & 1ull) slot = base + (uint32_t)__builtin_popcountll(same & ((1ull << lane) - 1ull));
            todo &= ~same;
        }
        if (hit && slot < hit_cap) {
            const int32_t cs = b.ctg_start[b.asm_first_ctg[t.asm_id] + t.contig];
            raw[(size_t)t.asm_id * hit_cap + slot] =
                kp_make_hit(t.gs, t.contig, cs, qlen, r.score, r.q_start, r.q_end, r.t_start, r.t_end, r.matches, r.block_len,
                            t.n_anchors, t.chain_score);
        }
    }
    if (my_cells) atomicAdd(cells, my_cells);
}

// ---- 1b. joined paths (kp-align v4, kp_join.hip) -> raw hits: one thread per join, an atomic per hit (joins are rare) --------
__global__ __launch_bounds__(64) void kp_join_hits_kernel(KpBatchView b, const int32_t *__restrict__ gene_len, const KpJoin *__restrict__ joins,
                                                          const uint32_t *__restrict__ join_count, uint32_t join_cap,
                                                          kp_hit *__restrict__ raw, uint32_t *__restrict__ n_raw, uint32_t hit_cap,
                                                          unsigned long long *__restrict__ cells) {
    const int cls = blockIdx.y;
    uint32_t n = join_count[cls];
    if (n > join_cap) n = join_cap;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const KpJoin &J = joins[(size_t)cls * join_cap + i];
        const int qlen = gene_len[J.gs >> 1];
        unsigned long long rows = 0;  // (every piece over its own rows: kp_spec.h, JOINED FILL)
        for (int k = 0; k < J.n_pieces; ++k) rows += (unsigned long long)max(0, min(J.r1[k], qlen) - J.r0[k]);
        atomicAdd(cells, rows * (unsigned)J.width);
        for (int k = 1; k < J.n_pieces; ++k) {
            if (J.state[k] != 1) continue;
            const uint32_t slot = atomicAdd(&n_raw[J.asm_id], 1u);
            if (slot >= hit_cap) continue;
            const int32_t cs = b.ctg_start[b.asm_first_ctg[J.asm_id] + J.contig];
            const int32_t *r = J.res[k];
            raw[(size_t)J.asm_id * hit_cap + slot] =
                kp_make_hit(J.gs, J.contig, cs, qlen, (int)((uint32_t)r[7] | ((uint32_t)r[8] << KP_HIT_BONUS_SHIFT)), r[1], r[2], r[3], r[4],
                            r[5], r[6], J.n_anchors, J.chain_score);
        }
    }
}

// ---- 2. emission order, duplicates, mapq (kp_spec.h) -------------------------------------------------------------------
// One block per assembly: the hits sorted by their leading keys with a bitonic network in LDS, ties settled by the full
// order (a rank sort -- every thread counts the hits that precede its own -- for more than SORT_LDS hits), then the
// duplicate / mapq pass on neighbours of the sorted list (kp_same_span is an equivalence, so "equal to the last kept
// hit" is "equal to the predecessor") with a block prefix sum for the compaction, then the mapping qualities per gene.  `raw` is scratch once the ranks are
// known: the compacted list is built there and copied back.
#ifndef KP_SORT_THREADS
#define KP_SORT_THREADS 256
#endif
constexpr int SORT_THREADS = KP_SORT_THREADS;

__global__ __launch_bounds__(SORT_THREADS) void kp_hit_sort_kernel(kp_hit *__restrict__ raw, const uint32_t *__restrict__ n_raw,
                                                                   uint32_t hit_cap, uint64_t *__restrict__ keys,
                                                                   kp_hit *__restrict__ hits, uint32_t *__restrict__ n_hits,
                                                                   const float *__restrict__ ln_half,
                                                                   const float *__restrict__ ln_int) {
    const int a = blockIdx.x, tid = threadIdx.x;
    uint32_t n = n_raw[a];
    if (n > hit_cap) n = hit_cap;
    kp_hit *src = raw + (size_t)a * hit_cap;
    kp_hit *dst = hits + (size_t)a * hit_cap;
    uint64_t *k = keys + (size_t)a * hit_cap * 3;
    __shared__ uint64_t s_k0[SORT_LDS];  // leading key of every hit: almost every comparison is decided by it
    __shared__ uint16_t s_ix[SORT_LDS];
    __shared__ uint32_t s_scan[SORT_THREADS];
    for (uint32_t i = tid; i < n; i += SORT_THREADS) {
        kp_hit_keys(src[i], k + 3 * (size_t)i);
        if (i < SORT_LDS) s_k0[i] = k[3 * (size_t)i];
    }
    __syncthreads();  // both branches read keys (LDS and global) that other waves of the block wrote
    if (n <= SORT_LDS) {
        // Up to SORT_LDS hits (every assembly but constructed ones): a bitonic network in LDS on (leading key, index) pairs --
        // the index makes the pairs distinct and the result the stable order by the leading key -- then every run of equal
        // leading keys (rare and short: hits of one gene with the same score on the same contig) is ranked by the full
        // order.  n log^2 n compare-exchanges; the rank sort this replaces (every hit against every other: n^2 pairs at eight
        // instructions each) was 0.2 ms of the whole chip's vector issue per 1000 assemblies of 1000 hits and 0.7 ms at 1800.
        uint32_t n2 = 64;
        while (n2 < n) n2 <<= 1;
        for (uint32_t i = tid; i < n2; i += SORT_THREADS) {
            if (i >= n) s_k0[i] = ~0ull;
            s_ix[i] = (uint16_t)i;
        }
        __syncthreads();
        for (uint32_t kk = 2; kk <= n2; kk <<= 1) {
            for (uint32_t j = kk >> 1; j >= 1; j >>= 1) {
                for (uint32_t t = tid; t < n2 / 2; t += SORT_THREADS) {
                    const uint32_t lo_i = ((t & ~(j - 1)) << 1) | (t & (j - 1)), hi_i = lo_i | j;  // the pair (i, i ^ j), i < i ^ j
                    const bool asc = (lo_i & kk) == 0;
                    const uint64_t a = s_k0[lo_i], c = s_k0[hi_i];
                    const uint16_t ia = s_ix[lo_i], ic = s_ix[hi_i];
                    const bool a_first = a < c || (a == c && ia < ic);
                    if (a_first != asc) { s_k0[lo_i] = c; s_k0[hi_i] = a; s_ix[lo_i] = ic; s_ix[hi_i] = ia; }
                }
                __syncthreads();
            }
        }
        for (uint32_t p = tid; p < n; p += SORT_THREADS) {
            const uint64_t m0 = s_k0[p];
            const uint32_t i = s_ix[p];
            uint32_t a = p, e = p + 1;  // the run of equal leading keys around p
            while (a > 0 && s_k0[a - 1] == m0) --a;
            while (e < n && s_k0[e] == m0) ++e;
            uint32_t rank = 0;
            if (e - a > 1) {
                const uint64_t mine[3] = {m0, k[3 * (size_t)i + 1], k[3 * (size_t)i + 2]};
                const uint32_t seeds_mine = kp_hit_seeds_key(src[i]);
                for (uint32_t q = a; q < e; ++q) {
                    const uint32_t j = s_ix[q];
                    if (j != i) rank += kp_keys_less(k + 3 * (size_t)j, kp_hit_seeds_key(src[j]), j, mine, seeds_mine, i) ? 1u : 0u;
                }
            }
            dst[a + rank] = src[i];
        }
    } else
    for (uint32_t i = tid; i < n; i += SORT_THREADS) {  // more hits than LDS holds keys for: rank sort, every hit against every other
        const uint64_t m0 = k[3 * (size_t)i];
        uint32_t rank = 0, equal = 0;
        uint32_t j = 0;
        for (; j + 8 <= (uint32_t)SORT_LDS; j += 8) {  // eight broadcast reads in flight, then the compares
            uint64_t o[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) o[u] = s_k0[j + u];
#pragma unroll
            for (int u = 0; u < 8; ++u) { rank += o[u] < m0 ? 1u : 0u; equal += o[u] == m0 ? 1u : 0u; }
        }
        for (; j < n; ++j) { const uint64_t other = k[3 * (size_t)j]; rank += other < m0 ? 1u : 0u; equal += other == m0 ? 1u : 0u; }
        if (equal > 1) {  // (one is the hit itself) ties on the leading key: settled by the full order
            const uint64_t mine[3] = {m0, k[3 * (size_t)i + 1], k[3 * (size_t)i + 2]};
            for (j = 0; j < n; ++j) {
                const uint64_t other = j < SORT_LDS ? s_k0[j] : k[3 * (size_t)j];
                if (other == m0) rank += kp_keys_less(k + 3 * (size_t)j, kp_hit_seeds_key(src[j]), j, mine, kp_hit_seeds_key(src[i]), i) ? 1u : 0u;
            }
        }
        dst[rank] = src[i];
    }
    __syncthreads();
    // thread t owns the sorted positions [lo, hi); keep = not a duplicate of its predecessor
    const uint32_t per = (n + SORT_THREADS - 1) / SORT_THREADS;
    const uint32_t lo = min(n, (uint32_t)tid * per), hi = min(n, lo + per);
    uint32_t kept = 0;
    for (uint32_t i = lo; i < hi; ++i) kept += (i > 0 && kp_same_span(dst[i - 1], dst[i])) ? 0u : 1u;
    s_scan[tid] = kept;
    __syncthreads();
    for (int o = 1; o < SORT_THREADS; o <<= 1) {  // inclusive scan
        const uint32_t v = tid >= o ? s_scan[tid - o] : 0u;
        __syncthreads();
        s_scan[tid] += v;
        __syncthreads();
    }
    uint32_t m = s_scan[tid] - kept;  // first output slot of this thread
    for (uint32_t i = lo; i < hi; ++i) {
        if (i > 0 && kp_same_span(dst[i - 1], dst[i])) continue;
        src[m++] = dst[i];
    }
    const uint32_t total = s_scan[SORT_THREADS - 1];
    __syncthreads();
    for (uint32_t i = tid; i < total; i += SORT_THREADS) dst[i] = src[i];
    if (tid == 0) n_hits[a] = total;
    __syncthreads();
    // mapping qualities: every gene's run of the list is walked by the thread that holds its first hit (a run is a few
    // hits long: a gene, its fragments and its paralogues in one assembly); scratch = the sort keys, no longer needed
    int32_t *scratch = reinterpret_cast<int32_t *>(k);  // 6 ints per hit row available, 4 used
    for (uint32_t i = tid; i < total; i += SORT_THREADS) {
        if (i > 0 && dst[i - 1].gene == dst[i].gene) continue;
        uint32_t j = i + 1;
        while (j < total && dst[j].gene == dst[i].gene) ++j;
        kp_assign_mapq(dst + i, (int)(j - i), scratch + 4 * (size_t)i, scratch + 4 * (size_t)i + (j - i),
                       scratch + 4 * (size_t)i + 2 * (size_t)(j - i), scratch + 4 * (size_t)i + 3 * (size_t)(j - i), ln_half, ln_int);
    }
}

// ---- 3. locus scores (core.py:164-198) ---------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void kp_score_kernel(const kp_hit *__restrict__ hits, const uint32_t *__restrict__ n_hits,
                                                      uint32_t hit_cap, KpTypingDb db, double min_cov,
                                                      double *__restrict__ scores, int32_t *__restrict__ counts) {
    const int a = blockIdx.x;
    const kp_hit *h = hits + (size_t)a * hit_cap;
    const int n = (int)n_hits[a];
    for (int l = threadIdx.x; l < db.n_loci; l += 64)
        kp_locus_score(h, n, db, l, min_cov, &scores[(size_t)a * db.n_loci + l], &counts[(size_t)a * db.n_loci + l]);
}

// ---- 4. cull, clusters, pieces, translation ------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void kp_reduce_kernel(KpBatchView b, const kp_hit *__restrict__ hits,
                                                       const uint32_t *__restrict__ n_hits, uint32_t hit_cap,
                                                       KpTypingDb db, KpTypingParams prm, const int32_t *__restrict__ best,
                                                       uint64_t *__restrict__ keys, uint32_t *__restrict__ order,
                                                       uint8_t *__restrict__ kept_flag, KpKept *__restrict__ kept,
                                                       int kept_cap, KpPiece *__restrict__ pieces, int piece_cap,
                                                       KpAsmSummary *__restrict__ summary, uint8_t *__restrict__ prot,
                                                       int prot_cap, int32_t *__restrict__ pair_q_off,
                                                       int32_t *__restrict__ pair_q_len, int32_t *__restrict__ pair_t_off,
                                                       int32_t *__restrict__ pair_t_len, int32_t *__restrict__ n_pairs,
                                                       int32_t *__restrict__ pair_base) {
    // one 32 KB LDS block: first the cull keys (rank sort), afterwards the kept list of the greedy cull and the
    // permutation scratch of the clustering
    __shared__ uint64_t s_raw[SORT_LDS];
    static_assert(SORT_LDS * sizeof(uint64_t) == 4 * KEPT_LDS * sizeof(int32_t), "LDS block is shared");
    int32_t *s_ctg = reinterpret_cast<int32_t *>(s_raw), *s_s = s_ctg + KEPT_LDS, *s_e = s_s + KEPT_LDS,
            *s_perm = s_e + KEPT_LDS;
    __shared__ uint8_t s_codon[128];
    __shared__ int s_fail, s_base, s_total;
    const int a = blockIdx.x, lane = threadIdx.x;
    const int n = (int)n_hits[a];
    const kp_hit *h = hits + (size_t)a * hit_cap;
    uint64_t *k = keys + (size_t)a * hit_cap;
    uint32_t *ord = order + (size_t)a * hit_cap;
    uint8_t *flag = kept_flag + (size_t)a * hit_cap;
    KpKept *out = kept + (size_t)a * kept_cap;
    KpPiece *pc = pieces + (size_t)a * piece_cap;
    KpAsmSummary *sum = summary + a;
    const int best_locus = best[a];
    if (lane == 0) {
        kp_fill_codon_table(s_codon);
        KpAsmSummary z;
        z.n_hits = n; z.n_kept = 0; z.n_final = 0; z.n_pieces = 0; z.best_locus = best_locus;
        z.n_expected = 0; z.n_missing = 0; z.overflow = 0;
        for (int w = 0; w < KP_MAX_LOCUS_GENES / 64; ++w) z.missing_mask[w] = 0;
        z.ident_sum = 0.f; z.n_normal = 0;
        *sum = z;
        s_fail = 0;
    }
    // visit order of the cull
    for (int i = lane; i < n; i += 64) {
        k[i] = kp_cull_key(h[i], (int)db.gene_locus[h[i].gene] == best_locus, (uint32_t)i);
        if (i < SORT_LDS) s_raw[i] = k[i];
    }
    __syncthreads();
    if (n <= SORT_LDS) {
        // Keys are unique (the hit's index is their last field): a bitonic network over the LDS copy, (log n)^2 / 2 rounds of
        // n / 128 compare-exchanges per lane, and the sorted keys give the order.  (Counting, for every hit, the keys below
        // its own -- n / 64 x n compares per lane -- was a third of this kernel.)
        int np2 = 64;
        while (np2 < n) np2 <<= 1;
        for (int i = n + lane; i < np2; i += 64) s_raw[i] = ~0ull;
        __syncthreads();
        for (int kk = 2; kk <= np2; kk <<= 1)
            for (int j = kk >> 1; j > 0; j >>= 1) {
                for (int t = lane; t < (np2 >> 1); t += 64) {
                    const int lo = ((t & ~(j - 1)) << 1) | (t & (j - 1)), hi = lo | j;  // pair t of this round
                    const uint64_t x = s_raw[lo], y = s_raw[hi];
                    if ((x > y) == ((lo & kk) == 0)) { s_raw[lo] = y; s_raw[hi] = x; }
                }
                __syncthreads();
            }
        for (int r = lane; r < n; r += 64) ord[r] = (uint32_t)(s_raw[r] & 0x7FFFFull);  // kp_cull_key: the index is the low 19 bits
    } else
    for (int i = lane; i < n; i += 64) {  // more hits than the LDS copy holds: ranks by counting, against global memory
        const uint64_t mine = k[i];
        uint32_t rank = 0;
        const int n_lds = n < SORT_LDS ? n : SORT_LDS;
        int j = 0;
        for (; j + 8 <= n_lds; j += 8) {
            uint64_t o[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) o[u] = s_raw[j + u];
#pragma unroll
            for (int u = 0; u < 8; ++u) rank += o[u] < mine ? 1u : 0u;
        }
        for (; j < n_lds; ++j) rank += s_raw[j] < mine ? 1u : 0u;
        for (j = n_lds; j < n; ++j) rank += k[j] < mine ? 1u : 0u;
        ord[rank] = (uint32_t)i;
    }
    __syncthreads();
    // greedy cull, 64 candidates fetched at a time, each tested against the kept list by all lanes
    int nk = 0;
    const int cap = kept_cap < KEPT_LDS ? kept_cap : KEPT_LDS;
    bool fail = false;
    if (n < 2) {  // below two hits the reference returns the table unchanged (alignment.py:665-666)
        if (n == 1 && lane == 0) { flag[0] = 1; s_ctg[0] = h[0].contig; s_s[0] = h[0].t_start; s_e[0] = h[0].t_end; }
        nk = n;
    } else {
        // 64 candidates at a time, a lane each: first against everything kept so far (the list is read by all lanes at
        // once), then the survivors one after the other in visit order -- each one kept is tested by the lanes behind it.
        // (A candidate at a time, tested by all lanes against the kept list, spent a hundred instructions and a barrier
        // per hit on what is a few compares: 0.3 of this kernel's 1.2 ms.)
        for (int p0 = 0; p0 < n && !fail; p0 += 64) {
            const int mine = p0 + lane < n ? (int)ord[p0 + lane] : -1;
            int mc = 0, ms = 0, me = 0;
            if (mine >= 0) { mc = h[mine].contig; ms = h[mine].t_start; me = h[mine].t_end; }
            const int len = me - ms;
            auto clashes = [&](int c, int s, int e) {  // the reference's rule (interval.py:698-751): overlap > 10 % of the shorter
                const int ov = min(me, e) - max(ms, s);
                return c == mc && ov > 0 && (int64_t)ov * 10 > (int64_t)min(len, e - s);
            };
            bool alive = mine >= 0 && len > 0;
            for (int j = 0; j < nk; ++j) alive = alive && !clashes(s_ctg[j], s_s[j], s_e[j]);
            bool keep = false;
            unsigned long long pending = __ballot(alive);
            while (pending) {
                const int l = __builtin_ctzll(pending);  // the next survivor in visit order: kept
                if (nk >= cap) { fail = true; break; }
                const int c = __shfl(mc, l), s0 = __shfl(ms, l), e0 = __shfl(me, l);
                if (lane == l) { keep = true; s_ctg[nk] = c; s_s[nk] = s0; s_e[nk] = e0; }
                ++nk;
                if (lane > l && alive && clashes(c, s0, e0)) alive = false;
                pending = __ballot(alive && lane > l);
            }
            if (mine >= 0) flag[mine] = keep ? 1 : 0;
            __syncthreads();  // the kept list is complete for the next 64
        }
    }
    if (fail) {
        if (lane == 0) { sum->overflow |= 1; sum->n_kept = 0; pair_base[a] = 0; }
        return;
    }
    __syncthreads();
    // The kept records live in LDS while one lane clusters them (insertion sort, single linkage, pieces, inside flags,
    // missing genes: a few thousand dependent reads of their fields, otherwise at the latency of the L2s); the cull's
    // scratch is free by now.  More kept hits than fit there: in place, in global memory.
    const bool in_lds = (size_t)nk * sizeof(KpKept) <= 3 * (size_t)KEPT_LDS * sizeof(int32_t);
    KpKept *lk = in_lds ? reinterpret_cast<KpKept *>(s_raw) : out;
    // ... and beside them (behind the permutation scratch, in the same LDS block) what the database says about their genes
    // and the pieces' extents: both are read again and again by that one lane
    static_assert(KEPT_LDS >= 640 + (3 * KEPT_LDS * 4 / sizeof(KpKept)) * sizeof(KpGeneInfo) / 4 + 4, "LDS carve-up of the clustering");
    int32_t *piece_tmp = in_lds && piece_cap <= 40 ? s_perm + 512 : nullptr;
    KpGeneInfo *info = in_lds ? reinterpret_cast<KpGeneInfo *>(s_perm + 640) : nullptr;
    __threadfence_block();  // (lane 0's flags are read by every lane below)
    __syncthreads();        // (every lane is done with the cull's lists)
    // kept list in emission order: 64 hits per round, the kept ones of a round land behind those of the rounds before
    {
        const unsigned long long below = (1ull << lane) - 1ull;
        int m = 0;
        for (int i0 = 0; i0 < n; i0 += 64) {
            const int i = i0 + lane;
            const bool f = i < n && flag[i] != 0;
            const unsigned long long mask = __ballot(f);
            if (f) {
                KpKept o;
                o.gene = h[i].gene; o.contig = h[i].contig; o.q_start = h[i].q_start; o.q_end = h[i].q_end;
                o.t_start = h[i].t_start; o.t_end = h[i].t_end; o.score = h[i].score; o.strand = h[i].strand;
                o.prot_off = 0; o.prot_len = 0; o.cluster = 0; o.pident = 0.f; o.coverage = 0.f; o.state = 0; o.flags = 0;
                o.pad_ = 0;
                for (int x = 0; x < 8; ++x) o.dp[x] = 0;
                const int at = m + __builtin_popcountll(mask & below);
                lk[at] = o;
                if (info) info[at] = KpGeneInfo{db.gene_locus[o.gene], db.gene_pos[o.gene], db.gene_strand[o.gene], db.gene_extra[o.gene]};
            }
            m += __builtin_popcountll(mask);
        }
    }
    __threadfence_block();
    __syncthreads();
    // clusters, pieces, inside flags, missing genes, protein slots: short sequential tail
    if (lane == 0) {
        sum->n_kept = nk;
        kp_cluster_and_pieces(lk, nk, db, best_locus, prm.max_locus_length, s_perm, pc, piece_cap, sum, info, piece_tmp);
        int used = 0;
        for (int i = 0; i < nk; ++i) {
            const int frame = (3 - lk[i].q_start % 3) % 3, len = lk[i].t_end - lk[i].t_start;
            const int max_codons = len > frame ? (len - frame) / 3 : 0;
            if (used + max_codons > prot_cap) { sum->overflow |= 8; s_fail = 1; break; }
            lk[i].prot_off = used;
            lk[i].prot_len = max_codons;  // upper bound; the translation below shortens it at the first stop
            used += max_codons;
        }
        if (s_fail) sum->n_kept = 0;
        // this assembly's slots in the batch-wide list of protein pairs
        s_base = s_fail ? 0 : atomicAdd(n_pairs, nk);
        pair_base[a] = s_base;
    }
    __syncthreads();
    if (in_lds) {  // the records go to their place in global memory (also when the protein buffer overflowed: flags and clusters are valid)
        const uint32_t *src = reinterpret_cast<const uint32_t *>(lk);
        uint32_t *dst = reinterpret_cast<uint32_t *>(out);
        static_assert(sizeof(KpKept) % 4 == 0, "copied as words");
        for (int x = lane; x < nk * (int)(sizeof(KpKept) / 4); x += 64) dst[x] = src[x];
        __threadfence_block();
        __syncthreads();
    }
    if (s_fail) return;
    const size_t base = (size_t)s_base;
    // Translation: the codons of all kept hits as one list, one codon per lane and round, four rounds' loads in flight
    // (a hit at a time, round after round until its first stop, was a chain of dependent trips to memory).  Every codon of a hit's slot is written; the first stop codon of each hit is found with an LDS
    // minimum and becomes the protein's length -- what lies behind it in the slot is never read.
    const uint32_t *asm_words = b.words + b.asm_word_off[a];
    const int c0 = b.asm_first_ctg[a];
    const int r0 = b.asm_first_nrun[a], n_runs = b.asm_first_nrun[a + 1] - r0;
    const int32_t *runs = b.n_runs + 2 * (size_t)r0;
    uint8_t *pa = prot + (size_t)a * prot_cap;
    // per kept hit, in LDS (the cull's and the clustering's scratch is free again): slot offset * 4 + reading frame, first
    // stop so far, first base, last base + 1 (negative: minus strand) -- a codon then costs LDS reads and its own words only
    int32_t *s_off4 = s_ctg, *s_stop = s_s, *s_a0 = s_e, *s_a1 = s_perm;
    for (int i = lane; i < nk; i += 64) {
        const KpKept &o = out[i];
        const int32_t cs = b.ctg_start[c0 + o.contig];
        s_off4[i] = (o.prot_off << 2) | ((3 - o.q_start % 3) % 3);
        s_stop[i] = o.prot_len;
        s_a0[i] = cs + o.t_start;
        s_a1[i] = o.strand >= 0 ? cs + o.t_end : -(cs + o.t_end);
    }
    if (lane == 0) s_total = nk ? out[nk - 1].prot_off + out[nk - 1].prot_len : 0;
    __syncthreads();
    const int total = s_total;
    constexpr int TR = 4;
    for (int x0 = 0; x0 < total; x0 += 64 * TR) {
        int hit[TR], cod[TR];
        uint8_t aa[TR];
#pragma unroll
        for (int u = 0; u < TR; ++u) {
            const int x = x0 + 64 * u + lane;
            hit[u] = -1; cod[u] = 0; aa[u] = 0;
            if (x < total) {
                int lo = 0, hi = nk - 1;  // the last hit whose slot starts at or before x (empty slots share their start with the next)
                while (lo < hi) {
                    const int mid = (lo + hi + 1) >> 1;
                    if ((s_off4[mid] >> 2) <= x) lo = mid; else hi = mid - 1;
                }
                hit[u] = lo; cod[u] = x - (s_off4[lo] >> 2);
            }
        }
#pragma unroll
        for (int u = 0; u < TR; ++u)
            if (hit[u] >= 0) {
                const int32_t a1 = s_a1[hit[u]];
                aa[u] = kp_codon_aa(asm_words, runs, n_runs, s_a0[hit[u]], a1 < 0 ? -a1 : a1, a1 < 0 ? -1 : 1, s_off4[hit[u]] & 3,
                                    cod[u], s_codon);
            }
#pragma unroll
        for (int u = 0; u < TR; ++u)
            if (hit[u] >= 0) {
                pa[(s_off4[hit[u]] >> 2) + cod[u]] = aa[u];
                if (aa[u] == '*') atomicMin(&s_stop[hit[u]], cod[u]);
            }
    }
    __syncthreads();
    for (int i = lane; i < nk; i += 64) {
        const int first_stop = s_stop[i];
        out[i].prot_len = first_stop;
        const size_t slot = base + i;
        pair_q_off[slot] = (int32_t)((size_t)a * prot_cap + (s_off4[i] >> 2));  // offset into the batch protein buffer
        pair_q_len[slot] = first_stop;
        pair_t_off[slot] = db.prot_off[out[i].gene];
        pair_t_len[slot] = db.prot_len[out[i].gene];
    }
}

// ---- 6. identities, coverages, states (core.py:363-394) ------------------------------------------------------------------
__global__ __launch_bounds__(64) void kp_state_kernel(KpBatchView b, KpTypingDb db, KpTypingParams prm,
                                                      KpKept *__restrict__ kept, int kept_cap,
                                                      KpAsmSummary *__restrict__ summary, const int32_t *__restrict__ dp8,
                                                      const int32_t *__restrict__ pair_base) {
    const int a = blockIdx.x, lane = threadIdx.x;
    KpAsmSummary *sum = summary + a;
    const int nk = sum->n_kept;
    KpKept *out = kept + (size_t)a * kept_cap;
    const int c0 = b.asm_first_ctg[a];
    int alive = 0;
    for (int i = lane; i < nk; i += 64) {
        KpKept o = out[i];
        const int32_t *dp = dp8 + 8 * ((size_t)pair_base[a] + i);
        for (int x = 0; x < 8; ++x) o.dp[x] = dp[x];
        kp_gene_state(&o, db.gene_len[o.gene], b.ctg_len[c0 + o.contig], prm);
        out[i] = o;
        alive += (o.flags & KP_F_SPURIOUS) ? 0 : 1;
    }
    for (int o = 32; o >= 1; o >>= 1) alive += __shfl_xor(alive, o);
    if (lane == 0) sum->n_final = alive;
}

// the hits of one typing group: a contiguous run of the assembly's gene-sorted hit list
__global__ __launch_bounds__(256) void kp_hit_split_kernel(const kp_hit *__restrict__ hits, const uint32_t *__restrict__ n_hits,
                                                           uint32_t hit_cap, int32_t gene_lo, int32_t gene_hi,
                                                           kp_hit *__restrict__ out, uint32_t *__restrict__ out_n) {
    const int a = blockIdx.x;
    const kp_hit *h = hits + (size_t)a * hit_cap;
    const int n = (int)n_hits[a];
    const int first = kp_lower_bound_gene(h, n, gene_lo), last = kp_lower_bound_gene(h, n, gene_hi);
    kp_hit *o = out + (size_t)a * hit_cap;
    for (int i = first + (int)threadIdx.x; i < last; i += (int)blockDim.x) {
        kp_hit x = h[i];
        x.gene -= gene_lo;
        o[i - first] = x;
    }
    if (threadIdx.x == 0) out_n[a] = (uint32_t)(last - first);
}

// rows of `width` words from a matrix with row pitch src_pitch into one with row pitch dst_pitch (words beyond `width`
// of a destination row are zeroed)
__global__ __launch_bounds__(256) void kp_pack_rows_kernel(const uint32_t *__restrict__ src, size_t src_pitch,
                                                           uint32_t *__restrict__ dst, size_t dst_pitch, size_t width) {
    const uint32_t *s = src + blockIdx.x * src_pitch;
    uint32_t *d = dst + blockIdx.x * dst_pitch;
    for (size_t i = threadIdx.x; i < dst_pitch; i += blockDim.x) d[i] = i < width ? s[i] : 0u;
}

// device memory -> page-locked host memory by the shader instead of a copy engine (kp_capi.hip: Fetch)
__global__ __launch_bounds__(256) void kp_read_back_kernel(const uint32_t *__restrict__ src, uint32_t *__restrict__ dst, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) dst[i] = src[i];
}

}  // namespace

void kp_launch_read_back(const void *src, void *dst_pinned, size_t bytes, hipStream_t stream) {
    const size_t n = bytes / 4;  // (callers read whole 32-bit words)
    if (n == 0) return;
    const unsigned blocks = (unsigned)std::min<size_t>((n + 255) / 256, 512);
    hipLaunchKernelGGL(kp_read_back_kernel, dim3(blocks), dim3(256), 0, stream, reinterpret_cast<const uint32_t *>(src),
                       reinterpret_cast<uint32_t *>(dst_pinned), n);
}

void kp_launch_hit_split(const kp_hit *hits, const uint32_t *n_hits, uint32_t hit_cap, int32_t gene_lo, int32_t gene_hi,
                         kp_hit *out, uint32_t *out_n, int32_t n_asm, hipStream_t stream) {
    if (n_asm == 0) return;
    hipLaunchKernelGGL(kp_hit_split_kernel, dim3(n_asm), dim3(256), 0, stream, hits, n_hits, hit_cap, gene_lo, gene_hi, out,
                       out_n);
}

void kp_launch_pack_rows(const uint32_t *src, size_t src_pitch, uint32_t *dst, size_t dst_pitch, size_t width, int rows,
                         hipStream_t stream) {
    if (rows == 0 || dst_pitch == 0) return;
    hipLaunchKernelGGL(kp_pack_rows_kernel, dim3(rows), dim3(256), 0, stream, src, src_pitch, dst, dst_pitch, width);
}

void kp_launch_hit_finalise(const KpBatchView &b, const int32_t *gene_len, const KpTask *tasks, const KpSwResult *results,
                            const uint8_t *task_drop, const uint32_t *task_count, uint32_t task_cap, kp_hit *raw, uint32_t *n_raw, uint32_t hit_cap,
                            uint64_t *keys, kp_hit *hits, uint32_t *n_hits, unsigned long long *cells,
                            const float *ln_half, const float *ln_int, const KpJoin *joins, const uint32_t *join_count, uint32_t join_cap,
                            hipStream_t stream) {
    if (b.n_asm == 0) return;
    hipLaunchKernelGGL(kp_hit_compact_kernel, dim3(512, KP_N_CLASSES), dim3(256), 0, stream, b, gene_len, tasks, results, task_drop, task_count,
                       task_cap, raw, n_raw, hit_cap, cells);
    hipLaunchKernelGGL(kp_join_hits_kernel, dim3(16, KP_N_CLASSES), dim3(64), 0, stream, b, gene_len, joins, join_count, join_cap, raw,
                       n_raw, hit_cap, cells);
    hipLaunchKernelGGL(kp_hit_sort_kernel, dim3(b.n_asm), dim3(SORT_THREADS), 0, stream, raw, n_raw, hit_cap, keys, hits,
                       n_hits, ln_half, ln_int);
}

void kp_launch_score(const KpBatchView &b, const kp_hit *hits, const uint32_t *n_hits, uint32_t hit_cap,
                     const KpTypingDb &db, double min_cov, double *scores, int32_t *counts, hipStream_t stream) {
    if (b.n_asm == 0) return;
    hipLaunchKernelGGL(kp_score_kernel, dim3(b.n_asm), dim3(64), 0, stream, hits, n_hits, hit_cap, db, min_cov, scores, counts);
}

void kp_launch_reduce(const KpBatchView &b, const kp_hit *hits, const uint32_t *n_hits, uint32_t hit_cap,
                      const KpTypingDb &db, const KpTypingParams &prm, const int32_t *best, uint64_t *keys,
                      uint32_t *order, uint8_t *kept_flag, KpKept *kept, int kept_cap, KpPiece *pieces, int piece_cap,
                      KpAsmSummary *summary, uint8_t *prot, int prot_cap, int32_t *pair_q_off, int32_t *pair_q_len,
                      int32_t *pair_t_off, int32_t *pair_t_len, int32_t *n_pairs, int32_t *pair_base,
                      hipStream_t stream) {
    if (b.n_asm == 0) return;
    hipLaunchKernelGGL(kp_reduce_kernel, dim3(b.n_asm), dim3(64), 0, stream, b, hits, n_hits, hit_cap, db, prm, best, keys,
                       order, kept_flag, kept, kept_cap, pieces, piece_cap, summary, prot, prot_cap, pair_q_off, pair_q_len,
                       pair_t_off, pair_t_len, n_pairs, pair_base);
}

void kp_launch_states(const KpBatchView &b, const KpTypingDb &db, const KpTypingParams &prm, KpKept *kept, int kept_cap,
                      KpAsmSummary *summary, const int32_t *dp8, const int32_t *pair_base, hipStream_t stream) {
    if (b.n_asm == 0) return;
    hipLaunchKernelGGL(kp_state_kernel, dim3(b.n_asm), dim3(64), 0, stream, b, db, prm, kept, kept_cap, summary, dp8,
                       pair_base);
}
