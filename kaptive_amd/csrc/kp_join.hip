// kp_join.hip -- kp-align v5: chains of a group's anchors across diagonal jumps of up to KP_JOIN_BW, and their joined alignment.
//
// minimap2 (which the reference's aligner wraps: src/kaptive/serotyping/core.py:147-155, docs/serotyping/method.md:23-28)
// chains anchors whose diagonals differ by up to bw = 500 and aligns through the gap, so a gene with an insertion or
// deletion of 33-500 bases is one hit there.  The band tasks of kp_chain.hip / kp_sw.hip stop at KP_DIAG_GAP = 32 diagonals;
// this file puts the join on top of them (include/kp_spec.h, GROUPS .. CONSUMED PIECES, is the specification;
// oracle/kp_oracle.c chain_group / make_joins / join_run the CPU statement):
//
//   kp_join_chain_kernel   one wave per GROUP of clusters (kp_chain.hip finds them; weak clusters are members): minimap2's
//                          chaining DP over all the group's ANCHORS, the backtracking, every chain cut into pieces where its
//                          diagonal jumps: one KpJoin per chain of two or more pieces
//   kp_join_fill_kernel    P lanes per join (band of 4P diagonals, the mapping of kp_sw.hip's 32-bit kernel): the pieces one
//                          after the other, every one a local alignment over the rows between its neighbours' anchors; in
//                          the junction zones a piece takes the cross gaps the piece before offers (per row or per column:
//                          the best H + e * position for both pieces of the gap cost, collected with 64-bit atomic maxima
//                          in memory); direction bytes go to the trace buffer
//   kp_join_trace_kernel   one lane per join: the pieces in the order of their best cells, the walk back through the cross
//                          gaps with the two-sided drop test, the joined hit; band tasks whose clusters a chain consumes get
//                          their drop flag (task_drop: read by the hit compaction -- the band fill is writing the task
//                          results while this runs)
//
// All of it runs on a stream of its own beside the band tasks' fill and traceback (kp_capi.hip).  Plain 32-bit arithmetic:
// some twenty joins per thousand assemblies on the headline workload, 55 per assembly on `bench.py --mix joins`.
#include "kp_internal.h"
#include <cstdio>
#include <cstdlib>
#include <type_traits>

namespace {

__constant__ uint8_t c_join_pen[KP_CHAIN_PEN_SIZE] = KP_CHAIN_PEN_TABLE;

constexpr int JNEG = KP_NEG_INF;
__device__ __forceinline__ bool dead(int v) { return v < JNEG / 2; }
constexpr int XBIAS = 1 << 23;  // export keys: (value + e * position + XBIAS) in the high word -- values of a continuation can be negative

__device__ __forceinline__ int class_of_width(int w) { return w == 16 ? 0 : (w == 32 ? 1 : (w == 64 ? 2 : 3)); }

// ---- groups -> joins ---------------------------------------------------------------------------------------------------------------
// One wave per group (kp_spec.h, CHAINS OF ANCHORS; oracle/kp_oracle.c chain_group): the group's anchors are gathered into LDS
// as (target, query, cluster) words and sorted by a bitonic network; minimap2's chaining DP runs over them -- anchor by anchor,
// the 64 lanes sharing the earlier anchors of each, the winner by a wave reduction on (score, index) --; lane 0 backtracks
// (mg_chain_backtrack), cuts every chain into pieces and appends a KpJoin per chain of two or more pieces.
constexpr int JA_MAX = KP_JOIN_ANCHOR_MAX;
constexpr int JA_SMALL = 1024;  // nearly every group (a gene of up to ~5 kb): 15 KB of LDS, ten blocks a CU; the rest take the variant without LDS
constexpr size_t JOIN_CHAIN_SCRATCH = 15 * (size_t)JA_MAX;  // bytes of device memory per block of that variant
static_assert((JA_MAX & (JA_MAX - 1)) == 0 && JA_MAX <= 4096 && KP_K * JA_MAX < 65536, "bitonic network; 13-bit indices; 16-bit scores");

__device__ __forceinline__ int ja_t(uint64_t w) { return (int)(w >> 20); }
__device__ __forceinline__ int ja_q(uint64_t w) { return (int)((w >> 4) & 0xFFFFu); }
__device__ __forceinline__ int ja_c(uint64_t w) { return (int)(w & 15u); }
static_assert(KP_JOIN_GROUP_MAX <= 16 && KP_MAX_GENE_LEN <= 0xFFFF, "anchor word layout");

template <int JA, int JA_BELOW>  // groups of JA_BELOW < n <= JA anchors are this instance's
__global__ __launch_bounds__(64) void kp_join_chain_kernel(const uint64_t *__restrict__ keys, uint32_t cap, KpKeyBits kb,
                                                          const KpTask *__restrict__ tasks, uint32_t task_cap,
                                                          const KpGroup *__restrict__ groups, const uint32_t *__restrict__ group_count,
                                                          uint32_t group_cap, KpJoin *__restrict__ joins, uint32_t *__restrict__ join_count,
                                                          uint32_t join_cap, int prio, uint8_t *__restrict__ scratch) {
    // working arrays: in LDS for the groups of up to JA_SMALL anchors; the instance for the rare larger ones keeps them in
    // device memory (a slice of `scratch` per block, L2-resident) -- 60 KB of LDS a block could not be placed on a CU beside the
    // band fill's blocks, and the join kernels behind it then started only when that fill was over (round 6: 2.7 % of the step)
    uint64_t *s_a;
    uint16_t *s_f;   // a chain of n anchors scores at most KP_K * n
    uint16_t *s_pm;  // prefix maximum of f: the scan for predecessors stops where nothing earlier can win
    int16_t *s_p;
    uint8_t *s_used;  // 0 free, 1 member of a chain, 2 free but already tried as a chain's end
    if constexpr (JA_BELOW == 0) {
        __shared__ uint64_t l_a[JA];
        __shared__ uint16_t l_f[JA], l_pm[JA];
        __shared__ int16_t l_p[JA];
        __shared__ uint8_t l_used[JA];
        s_a = l_a; s_f = l_f; s_pm = l_pm; s_p = l_p; s_used = l_used;
    } else {
        uint8_t *base = scratch + (size_t)blockIdx.x * JOIN_CHAIN_SCRATCH;
        s_a = reinterpret_cast<uint64_t *>(base);
        s_f = reinterpret_cast<uint16_t *>(base + 8 * (size_t)JA);
        s_pm = reinterpret_cast<uint16_t *>(base + 10 * (size_t)JA);
        s_p = reinterpret_cast<int16_t *>(base + 12 * (size_t)JA);
        s_used = base + 14 * (size_t)JA;
    }
    int16_t *s_chain = reinterpret_cast<int16_t *>(s_pm);  // (the backtracking no longer needs the prefix maxima)
    uint32_t n_groups = *group_count;
    if (n_groups > group_cap) n_groups = group_cap;
    const int lane = threadIdx.x;
    (void)tasks; (void)task_cap;
    if (prio) __builtin_amdgcn_s_setprio(3);  // one wave per group, a dependent step per anchor: latency, not throughput
    for (uint32_t g = blockIdx.x; g < n_groups; g += gridDim.x) {
        const KpGroup &G = groups[g];
        const int n = (int)G.total;
        if (n > JA || n <= JA_BELOW || n < KP_MIN_ANCHORS) continue;  // (kp_spec.h: a group beyond JA_MAX anchors is not chained)
        const int n_members = G.n;
        __syncthreads();  // the group before is done with the arrays
        int base = 0;
        for (int c = 0; c < n_members; ++c) {
            const uint64_t *k = keys + (size_t)G.asm_id * cap + G.first[c];
            const int cnt = (int)G.cnt[c];
            for (int i = lane; i < cnt; i += 64) {
                const uint64_t key = k[i];
                const uint32_t q = kp_ckey_qpos(key, kb);
                const uint32_t t = kp_ckey_diag(key, kb) - (uint32_t)KP_DIAG_BIAS + q;
                s_a[base + i] = ((uint64_t)t << 20) | ((uint64_t)q << 4) | (uint64_t)c;
            }
            base += cnt;
        }
        int n2 = 64;
        while (n2 < n) n2 <<= 1;
        for (int i = n + lane; i < n2; i += 64) s_a[i] = ~0ull;
        for (int i = lane; i < n; i += 64) s_used[i] = 0;
        __syncthreads();
        for (int kk = 2; kk <= n2; kk <<= 1)  // by (target, query): no two anchors of a gene/strand share both
            for (int j = kk >> 1; j >= 1; j >>= 1) {
                for (int t = lane; t < n2 / 2; t += 64) {
                    const int lo_i = ((t & ~(j - 1)) << 1) | (t & (j - 1)), hi_i = lo_i | j;
                    const bool asc = (lo_i & kk) == 0;
                    const uint64_t a = s_a[lo_i], c = s_a[hi_i];
                    if ((a < c) != asc) { s_a[lo_i] = c; s_a[hi_i] = a; }
                }
                __syncthreads();
            }
        // the chaining DP: f[i] = max(K, max over earlier j of f[j] + sc(i, j)), the first maximum met going backwards
        for (int i = 0; i < n; ++i) {
            const uint64_t wi = s_a[i];
            const int ti = ja_t(wi), qi = ja_q(wi);
            int best = KP_K, bj = 0;  // bj = j + 1, 0 = none
            for (int jb = i - 1; jb >= 0; jb -= 64) {
                if (ti - ja_t(s_a[jb]) > KP_CHAIN_MAX_DIST) break;  // (sorted by target: every earlier anchor is further still)
                // a link adds at most KP_K: once the best f of everything from jb down cannot beat what a lane already holds,
                // nothing earlier can (exact pruning: on a co-linear chain the scan ends after its first round)
                int seen = best;
#pragma unroll
                for (int o = 32; o >= 1; o >>= 1) seen = max(seen, __shfl_xor(seen, o));
                if ((int)s_pm[jb] + KP_K <= seen) break;
                const int j = jb - lane;
                if (j >= 0) {
                    const uint64_t wj = s_a[j];
                    const int dr = ti - ja_t(wj), dq = qi - ja_q(wj);
                    if (dr <= KP_CHAIN_MAX_DIST && dq > 0 && dq <= KP_CHAIN_MAX_DIST && dr != 0) {
                        const int dd = dr > dq ? dr - dq : dq - dr, dg = dr < dq ? dr : dq;
                        if (dd <= KP_JOIN_BW) {
                            int sc = dg < KP_K ? dg : KP_K;
                            if (dd || dg > KP_K) sc -= (int)c_join_pen[dd];
                            sc += (int)s_f[j];
                            if (sc > best) { best = sc; bj = j + 1; }
                        }
                    }
                }
            }
            uint32_t key = ((uint32_t)best << 13) | (uint32_t)bj;  // the largest score, the largest j among equals
#pragma unroll
            for (int o = 32; o >= 1; o >>= 1) key = max(key, (uint32_t)__shfl_xor((int)key, o));
            if (lane == 0) {
                const uint16_t fi = (uint16_t)(key >> 13);
                s_f[i] = fi; s_p[i] = (int16_t)((int)(key & 0x1FFFu) - 1);
                s_pm[i] = i > 0 && s_pm[i - 1] > fi ? s_pm[i - 1] : fi;
            }
            __syncthreads();
        }
        // mg_chain_backtrack: ends by (f, index) descending
        for (;;) {
            uint32_t key = 0;
            for (int i = lane; i < n; i += 64)
                if (s_used[i] == 0 && s_f[i] >= KP_MIN_CHAIN_SCORE) key = max(key, ((uint32_t)s_f[i] << 13) | (uint32_t)(i + 1));
#pragma unroll
            for (int o = 32; o >= 1; o >>= 1) key = max(key, (uint32_t)__shfl_xor((int)key, o));
            if (key == 0) break;
            const int end = (int)(key & 0x1FFFu) - 1;
            if (lane == 0) {
                const int f_end = (int)s_f[end];
                int i = end, max_s = 0, steps = 0, cut = 0;
                do {  // back until a used anchor or the start; the chain is cut where the score counted from its end peaks
                    i = s_p[i]; ++steps;
                    const int sc = i < 0 ? f_end : f_end - (int)s_f[i];
                    if (sc > max_s) { max_s = sc; cut = steps; }
                    else if (max_s - sc > KP_JOIN_BW) break;
                } while (i >= 0 && s_used[i] != 1);
                int len = 0;
                for (i = end; len < cut; i = s_p[i]) { s_chain[len++] = (int16_t)i; s_used[i] = 1; }
                // (the walk ran into a used anchor below its own level: no chain ends here.  The anchor is not tried as an end
                // again -- the ends are visited once, in order -- but stays free for the walks of later ends: 2)
                if (cut == 0) s_used[end] = 2;
                if (max_s >= KP_MIN_CHAIN_SCORE && len >= KP_MIN_ANCHORS) {
                    // pieces, in query order (the chain was walked backwards)
                    int np = 0, dmin[KP_JOIN_MAX_PIECES], dmax[KP_JOIN_MAX_PIECES], cm[KP_JOIN_MAX_PIECES];
                    int qlo[KP_JOIN_MAX_PIECES], qhi[KP_JOIN_MAX_PIECES], jump_before[KP_JOIN_MAX_PIECES];
                    int prev_d = 0;
                    bool over = false;
                    for (int z = len - 1; z >= 0; --z) {
                        const uint64_t w = s_a[s_chain[z]];
                        const int d = ja_t(w) - ja_q(w);
                        bool fresh = np == 0;
                        if (!fresh) {
                            const int jump = d > prev_d ? d - prev_d : prev_d - d;
                            const int lo2 = min(d, dmin[np - 1]), hi2 = max(d, dmax[np - 1]);
                            fresh = jump > KP_DIAG_GAP || hi2 - lo2 > KP_MAX_SPREAD;
                        }
                        if (fresh) {
                            if (np == KP_JOIN_MAX_PIECES) { over = true; break; }
                            dmin[np] = dmax[np] = d; cm[np] = 0; qlo[np] = ja_q(w);
                            jump_before[np] = np ? (d > prev_d ? d - prev_d : prev_d - d) : 0;
                            ++np;
                        }
                        dmin[np - 1] = min(dmin[np - 1], d); dmax[np - 1] = max(dmax[np - 1], d);
                        cm[np - 1] |= 1 << ja_c(w);
                        qhi[np - 1] = ja_q(w);
                        prev_d = d;
                    }
                    if (!over && np >= 2) {
                        int spread = 0;  // the widest piece's diagonal range sets the band of all (kp_spec.h: kp_piece_margin / kp_piece_width)
                        for (int k = 0; k < np; ++k) spread = max(spread, dmax[k] - dmin[k]);
                        const int margin = kp_piece_margin(spread), width = kp_piece_width(spread);
                        const int cls = class_of_width(width);
                        const uint32_t slot = atomicAdd(&join_count[cls], 1u);
                        if (slot < join_cap) {  // beyond cap: counted, not stored (host retries)
                            KpJoin *J = joins + (size_t)cls * join_cap + slot;
                            J->asm_id = G.asm_id; J->gs = G.gs; J->contig = G.contig; J->n_pieces = np; J->n_anchors = len;
                            J->chain_score = max_s; J->width = width; J->n_members = n_members; J->drop_mask = 0;
                            J->weak_mask = kp_weak_ends(np, qlo, qhi, jump_before);
                            for (int k = 0; k < KP_JOIN_MAX_PIECES; ++k) {
                                J->r0[k] = k > 0 && k < np ? qhi[k - 1] & ~7 : 0;
                                J->r1[k] = k + 1 < np ? qlo[k + 1] + KP_K : KP_MAX_GENE_LEN + 1;
                            }
                            for (int c = 0; c < KP_JOIN_GROUP_MAX; ++c) J->member_task[c] = c < n_members ? G.task[c] : KP_REF_NONE;
                            for (int k = 0; k < KP_JOIN_MAX_PIECES; ++k) {
                                if (k < np) {
                                    const int need = dmax[k] - dmin[k] + 1 + 2 * margin;
                                    J->lo[k] = dmin[k] - margin - (width - need) / 2;
                                    J->cmask[k] = cm[k];
                                } else { J->lo[k] = 0; J->cmask[k] = 0; }
                                J->trace_off[k] = J->export_off[k] = 0xFFFFFFFFu;
                                J->end_s[k] = 0; J->end_r[k] = J->end_b[k] = -1;
                                J->state[k] = 0; J->visited[k] = 0;
                                for (int z = 0; z < 9; ++z) J->res[k][z] = 0;
                            }
                        }
                    }
                }
            }
            __syncthreads();
        }
    }
}

// ---- joined fill ------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ unsigned nib4(unsigned word, int i) { return (word >> (4 * i)) & 15u; }

// The pieces of a join are filled one after the other by the SAME lanes of one wave: what a piece exports is read by its own
// wave only, so the exports' atomics, the loads that import them and the fences between pieces need the scope of a work group
// (one wave here), not of the device.  A device-scope fence is an L2 write-back and invalidate on gfx950 (`buffer_wbl2 sc1`,
// `buffer_inv sc1`: the L2s of the eight XCDs are not coherent with each other) -- two per piece, in the middle of a band fill
// that streams 10 GB of direction bits through those L2s: the 20 joins of a headline batch cost the step 2.2 % that way
// (tools/experiments/join_cost_bits.sh).  The walk-back is a later kernel on the same stream.
__device__ __forceinline__ void join_fence() { __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup"); }
__device__ __forceinline__ void join_export(unsigned long long *p, unsigned long long v) {
    (void)__hip_atomic_fetch_max(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

// Direction byte of a cell: bits 0-2 the source of H (XT_*), bit 3 "E was extended", bit 4 "F was extended".
enum { XT_DIAG = 0, XT_E = 1, XT_F = 2, XT_RESTART = 3, XT_X1 = 4, XT_X2 = 5 };

template <int P>
__device__ __forceinline__ void join_fill_class(const KpBatchView &b, const KpGenes &genes, KpJoin *__restrict__ joins, uint32_t n_joins,
                                                uint4 *__restrict__ trace, unsigned long long *__restrict__ trace_top, uint64_t trace_cap,
                                                uint32_t block, uint32_t n_blocks) {
    constexpr int G = 64 / P, W = 4 * P;
    const int lane = threadIdx.x, g = lane / P, l = lane % P;
    for (uint32_t quad = block; (uint64_t)quad * G < n_joins; quad += n_blocks) {
        const uint32_t slot = quad * G + g;
        const bool have = slot < n_joins;
        KpJoin *J = joins + (have ? slot : 0u);
        const int n_pieces = have ? J->n_pieces : 0;
        const int gs = have ? J->gs : 0, gene = gs >> 1, asm_id = have ? J->asm_id : 0;
        const int qlen = have ? genes.len[gene] : 0;
        const uint32_t *qnib = genes.nib + genes.word_off[(gs & 1) ? genes.n_genes + gene : gene];
        const uint32_t *asm_words = b.words + b.asm_word_off[asm_id];
        const int asm_n_words = (int)(b.asm_word_off[asm_id + 1] - b.asm_word_off[asm_id]);
        const int c_abs = b.asm_first_ctg[asm_id] + (have ? J->contig : 0);
        const int cstart = b.ctg_start[c_abs], cend = cstart + b.ctg_len[c_abs];
        const int r0n = b.asm_first_nrun[asm_id], n_runs = b.asm_first_nrun[asm_id + 1] - r0n;
        const int32_t *runs = b.n_runs + 2 * (size_t)r0n;
        int max_pieces = n_pieces;
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) max_pieces = max(max_pieces, __shfl_xor(max_pieces, o));
        // what the piece before left for this one
        unsigned long long *imp = nullptr;  // its export array: two keys per index
        bool imp_horizontal = false;
        int imp_len = 0, imp_lo = 0;
        bool ok = have;  // false once the trace buffer has run out
        for (int k = 0; k < max_pieces; ++k) {
            const bool act = ok && k < n_pieces;
            const int lo = act ? J->lo[k] : 0;
            const bool cont = k > 0;  // (pieces after the first take the cross gaps of the piece before; every piece is local: H >= 0, restarts)
            const int none = 0;
            int q0 = 0, r_hi = 0;
            if (act) kp_piece_rows(lo, W, cstart, cend, qlen, J->r0[k], J->r1[k], &q0, &r_hi);
            const int steps = act ? (r_hi - q0) + P - 1 : 0;
            const int steps8 = (steps + 7) & ~7;
            const bool exports = act && k + 1 < n_pieces;
            const bool exp_horizontal = exports && J->lo[k + 1] > lo;
            const int exp_len = exports ? (exp_horizontal ? qlen : qlen + W) : 0;
            const int lo_next = exports ? J->lo[k + 1] : 0;
            // room: direction bytes (a word per lane and step) and the export array (two 64-bit keys per index)
            // (both multiples of eight 16-byte units: the band fill's task blocks -- same buffer, same bump counter -- count on
            // starting on 128-byte lines, kp_sw.hip; an export array of an odd length shifted every block allocated after it)
            const unsigned long long t_units = (unsigned long long)steps8 * P / 4, x_units = ((unsigned long long)exp_len + 7ull) & ~7ull;
            unsigned long long toff = 0;
            if (act && l == 0) toff = atomicAdd(trace_top, t_units + x_units);
            toff = ((unsigned long long)__shfl((unsigned)(toff >> 32), g * P) << 32) | __shfl((unsigned)toff, g * P);
            const bool fits = act && toff + t_units + x_units <= trace_cap && toff + t_units + x_units < 0xFFFFFFFFull;
            uint32_t *tr = reinterpret_cast<uint32_t *>(trace + toff);
            unsigned long long *exp = reinterpret_cast<unsigned long long *>(trace + toff + t_units);
            if (fits && exports) {
                for (int i = l; i < 2 * exp_len; i += P) exp[i] = 0ull;
                join_fence();
            }
            int max_steps = fits ? steps8 : 0;
#pragma unroll
            for (int o = 32; o >= 1; o >>= 1) max_steps = max(max_steps, __shfl_xor(max_steps, o));
            int H[4], E[4], F[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) { H[c] = none; E[c] = F[c] = JNEG; }
            int best = 0, best_r = -1, best_b = 4 * l;
            // Eight steps at a time: what they read -- the query rows, the target bases and the keys of the cross gaps -- does not
            // depend on the cells before them, so it is all requested up front and the eight dependent steps then run from
            // registers (a step that fetched its own operands waited for memory three times: 3.3 ms for a 1300-row piece).
            // The junction zones (kp_spec.h): cross gaps are taken in rows below r1 of the piece before, and offered from row r0
            // of the piece after -- the few dozen rows between the neighbours' anchors.  Everywhere else a chunk is a plain banded
            // fill: no key loads, no candidates, no atomics (the kernel's 800 instructions a step were mostly those).
            const int imp_r1 = act && cont ? min(J->r1[k - 1], qlen) : 0;       // imports in rows < imp_r1
            const int exp_r0 = exports ? J->r0[k + 1] : 0x7FFFFFFF;           // exports from rows >= exp_r0
            // (round 6) The windows of a chunk are requested while the chunk before it computes: the headline workload has some
            // twenty joins a batch, a few waves whose chunks each waited out a memory round trip on a device full of other
            // passes' kernels (3.5 ms per launch, 2 % of the step).  Columns inside N runs come as an eleven-bit mask per
            // chunk, not as a binary search of the assembly's run list per cell.
            auto load_windows = [&](const int m0, uint64_t &qwin, uint64_t &twin) {
                const int r0 = q0 + m0 - l;  // this lane's row at the chunk's first step
                const int tb0 = lo + r0 + 4 * l;  // ... and the column of its cell 0 there: step s, cell c sits on tb0 + s + c
                const int w0 = r0 >> 3, nq = (qlen + 7) >> 3;  // (arithmetic shifts: rows before the gene read as N)
                const uint32_t lo_w = (fits && w0 >= 0 && w0 < nq) ? qnib[w0] : 0x44444444u;
                const uint32_t hi_w = (fits && w0 + 1 >= 0 && w0 + 1 < nq) ? qnib[w0 + 1] : 0x44444444u;
                qwin = (((uint64_t)hi_w << 32) | lo_w) >> (4 * (r0 & 7));
                const int v0 = tb0 >> 4;
                const uint32_t lo_t = (fits && v0 >= 0 && v0 < asm_n_words) ? asm_words[v0] : 0u;
                const uint32_t hi_t = (fits && v0 + 1 >= 0 && v0 + 1 < asm_n_words) ? asm_words[v0 + 1] : 0u;
                twin = (((uint64_t)hi_t << 32) | lo_t) >> (2 * (tb0 & 15));  // eleven bases: 22 of the 34 bits that are left
            };
            auto chunk = [&](const int m0, const uint64_t qwin, const uint64_t twin, auto io_tag, auto nr_tag) {
                constexpr bool IO = decltype(io_tag)::value, NR = decltype(nr_tag)::value;
                const int r0 = q0 + m0 - l;
                const int tb0 = lo + r0 + 4 * l;
                uint32_t nmask = 0;  // bit j: column tb0 + j lies in an N run
                if constexpr (NR) {
                    if (fits && n_runs > 0) {  // (rare: assemblies with scaffold gaps)
                        int a2 = 0, z2 = n_runs;
                        while (a2 < z2) {  // the first run that ends beyond tb0
                            const int mid = (a2 + z2) >> 1;
                            if (runs[2 * mid + 1] <= tb0) a2 = mid + 1; else z2 = mid;
                        }
                        for (; a2 < n_runs && runs[2 * a2] < tb0 + 11; ++a2) {
                            const int s0 = max(runs[2 * a2] - tb0, 0), e0 = min(runs[2 * a2 + 1] - tb0, 11);
                            if (e0 > s0) nmask |= ((1u << e0) - 1u) & ~((1u << s0) - 1u);
                        }
                    }
                }
                unsigned long long kx1[IO ? 11 : 1], kx2[IO ? 11 : 1];  // cross-gap keys: of rows r0 + s (horizontal) or of columns tb0 + j (vertical)
                if constexpr (IO) {
#pragma unroll
                    for (int j = 0; j < 11; ++j) {
                        kx1[j] = kx2[j] = 0ull;
                        if (fits && cont && imp) {
                            const int xi = imp_horizontal ? r0 + j : tb0 + j - imp_lo;
                            if ((!imp_horizontal || j < 8) && xi >= 0 && xi < imp_len) {
                                kx1[j] = __hip_atomic_load(&imp[2 * xi], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                                kx2[j] = __hip_atomic_load(&imp[2 * xi + 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                            }
                        }
                    }
                }
#pragma unroll
                for (int sidx = 0; sidx < 8; ++sidx) {
                    const int m = m0 + sidx, r = r0 + sidx;
                    const bool row_ok = fits && m < steps8 && r >= q0 && r < r_hi;
                    const int qc = (int)((qwin >> (4 * sidx)) & 15u);
                    // left neighbour of cell 0: lane l - 1's cell 3 as the previous step left it; upper neighbour of cell 3: lane
                    // l + 1's cell 0 of this step (DPP moves, one vector instruction each)
                    int hl = __builtin_amdgcn_update_dpp(0, H[3], 0x138, 0xf, 0xf, false), el = __builtin_amdgcn_update_dpp(0, E[3], 0x138, 0xf, 0xf, false);  // wave_shr:1
                    if (l == 0) { hl = none; el = JNEG; }
                    const int oldH[4] = {H[0], H[1], H[2], H[3]}, oldF[4] = {F[0], F[1], F[2], F[3]};
                    uint32_t word = 0;
                    int hu_d = none, fu_d = JNEG;
                    unsigned long long ex1 = 0ull, ex2 = 0ull;  // a row's offers to the next piece from this lane's four cells (insertion)
                    const bool imp_row = IO && cont && imp && r < imp_r1, exp_row = IO && exports && r >= exp_r0;
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        if (c == 3) {  // (every lane has computed its cell 0 by now)
                            hu_d = __builtin_amdgcn_update_dpp(0, H[0], 0x130, 0xf, 0xf, false); fu_d = __builtin_amdgcn_update_dpp(0, F[0], 0x130, 0xf, 0xf, false);  // wave_shl:1
                            if (l == P - 1) { hu_d = none; fu_d = JNEG; }
                        }
                        const int t = tb0 + sidx + c;
                        int code = 5;  // 0..3, 4 = N, 5 = outside the contig
                        if (row_ok && t >= cstart && t < cend) {
                            code = (int)((twin >> (2 * (sidx + c))) & 3u);
                            if constexpr (NR) {
                                if ((nmask >> (sidx + c)) & 1u) code = 4;
                            }
                        }
                        const bool inside = code < 5;
                        const int hleft = c == 0 ? hl : H[c - 1], eleft = c == 0 ? el : E[c - 1];
                        const int hup = c == 3 ? hu_d : oldH[c + 1], fup = c == 3 ? fu_d : oldF[c + 1];
                        const int hd = oldH[c];
                        const int e_open = hleft - (KP_GAP_OPEN + KP_GAP_EXT), e_ext = eleft - KP_GAP_EXT;
                        const int f_open = hup - (KP_GAP_OPEN + KP_GAP_EXT), f_ext = fup - KP_GAP_EXT;
                        int e = e_open >= e_ext ? e_open : e_ext, f = f_open >= f_ext ? f_open : f_ext;
                        const uint32_t e_extd = e_open >= e_ext ? 0u : 1u, f_extd = f_open >= f_ext ? 0u : 1u;
                        if (dead(e)) e = JNEG;
                        if (dead(f)) f = JNEG;
                        const int sc = (qc > 3 || code > 3) ? KP_SC_N : (qc == code ? KP_SC_MATCH : KP_SC_MISMATCH);
                        int bv = hd + sc;
                        uint32_t tb = XT_DIAG;
                        if (e > bv) { bv = e; tb = XT_E; }
                        if (f > bv) { bv = f; tb = XT_F; }
                        if constexpr (IO) {
                            if (inside && imp_row) {
                                const unsigned long long k1 = imp_horizontal ? kx1[sidx] : kx1[sidx + c], k2 = imp_horizontal ? kx2[sidx] : kx2[sidx + c];
                                const int pos = imp_horizontal ? t - imp_lo : r;  // (the exporter's frame: columns count from its band's origin)
                                if (k1) {  // (a row / column with one key has both)
                                    const int c1 = (int)(k1 >> 32) - XBIAS - KP_GAP_OPEN - KP_GAP_EXT * pos;
                                    const int c2 = (int)(k2 >> 32) - XBIAS - KP_GAP_OPEN2 - KP_GAP_EXT2 * pos;
                                    if (c1 > bv) { bv = c1; tb = XT_X1; }
                                    if (c2 > bv) { bv = c2; tb = XT_X2; }
                                }
                            }
                        }
                        const bool live = inside && bv > 0;
                        E[c] = inside ? e : JNEG; F[c] = inside ? f : JNEG;
                        H[c] = live ? bv : none;
                        word |= ((live ? tb : (uint32_t)XT_RESTART) | (e_extd << 3) | (f_extd << 4)) << (8 * c);
                        if (live) {
                            if (bv > best) { best = bv; best_r = r; best_b = 4 * l + c; }
                            if constexpr (IO) {
                                if (exp_row && (exp_horizontal ? (4 * l + c < lo_next - lo) : (lo + 4 * l + c > lo_next + W - 1))) {
                                    const int xi = exp_horizontal ? r : t - lo, pos = exp_horizontal ? t - lo : r;
                                    const unsigned long long low = 0xFFFFFFFFull - (unsigned)pos;
                                    const unsigned long long k1 = ((unsigned long long)(unsigned)(bv + KP_GAP_EXT * pos + XBIAS) << 32) | low;
                                    const unsigned long long k2 = ((unsigned long long)(unsigned)(bv + KP_GAP_EXT2 * pos + XBIAS) << 32) | low;
                                    if (exp_horizontal) {  // the lane's four cells lie in one row: one pair of atomics per lane and step, not
                                        ex1 = k1 > ex1 ? k1 : ex1; ex2 = k2 > ex2 ? k2 : ex2;  // four on the same two words (what the L2 serialises)
                                    } else {
                                        join_export(&exp[2 * xi], k1);
                                        join_export(&exp[2 * xi + 1], k2);
                                    }
                                }
                            }
                        }
                    }
                    if constexpr (IO) {
                        if (ex1) { join_export(&exp[2 * r], ex1); join_export(&exp[2 * r + 1], ex2); }
                    }
                    if (fits && m < steps8) tr[(size_t)m * P + l] = word;
                }
            };
            const bool nr_any = __any(n_runs > 0);
            uint64_t q_next = 0, t_next = 0;
            if (max_steps > 0) load_windows(0, q_next, t_next);
            for (int m0 = 0; m0 < max_steps; m0 += 8) {
                const uint64_t qwin = q_next, twin = t_next;
                if (m0 + 8 < max_steps) load_windows(m0 + 8, q_next, t_next);
                // does any lane's chunk touch a junction zone?  (its rows: q0 + m0 - l .. + 7)
                const int ra = q0 + m0 - l, rb = ra + 7;
                const bool io = fits && m0 < steps8 && ((cont && imp && ra < imp_r1) || (exports && rb >= exp_r0));
                if (__any(io)) {
                    if (nr_any) chunk(m0, qwin, twin, std::true_type{}, std::true_type{}); else chunk(m0, qwin, twin, std::true_type{}, std::false_type{});
                } else {
                    if (nr_any) chunk(m0, qwin, twin, std::false_type{}, std::true_type{}); else chunk(m0, qwin, twin, std::false_type{}, std::false_type{});
                }
            }
            // END of the piece: the largest score, then the first row, then the first column
#pragma unroll
            for (int o = 1; o < P; o <<= 1) {
                const int s2 = __shfl_xor(best, o), r2 = __shfl_xor(best_r, o), b2 = __shfl_xor(best_b, o);
                if (s2 > best || (s2 == best && (r2 < best_r || (r2 == best_r && b2 < best_b)))) { best = s2; best_r = r2; best_b = b2; }
            }
            if (act && !fits) ok = false;
            if (act && l == 0) {
                J->trace_off[k] = fits ? (uint32_t)toff : 0xFFFFFFFFu;
                J->export_off[k] = fits && exports ? (uint32_t)(toff + t_units) : 0xFFFFFFFFu;
                J->end_s[k] = fits && best_r >= 0 ? best : 0; J->end_r[k] = fits ? best_r : -1; J->end_b[k] = best_b;
            }
            join_fence();  // this piece's exports are in memory before the next piece reads them
            imp = fits && exports ? exp : nullptr;
            imp_horizontal = exp_horizontal; imp_len = exp_len; imp_lo = lo;
        }
    }
}

__global__ __launch_bounds__(64) void kp_join_fill_kernel(KpBatchView b, KpGenes genes, KpJoin *__restrict__ joins,
                                                          const uint32_t *__restrict__ join_count, uint32_t join_cap,
                                                          uint4 *__restrict__ trace, unsigned long long *__restrict__ trace_top,
                                                          uint64_t trace_cap, int prio) {
    const int c = blockIdx.y;
    uint32_t n = join_count[c];
    if (n > join_cap) n = join_cap;
    if (n == 0) return;
    // A handful of waves, each a chain of a few thousand dependent steps, on a device that other passes keep full: at the
    // default priority a step waits its turn behind eight other waves of its SIMD (8 ms per pass for 19 joins).  They go first.
    if (prio) __builtin_amdgcn_s_setprio(3);
    KpJoin *list = joins + (size_t)c * join_cap;
    if (c == 3) join_fill_class<32>(b, genes, list, n, trace, trace_top, trace_cap, blockIdx.x, gridDim.x);
    else if (c == 2) join_fill_class<16>(b, genes, list, n, trace, trace_top, trace_cap, blockIdx.x, gridDim.x);
    else if (c == 1) join_fill_class<8>(b, genes, list, n, trace, trace_top, trace_cap, blockIdx.x, gridDim.x);
    else join_fill_class<4>(b, genes, list, n, trace, trace_top, trace_cap, blockIdx.x, gridDim.x);
}

// ---- walk-back: one lane per join -------------------------------------------------------------------------------------------------
// THE JOINED PATH and the CONSUMED PIECES of kp_spec.h (oracle: join_run): the pieces are tried in the order of their best cells'
// scores; the first path the drop test does not reject settles the chain -- a hit if it crosses a gap --, and the band tasks of
// the group's clusters that the chain's pieces hold anchors of lose their own hits (task_drop).
__global__ __launch_bounds__(64) void kp_join_trace_kernel(KpBatchView b, KpGenes genes, KpJoin *__restrict__ joins,
                                                           const uint32_t *__restrict__ join_count, uint32_t join_cap,
                                                           uint32_t task_cap, const uint4 *__restrict__ trace,
                                                           uint8_t *__restrict__ task_drop, int prio) {
    const int cls = blockIdx.y;
    uint32_t n = join_count[cls];
    if (n > join_cap) n = join_cap;
    const int P = 4 << cls, W = 4 * P;
    if (prio) __builtin_amdgcn_s_setprio(3);  // (as the joined fill: few lanes, long dependent walks)
    for (uint32_t ji = blockIdx.x * blockDim.x + threadIdx.x; ji < n; ji += gridDim.x * blockDim.x) {
        KpJoin *J = joins + (size_t)cls * join_cap + ji;
        const int m = J->n_pieces, gs = J->gs, gene = gs >> 1, asm_id = J->asm_id;
        const int qlen = genes.len[gene];
        const uint32_t *qnib = genes.nib + genes.word_off[(gs & 1) ? genes.n_genes + gene : gene];
        const uint32_t *asm_words = b.words + b.asm_word_off[asm_id];
        const int c_abs = b.asm_first_ctg[asm_id] + J->contig;
        const int cstart = b.ctg_start[c_abs], cend = cstart + b.ctg_len[c_abs];
        const int r0n = b.asm_first_nrun[asm_id], n_runs = b.asm_first_nrun[asm_id + 1] - r0n;
        const int32_t *runs = b.n_runs + 2 * (size_t)r0n;
        bool complete = true;
        for (int k = 0; k < m; ++k) complete = complete && J->trace_off[k] != 0xFFFFFFFFu;
        for (int k = 0; k < m; ++k) { J->state[k] = 0; J->visited[k] = 0; }
        J->drop_mask = 0;
        if (!complete) continue;  // (the trace buffer ran out: the host grows it and reruns the pass)
        int settled = 0, hit_k = -1, alone_k = -1;
        bool any_rejected = false;
        for (;;) {
            int k = -1;
            for (int z = 0; z < m; ++z)
                if (!((settled >> z) & 1) && (k < 0 || J->end_s[z] > J->end_s[k])) k = z;
            if (k < 0) break;
            if (J->end_r[k] < 0 || J->end_s[k] < KP_MIN_DP_SCORE) { alone_k = k; break; }
            if (k == 0) { J->visited[0] = 1; alone_k = 0; break; }  // (the first piece has no gap to cross: its path need not be walked to know that)
            int pk = k, r = J->end_r[k], bi = J->end_b[k], state = 0, matches = 0, cols = 0, gap = 0, credit = 0;
            int sr = r, sb = bi, spk = k, suf = 0, sufmax = 0, gsum = 0, visited = 1 << k, bonus = 0;
            bool rejected = false;
            int lo = J->lo[pk], q0 = 0, r_hi = 0;
            kp_piece_rows(lo, W, cstart, cend, qlen, J->r0[pk], J->r1[pk], &q0, &r_hi);
            const uint32_t *tr = reinterpret_cast<const uint32_t *>(trace + J->trace_off[pk]);
            for (;;) {
                const int t = r + lo + bi;
                if (state == 0 && (r < q0 || r >= r_hi || bi < 0 || bi >= W || t < cstart || t >= cend)) break;
                const uint32_t byte = (tr[(size_t)(r - q0 + (bi >> 2)) * P + (bi >> 2)] >> (8 * (bi & 3))) & 255u;
                if (state == 0) {
                    const uint32_t tb = byte & 7u;
                    if (tb == XT_RESTART) break;
                    // the drop test: at every cross gap, and at every cell once a gap has been crossed (cross-gap costs left out)
                    if (suf + gsum > sufmax) sufmax = suf + gsum;
                    else if ((tb >= XT_X1 || visited != (1 << k)) && sufmax - (suf + gsum) > KP_JOIN_DROP) { rejected = true; break; }
                    if (tb == XT_DIAG) {
                        sr = r; sb = bi; spk = pk; ++cols;
                        const uint32_t qc = nib4(qnib[r >> 3], r & 7);
                        uint32_t tc = (asm_words[t >> 4] >> (2 * (t & 15))) & 3u;
                        if (n_runs > 0) {
                            int a = 0, z = n_runs;
                            while (a < z) {
                                const int mid = (a + z) >> 1;
                                if (runs[2 * mid + 1] <= t) a = mid + 1; else z = mid;
                            }
                            if (a < n_runs && runs[2 * a] <= t) tc = 4u;
                        }
                        if (qc < 4u && qc == tc) ++matches;
                        suf += (qc > 3u || tc > 3u) ? KP_SC_N : (qc == tc ? KP_SC_MATCH : KP_SC_MISMATCH);
                        --r;
                    } else if (tb == XT_E || tb == XT_F) {
                        state = (int)tb;
                    } else {  // a cross gap: on to the cell of piece pk - 1 it came from
                        const int lo_prev = J->lo[pk - 1];
                        const bool horizontal = lo > lo_prev;
                        const unsigned long long *exp = reinterpret_cast<const unsigned long long *>(trace + J->export_off[pk - 1]);
                        const int xi = horizontal ? r : t - lo_prev;
                        const unsigned long long key = exp[2 * xi + (tb == XT_X1 ? 0 : 1)];
                        const int pos = (int)(0xFFFFFFFFu - (uint32_t)key);  // t' - lo_prev (horizontal) or r'
                        const int ngap = horizontal ? (t - lo_prev) - pos : r - pos;
                        cols += ngap;
                        const int cost = tb == XT_X1 ? KP_GAP_OPEN + KP_GAP_EXT * ngap : KP_GAP_OPEN2 + KP_GAP_EXT2 * ngap;
                        suf -= cost; gsum += cost;
                        const int lg = KP_GAP_OPEN + kp_log2x2((uint32_t)ngap);
                        if (cost > lg) bonus += cost - lg;
                        if (horizontal) bi = pos - r;               // same row, column lo_prev + pos
                        else { bi = t - pos - lo_prev; r = pos; }   // same column, row pos
                        --pk; visited |= 1 << pk;
                        lo = lo_prev;
                        kp_piece_rows(lo, W, cstart, cend, qlen, J->r0[pk], J->r1[pk], &q0, &r_hi);
                        tr = reinterpret_cast<const uint32_t *>(trace + J->trace_off[pk]);
                    }
                } else if (state == XT_E) {
                    ++cols; ++gap; --bi; suf -= KP_GAP_EXT;
                    if (!(byte & 8u)) { state = 0; suf -= KP_GAP_OPEN; credit += max(gap - KP_GAP_LONG, 0); gap = 0; }
                } else {
                    ++cols; ++gap; --r; ++bi; suf -= KP_GAP_EXT;
                    if (!(byte & 16u)) { state = 0; suf -= KP_GAP_OPEN; credit += max(gap - KP_GAP_LONG, 0); gap = 0; }
                }
            }
            J->visited[k] = visited;
            if (rejected) { J->state[k] = 2; any_rejected = true; settled |= 1 << k; continue; }
            if (visited == (1 << k)) { alone_k = k; break; }  // crosses no gap: the band task of the piece's cluster covers it
            J->state[k] = 1; hit_k = k;
            J->res[k][0] = J->end_s[k]; J->res[k][1] = sr; J->res[k][2] = J->end_r[k] + 1;
            J->res[k][3] = sr + J->lo[spk] + sb; J->res[k][4] = J->end_r[k] + J->lo[k] + J->end_b[k] + 1;
            J->res[k][5] = matches; J->res[k][6] = cols; J->res[k][7] = J->end_s[k] + credit;
            J->res[k][8] = bonus < KP_HIT_BONUS_MAX ? bonus : KP_HIT_BONUS_MAX;
            break;
        }
        // CONSUMED PIECES: the clusters of the pieces the joined hit runs through, and those of the chain's weak ends that no
        // reported path reaches
        const int on = hit_k >= 0 ? J->visited[hit_k] : 0;
        int drop = 0;
        for (int k = 0; k < m; ++k) {
            if ((on >> k) & 1) drop |= J->cmask[k];
            else if (!any_rejected && ((J->weak_mask >> k) & 1) && k != alone_k) drop |= J->cmask[k];
        }
        if (alone_k >= 0) drop &= ~J->cmask[alone_k];
        J->drop_mask = drop;
        for (int c = 0; c < J->n_members; ++c) {
            const uint32_t ref = J->member_task[c];
            if (!((drop >> c) & 1) || ref == KP_REF_NONE || KP_REF_SLOT(ref) >= task_cap) continue;
            task_drop[(size_t)KP_REF_CLS(ref) * task_cap + KP_REF_SLOT(ref)] = 1;  // (read by the hit compaction: kp_reduce.hip)
        }
    }
}

}  // namespace

// grids and wave priority of the join kernels (KAPTIVE_AMD_JOIN_GRID = "fill,walk,chain,chain_large" blocks (per band class for the
// first two), KAPTIVE_AMD_JOIN_PRIO = 0 | 1: experiments; the defaults are what tools/experiments/join_cost_ab.sh measured)
struct JoinLaunch { int fill = 2048, walk = 512, chain = 2560, chain_large = 512, prio = 1; };  // (blocks without work leave at once: a large grid costs nothing measurable)
static const JoinLaunch &join_launch() {
    static const JoinLaunch cfg = [] {
        JoinLaunch c;
        if (const char *e = std::getenv("KAPTIVE_AMD_JOIN_GRID")) std::sscanf(e, "%d,%d,%d,%d", &c.fill, &c.walk, &c.chain, &c.chain_large);
        if (const char *e = std::getenv("KAPTIVE_AMD_JOIN_PRIO")) c.prio = std::atoi(e);
        return c;
    }();
    return cfg;
}

void kp_launch_join_chain(const KpBatchView &b, const KpGenes &genes, const uint64_t *sorted_anchors, uint32_t anchor_cap, KpKeyBits kb,
                          const KpTask *tasks, uint32_t task_cap, const KpGroup *groups, const uint32_t *group_count, uint32_t group_cap,
                          KpJoin *joins, uint32_t *join_count, uint32_t join_cap, uint8_t *scratch, hipStream_t stream) {
    (void)b; (void)genes;
    if (const char *e = std::getenv("KAPTIVE_AMD_SKIP_JOINS")) if (std::atoi(e) & 1) return;  // (debugging aid: bit 0 chaining, 1 fill, 2 walk-back)
    const JoinLaunch &L = join_launch();
    hipLaunchKernelGGL((kp_join_chain_kernel<JA_SMALL, 0>), dim3(L.chain), dim3(64), 0, stream, sorted_anchors, anchor_cap, kb, tasks, task_cap, groups,
                       group_count, group_cap, joins, join_count, join_cap, L.prio, (uint8_t *)nullptr);
    hipLaunchKernelGGL((kp_join_chain_kernel<JA_MAX, JA_SMALL>), dim3(L.chain_large), dim3(64), 0, stream, sorted_anchors, anchor_cap, kb, tasks, task_cap, groups,
                       group_count, group_cap, joins, join_count, join_cap, L.prio, scratch);
}

size_t kp_join_chain_scratch_bytes() { return (size_t)join_launch().chain_large * JOIN_CHAIN_SCRATCH; }

// The join kernels need nothing of the band tasks' fill and traceback: they run beside them, on a stream of their own
// (kp_capi.hip); the walk-back marks the band tasks whose hits a chain consumes (task_drop), which the hit compaction reads.
void kp_launch_join_fill(const KpBatchView &b, const KpGenes &genes, KpJoin *joins, const uint32_t *join_count, uint32_t join_cap,
                         void *trace, unsigned long long *trace_top, uint64_t trace_cap_units, hipStream_t stream) {
    const char *skip_env = std::getenv("KAPTIVE_AMD_SKIP_JOINS");
    if (skip_env && (std::atoi(skip_env) & 2)) return;
    const JoinLaunch &L = join_launch();
    hipLaunchKernelGGL(kp_join_fill_kernel, dim3(L.fill, KP_N_CLASSES), dim3(64), 0, stream, b, genes, joins, join_count, join_cap,
                       reinterpret_cast<uint4 *>(trace), trace_top, trace_cap_units, L.prio);
}

void kp_launch_join_trace(const KpBatchView &b, const KpGenes &genes, KpJoin *joins, const uint32_t *join_count, uint32_t join_cap,
                          uint32_t task_cap, const void *trace, uint8_t *task_drop, hipStream_t stream) {
    const char *skip_env = std::getenv("KAPTIVE_AMD_SKIP_JOINS");
    if (skip_env && (std::atoi(skip_env) & 4)) return;
    const JoinLaunch &L = join_launch();
    hipLaunchKernelGGL(kp_join_trace_kernel, dim3(L.walk, KP_N_CLASSES), dim3(64), 0, stream, b, genes, joins, join_count, join_cap, task_cap,
                       reinterpret_cast<const uint4 *>(trace), task_drop, L.prio);
}
