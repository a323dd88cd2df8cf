// kp_join.hip -- kp-align v4: chains of clusters across diagonal jumps of up to KP_JOIN_BW, and their joined alignment.
//
// minimap2 (which the reference's aligner wraps: src/kaptive/serotyping/core.py:147-155, docs/serotyping/method.md:23-28)
// chains anchors whose diagonals differ by up to bw = 500 and aligns through the gap, so a gene with an insertion or
// deletion of 33-500 bases is one hit there.  The band tasks of kp_chain.hip / kp_sw.hip stop at KP_DIAG_GAP = 32 diagonals;
// this file puts the join on top of them (include/kp_spec.h, "kp-align v4", is the specification; oracle/kp_oracle.c
// make_joins / join_run the CPU statement):
//
//   kp_join_chain_kernel   one wave per GROUP of provisional clusters (kp_chain.hip finds them): the lanes share the scan of
//                          each cluster's anchors for its first and last one, lane 0 runs minimap2's chaining DP on the
//                          accepted clusters and the backtracking: one KpJoin per chain of two or more
//   kp_join_fill_kernel    P lanes per join (band of 4P diagonals, the mapping of kp_sw.hip's 32-bit kernel): the pieces one
//                          after the other -- piece 0 as a local alignment, the later ones as CONTINUATIONS that only the
//                          cross gaps from the piece before can enter; what a piece offers the next one (per row or per
//                          column: the best H + e * position for both pieces of the gap cost) is collected with 64-bit
//                          atomic maxima in memory, direction bytes go to the trace buffer
//   kp_join_trace_kernel   one lane per join: walks back from the END of the last piece through the cross gaps, applies the
//                          drop test, writes the joined hit and flips the sign of the replaced band tasks' scores
//
// Joins are rare (a few per assembly at most on anything but constructed inputs): these kernels are written for clarity
// in plain 32-bit arithmetic, not for the vector pipe.
#include "kp_internal.h"

namespace {

__constant__ uint8_t c_join_pen[KP_CHAIN_PEN_SIZE] = KP_CHAIN_PEN_TABLE;

constexpr int JNEG = KP_NEG_INF;
__device__ __forceinline__ bool dead(int v) { return v < JNEG / 2; }
constexpr int XBIAS = 1 << 23;  // export keys: (value + e * position + XBIAS) in the high word -- values of a continuation can be negative

__device__ __forceinline__ int class_of_width(int w) { return w == 16 ? 0 : (w == 32 ? 1 : (w == 64 ? 2 : 3)); }

// ---- groups -> joins ---------------------------------------------------------------------------------------------------------------
struct JNode {
    int hq, ht, tq, tt, cs, cnt, ctg, qmax, lo, width, d0, dmax;
    uint32_t ref;
};

__global__ __launch_bounds__(256) void kp_join_chain_kernel(const uint64_t *__restrict__ keys, uint32_t cap, KpKeyBits kb,
                                                           const KpTask *__restrict__ tasks, uint32_t task_cap,
                                                           const KpGroup *__restrict__ groups, const uint32_t *__restrict__ group_count,
                                                           uint32_t group_cap, KpJoin *__restrict__ joins, uint32_t *__restrict__ join_count,
                                                           uint32_t join_cap) {
    uint32_t n_groups = *group_count;
    if (n_groups > group_cap) n_groups = group_cap;
    // one wave per group: the lanes share out the scan of a cluster's anchors (a thread on its own took 0.2 ms for two clusters
    // of 240 anchors: a chain of memory round trips), lane 0 chains the nodes
    const int lane = threadIdx.x & 63;
    for (uint32_t g = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6); g < n_groups; g += gridDim.x * (blockDim.x >> 6)) {
        const KpGroup &G = groups[g];
        JNode node[KP_JOIN_GROUP_MAX];
        int m = 0;
        for (int c = 0; c < G.n; ++c) {
            const uint32_t ref = G.task[c];
            if (KP_REF_SLOT(ref) >= task_cap) continue;  // (the list overflowed: the host reruns the pass)
            const KpTask t = tasks[(size_t)KP_REF_CLS(ref) * task_cap + KP_REF_SLOT(ref)];
            if (t.n_anchors == 0) continue;  // rejected by its chain score: not a node
            // head: the anchor with the smallest (query position, diagonal); tail: the one with the largest
            const uint64_t *k = keys + (size_t)G.asm_id * cap + G.first[c];
            uint64_t head = ~0ull, tail = 0ull;
            for (uint32_t i = (uint32_t)lane; i < G.cnt[c]; i += 64) {
                const uint64_t key = k[i];
                const uint64_t qd = ((uint64_t)kp_ckey_qpos(key, kb) << 32) | kp_ckey_diag(key, kb);
                if (qd < head) head = qd;
                if (qd > tail) tail = qd;
            }
#pragma unroll
            for (int o = 32; o >= 1; o >>= 1) {
                const uint64_t h2 = ((uint64_t)__shfl_xor((uint32_t)(head >> 32), o) << 32) | __shfl_xor((uint32_t)head, o);
                const uint64_t t2 = ((uint64_t)__shfl_xor((uint32_t)(tail >> 32), o) << 32) | __shfl_xor((uint32_t)tail, o);
                if (h2 < head) head = h2;
                if (t2 > tail) tail = t2;
            }
            JNode &N = node[m++];
            N.hq = (int)(head >> 32); N.ht = (int)(uint32_t)head - KP_DIAG_BIAS + N.hq;
            N.tq = (int)(tail >> 32); N.tt = (int)(uint32_t)tail - KP_DIAG_BIAS + N.tq;
            N.cs = t.chain_score; N.cnt = t.n_anchors; N.ctg = t.contig; N.qmax = (int)(t.qspan >> 16);
            N.lo = t.lo; N.width = t.width; N.ref = ref;
            N.d0 = (int)kp_ckey_diag(k[0], kb); N.dmax = (int)kp_ckey_diag(k[G.cnt[c] - 1], kb);  // (sorted by diagonal first)
        }
        if (lane != 0) continue;
        if (m < 2) continue;
        int ord[KP_JOIN_GROUP_MAX];
        for (int i = 0; i < m; ++i) {  // by (head t, head q, list order): stable insertion sort
            int j = i;
            while (j > 0 && (node[ord[j - 1]].ht > node[i].ht || (node[ord[j - 1]].ht == node[i].ht && node[ord[j - 1]].hq > node[i].hq))) { ord[j] = ord[j - 1]; --j; }
            ord[j] = i;
        }
        int f[KP_JOIN_GROUP_MAX], p[KP_JOIN_GROUP_MAX];
        bool used[KP_JOIN_GROUP_MAX];
        for (int i = 0; i < m; ++i) {
            const JNode &ci = node[ord[i]];
            int best = 0, bj = -1;
            for (int j = i - 1; j >= 0; --j) {
                const JNode &cj = node[ord[j]];
                if (cj.ctg != ci.ctg) continue;
                if (ci.d0 - cj.dmax <= KP_DIAG_GAP && cj.d0 - ci.dmax <= KP_DIAG_GAP) continue;  // one run of diagonals, cut in two by another contig's anchors
                const int dq = ci.hq - cj.tq, dr = ci.ht - cj.tt;
                if (dq <= 0 || dr <= 0 || dq > KP_CHAIN_MAX_DIST || dr > KP_CHAIN_MAX_DIST) continue;
                const int dd = dr > dq ? dr - dq : dq - dr;
                if (dd > KP_JOIN_BW) continue;
                const int dg = dr < dq ? dr : dq;
                const int link = (dg < KP_K ? dg : KP_K) - KP_K - (int)c_join_pen[dd];
                if (f[j] + link > best) { best = f[j] + link; bj = j; }
            }
            f[i] = ci.cs + best; p[i] = bj; used[i] = false;
        }
        for (;;) {
            int end = -1;
            for (int i = 0; i < m; ++i)
                if (!used[i] && (end < 0 || f[i] >= f[end])) end = i;
            if (end < 0) break;
            int chain[KP_JOIN_MAX_PIECES], len = 0, i = end;
            while (i >= 0 && !used[i] && len < KP_JOIN_MAX_PIECES) { chain[len++] = i; used[i] = true; i = p[i]; }
            const int score = f[end] - (i >= 0 ? f[i] : 0);
            if (len < 2 || score < KP_MIN_CHAIN_SCORE) continue;
            KpJoin J;
            const JNode &E = node[ord[end]];
            J.asm_id = G.asm_id; J.gs = tasks[(size_t)KP_REF_CLS(E.ref) * task_cap + KP_REF_SLOT(E.ref)].gs; J.contig = E.ctg;
            J.n_pieces = len; J.chain_score = score; J.n_anchors = 0; J.width = 0;
            for (int k = 0; k < len; ++k) {  // the walk went backwards: piece 0 is the last node walked
                const JNode &N = node[ord[chain[len - 1 - k]]];
                J.task[k] = N.ref; J.qmax[k] = N.qmax; J.n_anchors += N.cnt;
                if (N.width > J.width) J.width = N.width;
            }
            for (int k = 0; k < KP_JOIN_MAX_PIECES; ++k) {
                if (k < len) {
                    const JNode &N = node[ord[chain[len - 1 - k]]];
                    J.lo[k] = N.lo - (J.width - N.width) / 2;
                } else { J.task[k] = 0; J.qmax[k] = 0; J.lo[k] = 0; }
                J.trace_off[k] = J.export_off[k] = 0xFFFFFFFFu;
                J.end_s[k] = JNEG; J.end_r[k] = J.end_b[k] = -1;
                J.state[k] = 0; J.visited[k] = 0;
                for (int z = 0; z < 9; ++z) J.res[k][z] = 0;
            }
            const int cls = class_of_width(J.width);
            const uint32_t slot = atomicAdd(&join_count[cls], 1u);
            if (slot < join_cap) joins[(size_t)cls * join_cap + slot] = J;  // beyond cap: counted, not stored (host retries)
        }
    }
}

// ---- joined fill ------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ unsigned nib4(unsigned word, int i) { return (word >> (4 * i)) & 15u; }

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
            const bool cont = k > 0;
            const int none = cont ? JNEG : 0;
            int q0 = 0, r_hi = 0;
            if (act) kp_task_rows(lo, W, cstart, cend, qlen, &q0, &r_hi);
            const int steps = act ? (r_hi - q0) + P - 1 : 0;
            const int steps8 = (steps + 7) & ~7;
            const bool exports = act && k + 1 < n_pieces;
            const bool exp_horizontal = exports && J->lo[k + 1] > lo;
            const int exp_len = exports ? (exp_horizontal ? qlen : qlen + W) : 0;
            const int lo_next = exports ? J->lo[k + 1] : 0;
            const int rmin = act ? J->qmax[k] + KP_K - 1 : 0;
            // room: direction bytes (a word per lane and step) and the export array (two 64-bit keys per index)
            const unsigned long long t_units = (unsigned long long)steps8 * P / 4, x_units = (unsigned long long)exp_len;
            unsigned long long toff = 0;
            if (act && l == 0) toff = atomicAdd(trace_top, t_units + x_units);
            toff = ((unsigned long long)__shfl((unsigned)(toff >> 32), g * P) << 32) | __shfl((unsigned)toff, g * P);
            const bool fits = act && toff + t_units + x_units <= trace_cap && toff + t_units + x_units < 0xFFFFFFFFull;
            uint32_t *tr = reinterpret_cast<uint32_t *>(trace + toff);
            unsigned long long *exp = reinterpret_cast<unsigned long long *>(trace + toff + t_units);
            if (fits && exports) {
                for (int i = l; i < 2 * exp_len; i += P) exp[i] = 0ull;
                __threadfence();
            }
            int max_steps = fits ? steps8 : 0;
#pragma unroll
            for (int o = 32; o >= 1; o >>= 1) max_steps = max(max_steps, __shfl_xor(max_steps, o));
            int H[4], E[4], F[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) { H[c] = none; E[c] = F[c] = JNEG; }
            int best = JNEG, best_r = -1, best_b = 4 * l;
            // Eight steps at a time: what they read -- the query rows, the target bases and the keys of the cross gaps -- does not
            // depend on the cells before them, so it is all requested up front and the eight dependent steps then run from
            // registers (a step that fetched its own operands waited for memory three times: 3.3 ms for a 1300-row piece).
            for (int m0 = 0; m0 < max_steps; m0 += 8) {
                const int r0 = q0 + m0 - l;  // this lane's row at the chunk's first step
                const int tb0 = lo + r0 + 4 * l;  // ... and the column of its cell 0 there: step s, cell c sits on tb0 + s + c
                uint64_t qwin, twin;
                {
                    const int w0 = r0 >> 3, nq = (qlen + 7) >> 3;  // (arithmetic shifts: rows before the gene read as N)
                    const uint32_t lo_w = (fits && w0 >= 0 && w0 < nq) ? qnib[w0] : 0x44444444u;
                    const uint32_t hi_w = (fits && w0 + 1 >= 0 && w0 + 1 < nq) ? qnib[w0 + 1] : 0x44444444u;
                    qwin = (((uint64_t)hi_w << 32) | lo_w) >> (4 * (r0 & 7));
                    const int v0 = tb0 >> 4;
                    const uint32_t lo_t = (fits && v0 >= 0 && v0 < asm_n_words) ? asm_words[v0] : 0u;
                    const uint32_t hi_t = (fits && v0 + 1 >= 0 && v0 + 1 < asm_n_words) ? asm_words[v0 + 1] : 0u;
                    twin = (((uint64_t)hi_t << 32) | lo_t) >> (2 * (tb0 & 15));  // eleven bases: 22 of the 34 bits that are left
                }
                unsigned long long kx1[11], kx2[11];  // cross-gap keys: of rows r0 + s (horizontal) or of columns tb0 + j (vertical)
#pragma unroll
                for (int j = 0; j < 11; ++j) {
                    kx1[j] = kx2[j] = 0ull;
                    if (fits && cont && imp) {
                        const int xi = imp_horizontal ? r0 + j : tb0 + j - imp_lo;
                        if ((!imp_horizontal || j < 8) && xi >= 0 && xi < imp_len) {
                            kx1[j] = __hip_atomic_load(&imp[2 * xi], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            kx2[j] = __hip_atomic_load(&imp[2 * xi + 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        }
                    }
                }
#pragma unroll
                for (int sidx = 0; sidx < 8; ++sidx) {
                    const int m = m0 + sidx, r = r0 + sidx;
                    const bool row_ok = fits && m < steps8 && r >= q0 && r < r_hi;
                    const int qc = (int)((qwin >> (4 * sidx)) & 15u);
                    // left neighbour of cell 0: lane l - 1's cell 3 as the previous step left it; upper neighbour of cell 3: lane
                    // l + 1's cell 0 of this step
                    int hl = __shfl_up(H[3], 1), el = __shfl_up(E[3], 1);
                    if (l == 0) { hl = none; el = JNEG; }
                    const int oldH[4] = {H[0], H[1], H[2], H[3]}, oldF[4] = {F[0], F[1], F[2], F[3]};
                    uint32_t word = 0;
                    int hu_d = none, fu_d = JNEG;
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        if (c == 3) {  // (every lane has computed its cell 0 by now)
                            hu_d = __shfl_down(H[0], 1); fu_d = __shfl_down(F[0], 1);
                            if (l == P - 1) { hu_d = none; fu_d = JNEG; }
                        }
                        const int t = tb0 + sidx + c;
                        int code = 5;  // 0..3, 4 = N, 5 = outside the contig
                        if (row_ok && t >= cstart && t < cend) {
                            code = (int)((twin >> (2 * (sidx + c))) & 3u);
                            if (n_runs > 0) {  // (rare: assemblies with scaffold gaps)
                                int a2 = 0, z2 = n_runs;
                                while (a2 < z2) {
                                    const int mid = (a2 + z2) >> 1;
                                    if (runs[2 * mid + 1] <= t) a2 = mid + 1; else z2 = mid;
                                }
                                if (a2 < n_runs && runs[2 * a2] <= t) code = 4;
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
                        if (inside && cont && imp) {
                            const unsigned long long k1 = imp_horizontal ? kx1[sidx] : kx1[sidx + c], k2 = imp_horizontal ? kx2[sidx] : kx2[sidx + c];
                            const int pos = imp_horizontal ? t - imp_lo : r;  // (the exporter's frame: columns count from its band's origin)
                            if (k1) {  // (a row / column with one key has both)
                                const int c1 = (int)(k1 >> 32) - XBIAS - KP_GAP_OPEN - KP_GAP_EXT * pos;
                                const int c2 = (int)(k2 >> 32) - XBIAS - KP_GAP_OPEN2 - KP_GAP_EXT2 * pos;
                                if (c1 > bv) { bv = c1; tb = XT_X1; }
                                if (c2 > bv) { bv = c2; tb = XT_X2; }
                            }
                        }
                        const bool live = inside && (cont ? !dead(bv) : bv > 0);
                        if (inside) { E[c] = e; F[c] = f; } else { E[c] = F[c] = JNEG; }
                        H[c] = live ? bv : none;
                        word |= ((live ? tb : (uint32_t)XT_RESTART) | (e_extd << 3) | (f_extd << 4)) << (8 * c);
                        if (live) {
                            if (cont && r >= rmin && bv > best) { best = bv; best_r = r; best_b = 4 * l + c; }
                            if (exports && (exp_horizontal ? (4 * l + c < lo_next - lo) : (lo + 4 * l + c > lo_next + W - 1))) {
                                const int xi = exp_horizontal ? r : t - lo, pos = exp_horizontal ? t - lo : r;
                                const unsigned long long low = 0xFFFFFFFFull - (unsigned)pos;
                                atomicMax(&exp[2 * xi], ((unsigned long long)(unsigned)(bv + KP_GAP_EXT * pos + XBIAS) << 32) | low);
                                atomicMax(&exp[2 * xi + 1], ((unsigned long long)(unsigned)(bv + KP_GAP_EXT2 * pos + XBIAS) << 32) | low);
                            }
                        }
                    }
                    if (fits && m < steps8) tr[(size_t)m * P + l] = word;
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
                J->end_s[k] = fits && best_r >= 0 ? best : JNEG; J->end_r[k] = best_r; J->end_b[k] = best_b;
            }
            __threadfence();  // this piece's exports are in memory before the next piece reads them
            imp = fits && exports ? exp : nullptr;
            imp_horizontal = exp_horizontal; imp_len = exp_len; imp_lo = lo;
        }
    }
}

__global__ __launch_bounds__(64) void kp_join_fill_kernel(KpBatchView b, KpGenes genes, KpJoin *__restrict__ joins,
                                                          const uint32_t *__restrict__ join_count, uint32_t join_cap,
                                                          uint4 *__restrict__ trace, unsigned long long *__restrict__ trace_top,
                                                          uint64_t trace_cap) {
    const int c = blockIdx.y;
    uint32_t n = join_count[c];
    if (n > join_cap) n = join_cap;
    if (n == 0) return;
    KpJoin *list = joins + (size_t)c * join_cap;
    if (c == 3) join_fill_class<32>(b, genes, list, n, trace, trace_top, trace_cap, blockIdx.x, gridDim.x);
    else if (c == 2) join_fill_class<16>(b, genes, list, n, trace, trace_top, trace_cap, blockIdx.x, gridDim.x);
    else if (c == 1) join_fill_class<8>(b, genes, list, n, trace, trace_top, trace_cap, blockIdx.x, gridDim.x);
    else join_fill_class<4>(b, genes, list, n, trace, trace_top, trace_cap, blockIdx.x, gridDim.x);
}

// ---- walk-back: one lane per join -------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void kp_join_trace_kernel(KpBatchView b, KpGenes genes, KpJoin *__restrict__ joins,
                                                           const uint32_t *__restrict__ join_count, uint32_t join_cap,
                                                           uint32_t task_cap, const uint4 *__restrict__ trace,
                                                           KpSwResult *__restrict__ results) {
    const int cls = blockIdx.y;
    uint32_t n = join_count[cls];
    if (n > join_cap) n = join_cap;
    const int P = 4 << cls, W = 4 * P;
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
        int on_path = 0;
        for (int k = m - 1; k >= 1; --k) {
            J->state[k] = 0; J->visited[k] = 0;
            if (!complete || ((on_path >> k) & 1)) continue;
            if (J->end_r[k] < 0 || J->end_s[k] < KP_MIN_DP_SCORE) continue;
            int pk = k, r = J->end_r[k], bi = J->end_b[k], state = 0, matches = 0, cols = 0, gap = 0, credit = 0;
            int sr = r, sb = bi, spk = k, suf = 0, sufmax = 0, visited = 1 << k, bonus = 0;
            bool rejected = false;
            int lo = J->lo[pk], q0 = 0, r_hi = 0;
            kp_task_rows(lo, W, cstart, cend, qlen, &q0, &r_hi);
            const uint32_t *tr = reinterpret_cast<const uint32_t *>(trace + J->trace_off[pk]);
            for (;;) {
                const int t = r + lo + bi;
                if (state == 0 && (r < q0 || r >= r_hi || bi < 0 || bi >= W || t < cstart || t >= cend)) break;
                const uint32_t byte = (tr[(size_t)(r - q0 + (bi >> 2)) * P + (bi >> 2)] >> (8 * (bi & 3))) & 255u;
                if (state == 0) {
                    const uint32_t tb = byte & 7u;
                    if (tb == XT_RESTART) break;
                    if (suf > sufmax) sufmax = suf;
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
                        if (sufmax - suf > KP_JOIN_DROP) { rejected = true; break; }
                        const int lo_prev = J->lo[pk - 1];
                        const bool horizontal = lo > lo_prev;
                        const unsigned long long *exp = reinterpret_cast<const unsigned long long *>(trace + J->export_off[pk - 1]);
                        const int xi = horizontal ? r : t - lo_prev;
                        const unsigned long long key = exp[2 * xi + (tb == XT_X1 ? 0 : 1)];
                        const int pos = (int)(0xFFFFFFFFu - (uint32_t)key);  // t' - lo_prev (horizontal) or r'
                        const int ngap = horizontal ? (t - lo_prev) - pos : r - pos;
                        cols += ngap;
                        const int cost = tb == XT_X1 ? KP_GAP_OPEN + KP_GAP_EXT * ngap : KP_GAP_OPEN2 + KP_GAP_EXT2 * ngap;
                        suf -= cost;
                        const int lg = KP_GAP_OPEN + kp_log2x2((uint32_t)ngap);
                        if (cost > lg) bonus += cost - lg;
                        if (horizontal) bi = pos - r;               // same row, column lo_prev + pos
                        else { bi = t - pos - lo_prev; r = pos; }   // same column, row pos
                        --pk; visited |= 1 << pk;
                        lo = lo_prev;
                        kp_task_rows(lo, W, cstart, cend, qlen, &q0, &r_hi);
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
            if (rejected) { J->state[k] = 2; continue; }
            J->state[k] = 1;
            on_path |= visited;
            J->res[k][0] = J->end_s[k]; J->res[k][1] = sr; J->res[k][2] = J->end_r[k] + 1;
            J->res[k][3] = sr + J->lo[spk] + sb; J->res[k][4] = J->end_r[k] + J->lo[k] + J->end_b[k] + 1;
            J->res[k][5] = matches; J->res[k][6] = cols; J->res[k][7] = J->end_s[k] + credit;
            J->res[k][8] = bonus < KP_HIT_BONUS_MAX ? bonus : KP_HIT_BONUS_MAX;
            for (int v = 0; v < m; ++v)  // the band tasks of the visited pieces no longer report a hit of their own
                if ((visited >> v) & 1) {
                    KpSwResult &R = results[(size_t)KP_REF_CLS(J->task[v]) * task_cap + KP_REF_SLOT(J->task[v])];
                    if (R.score > 0) R.score = -R.score;
                }
        }
    }
}

}  // namespace

void kp_launch_join_chain(const KpBatchView &b, const KpGenes &genes, const uint64_t *sorted_anchors, uint32_t anchor_cap, KpKeyBits kb,
                          const KpTask *tasks, uint32_t task_cap, const KpGroup *groups, const uint32_t *group_count, uint32_t group_cap,
                          KpJoin *joins, uint32_t *join_count, uint32_t join_cap, hipStream_t stream) {
    (void)b; (void)genes;
    hipLaunchKernelGGL(kp_join_chain_kernel, dim3(64), dim3(256), 0, stream, sorted_anchors, anchor_cap, kb, tasks, task_cap, groups,
                       group_count, group_cap, joins, join_count, join_cap);
}

void kp_launch_join_sw(const KpBatchView &b, const KpGenes &genes, KpJoin *joins, const uint32_t *join_count, uint32_t join_cap,
                       uint32_t task_cap, void *trace, unsigned long long *trace_top, uint64_t trace_cap_units, KpSwResult *results,
                       hipStream_t stream) {
    hipLaunchKernelGGL(kp_join_fill_kernel, dim3(128, KP_N_CLASSES), dim3(64), 0, stream, b, genes, joins, join_count, join_cap,
                       reinterpret_cast<uint4 *>(trace), trace_top, trace_cap_units);
    hipLaunchKernelGGL(kp_join_trace_kernel, dim3(32, KP_N_CLASSES), dim3(64), 0, stream, b, genes, joins, join_count, join_cap, task_cap,
                       reinterpret_cast<const uint4 *>(trace), results);
}
