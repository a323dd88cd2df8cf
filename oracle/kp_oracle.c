/* kp_oracle.c -- CPU restatement of the typing hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * Nothing in the product (kaptive_amd/) may include, link or call this file; only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg use it, as the checker.  Plain C, scalar, single-threaded, written for clarity.
 *
 * What it restates, and how each part is pinned:
 *   kpo_protein_align   reference src/kaptive/core/pairwise.py:395-584 (_batched_banded_gotoh, unseeded mode), with
 *                       the BLOSUM62 lookup of pairwise.py:343-391.  PINNED against tests/golden/protein_dp.npz,
 *                       produced by running the reference itself (oracle/make_golden.py).
 *   kpo_protein_align_seeded  the same kernel's seeded mode (pairwise.py:449-451).  PINNED against
 *                       tests/golden/compare.npz (the reference's PairwiseAligner.align_seeds on randstrobe seeds).
 *   kpo_cull_overlaps   reference src/kaptive/core/interval.py:698-751.          PINNED (tests/golden/intervals.npz)
 *   kpo_cluster         reference src/kaptive/core/interval.py:595-639.          PINNED (tests/golden/intervals.npz)
 *   kpo_translate       reference src/kaptive/core/seq.py:671-741.               PINNED (tests/golden/seqs.npz)
 *   kpo_extract         reference src/kaptive/core/seq.py:612-668.               PINNED (tests/golden/seqs.npz)
 *   kpo_align (+ stage entry points kpo_anchors / kpo_tasks / kpo_sw)
 *                       the gene-vs-contig aligner.  In the reference this stage is the third-party rammappy 0.1.3
 *                       wheel (call sites src/kaptive/core/genome.py:188-189, src/kaptive/serotyping/core.py:147-155;
 *                       consumer src/kaptive/core/alignment.py:409-446); its source is not in the reference tree and
 *                       no reference test pins any alignment.  PARITY UNPINNED for this stage: the specification is
 *                       include/kp_spec.h (kp-align v3: minimap2's (10, 15) minimizer sampling, chain-score test and
 *                       mapping quality restated), and this file is its executable statement.  Its seeds are checked
 *                       against an independent restatement of mm_sketch (oracle/mm2_model.c) and its report rows
 *                       against that model's (tests/test_mm2_concordance.py, profiles/concordance_r4.md).
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <math.h>

#include "../include/kp_spec.h"
#include "../include/kp_mapq.h"

#define KPO_API __attribute__((visibility("default")))

/* ================================================================================================================
 * Protein banded Smith-Waterman-Gotoh with traceback (pairwise.py:395-584)
 * ================================================================================================================ */

static int8_t g_blosum[256][256];
static int g_blosum_ready = 0;

static void blosum_init(void) {
    /* rows/cols in the order ARNDCQEGHILKMFPSTWYVBJZX*  (pairwise.py:353-383) */
    static const int8_t m[25][25] = {
        {4, -1, -2, -2, 0, -1, -1, 0, -2, -1, -1, -1, -1, -2, -1, 1, 0, -3, -2, 0, -2, -1, -1, -1, -4},
        {-1, 5, 0, -2, -3, 1, 0, -2, 0, -3, -2, 2, -1, -3, -2, -1, -1, -3, -2, -3, -1, -2, 0, -1, -4},
        {-2, 0, 6, 1, -3, 0, 0, 0, 1, -3, -3, 0, -2, -3, -2, 1, 0, -4, -2, -3, 4, -3, 0, -1, -4},
        {-2, -2, 1, 6, -3, 0, 2, -1, -1, -3, -4, -1, -3, -3, -1, 0, -1, -4, -3, -3, 4, -3, 1, -1, -4},
        {0, -3, -3, -3, 9, -3, -4, -3, -3, -1, -1, -3, -1, -2, -3, -1, -1, -2, -2, -1, -3, -1, -3, -1, -4},
        {-1, 1, 0, 0, -3, 5, 2, -2, 0, -3, -2, 1, 0, -3, -1, 0, -1, -2, -1, -2, 0, -2, 4, -1, -4},
        {-1, 0, 0, 2, -4, 2, 5, -2, 0, -3, -3, 1, -2, -3, -1, 0, -1, -3, -2, -2, 1, -3, 4, -1, -4},
        {0, -2, 0, -1, -3, -2, -2, 6, -2, -4, -4, -2, -3, -3, -2, 0, -2, -2, -3, -3, -1, -4, -2, -1, -4},
        {-2, 0, 1, -1, -3, 0, 0, -2, 8, -3, -3, -1, -2, -1, -2, -1, -2, -2, 2, -3, 0, -3, 0, -1, -4},
        {-1, -3, -3, -3, -1, -3, -3, -4, -3, 4, 2, -3, 1, 0, -3, -2, -1, -3, -1, 3, -3, 3, -3, -1, -4},
        {-1, -2, -3, -4, -1, -2, -3, -4, -3, 2, 4, -2, 2, 0, -3, -2, -1, -2, -1, 1, -4, 3, -3, -1, -4},
        {-1, 2, 0, -1, -3, 1, 1, -2, -1, -3, -2, 5, -1, -3, -1, 0, -1, -3, -2, -2, 0, -3, 1, -1, -4},
        {-1, -1, -2, -3, -1, 0, -2, -3, -2, 1, 2, -1, 5, 0, -2, -1, -1, -1, -1, 1, -3, 2, -1, -1, -4},
        {-2, -3, -3, -3, -2, -3, -3, -3, -1, 0, 0, -3, 0, 6, -4, -2, -2, 1, 3, -1, -3, 0, -3, -1, -4},
        {-1, -2, -2, -1, -3, -1, -1, -2, -2, -3, -3, -1, -2, -4, 7, -1, -1, -4, -3, -2, -2, -3, -1, -1, -4},
        {1, -1, 1, 0, -1, 0, 0, 0, -1, -2, -2, 0, -1, -2, -1, 4, 1, -3, -2, -2, 0, -2, 0, -1, -4},
        {0, -1, 0, -1, -1, -1, -1, -2, -2, -1, -1, -1, -1, -2, -1, 1, 5, -2, -2, 0, -1, -1, -1, -1, -4},
        {-3, -3, -4, -4, -2, -2, -3, -2, -2, -3, -2, -3, -1, 1, -4, -3, -2, 11, 2, -3, -4, -2, -2, -1, -4},
        {-2, -2, -2, -3, -2, -1, -2, -3, 2, -1, -1, -2, -1, 3, -3, -2, -2, 2, 7, -1, -3, -1, -2, -1, -4},
        {0, -3, -3, -3, -1, -2, -2, -3, -3, 3, 1, -2, 1, -1, -2, -2, 0, -3, -1, 4, -3, 2, -2, -1, -4},
        {-2, -1, 4, 4, -3, 0, 1, -1, 0, -3, -4, 0, -3, -3, -2, 0, -1, -4, -3, -3, 4, -3, 0, -1, -4},
        {-1, -2, -3, -3, -1, -2, -3, -4, -3, 3, 3, -3, 2, 0, -3, -2, -1, -2, -1, 2, -3, 3, -3, -1, -4},
        {-1, 0, 0, 1, -3, 4, 4, -2, 0, -3, -3, 1, -1, -3, -1, 0, -1, -2, -2, -2, 0, -3, 4, -1, -4},
        {-1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -4},
        {-4, -4, -4, -4, -4, -4, -4, -4, -4, -4, -4, -4, -4, -4, -4, -4, -4, -4, -4, -4, -4, -4, -4, -4, 1},
    };
    static const char alphabet[] = "ARNDCQEGHILKMFPSTWYVBJZX*";
    memset(g_blosum, KP_PROT_FILL, sizeof g_blosum);
    for (int a = 0; a < 25; a++)
        for (int b = 0; b < 25; b++) g_blosum[(uint8_t)alphabet[a]][(uint8_t)alphabet[b]] = m[a][b];
    g_blosum_ready = 1;
}

KPO_API void kpo_blosum62(int8_t *out /* [256*256] */) {
    if (!g_blosum_ready) blosum_init();
    memcpy(out, g_blosum, sizeof g_blosum);
}

static inline int imax(int a, int b) { return a > b ? a : b; }
static inline int imin(int a, int b) { return a < b ? a : b; }

/* One pair; follows the reference loop structure (band stored as rows x (2k+3) with per-row origin). */
/* seeded != 0: the band is k diagonals either side of the seed diagonal (centre column i - offset, pairwise.py:449-451);
 * otherwise it is centred on 0 and absorbs the length difference (pairwise.py:452-454) */
static void protein_pair(const uint8_t *s1, int len1, const uint8_t *s2, int len2, int k, int go, int ge, int seeded,
                         int offset, int32_t *out8 /* score, matches, mismatches, gaps, qs, qe, ts, te */) {
    const int rows = len1 + 1, cols = len2 + 1;
    int d = len1 - len2;
    if (d < 0) d = -d;
    const int kl = seeded ? k : imax(k, d + 1);
    if (!seeded) offset = 0;
    const int bw = 2 * kl + 3;
    const size_t cells = (size_t)rows * (size_t)bw;
    int32_t *M = malloc(cells * sizeof(int32_t)), *I = malloc(cells * sizeof(int32_t)),
            *D = malloc(cells * sizeof(int32_t));
    uint8_t *tM = malloc(cells), *tD = malloc(cells), *tI = malloc(cells);
#define AT(i, jm) ((size_t)(i) * (size_t)bw + (size_t)(jm))
    for (int i = 0; i < rows; i++) { /* band-only initialisation, pairwise.py:466-479 */
        int sj = imax(0, i - offset - kl - 1), ej = imin(cols, i - offset + kl + 2);
        if (sj >= cols || ej <= 0) continue;
        for (int j = sj; j < ej; j++) {
            int jm = j - sj;
            M[AT(i, jm)] = 0;
            I[AT(i, jm)] = KP_PROT_NEG_INF;
            D[AT(i, jm)] = KP_PROT_NEG_INF;
            tM[AT(i, jm)] = 3;
        }
    }
    int max_score = 0, max_i = 0, max_j = 0;
    for (int i = 1; i < rows; i++) { /* fill, pairwise.py:486-535 */
        int sj = imax(1, i - offset - kl), ej = imin(cols, i - offset + kl + 1);
        if (sj >= cols || ej <= 1) continue;
        int sp = imax(0, i - 1 - offset - kl - 1), sc = imax(0, i - offset - kl - 1);
        for (int j = sj; j < ej; j++) {
            int jt = j - sp, jm = j - sc, jl = j - 1 - sc, jd = j - 1 - sp;
            int d_open = M[AT(i - 1, jt)] - go - ge, d_ext = D[AT(i - 1, jt)] - ge;
            if (d_open >= d_ext) { D[AT(i, jm)] = d_open; tD[AT(i, jm)] = 0; }
            else { D[AT(i, jm)] = d_ext; tD[AT(i, jm)] = 1; }
            int i_open = M[AT(i, jl)] - go - ge, i_ext = I[AT(i, jl)] - ge;
            if (i_open >= i_ext) { I[AT(i, jm)] = i_open; tI[AT(i, jm)] = 0; }
            else { I[AT(i, jm)] = i_ext; tI[AT(i, jm)] = 2; }
            int best = M[AT(i - 1, jd)] + g_blosum[s1[i - 1]][s2[j - 1]], tb = 0;
            if (D[AT(i, jm)] > best) { best = D[AT(i, jm)]; tb = 1; }
            if (I[AT(i, jm)] > best) { best = I[AT(i, jm)]; tb = 2; }
            if (best <= 0) { M[AT(i, jm)] = 0; tM[AT(i, jm)] = 3; }
            else {
                M[AT(i, jm)] = best; tM[AT(i, jm)] = (uint8_t)tb;
                if (best > max_score) { max_score = best; max_i = i; max_j = j; }
            }
        }
    }
    int i = max_i, j = max_j, matches = 0, mism = 0, gaps = 0, state = 0; /* traceback, pairwise.py:538-572 */
    while (i > 0 && j > 0) {
        int sc = imax(0, i - offset - kl - 1), jm = j - sc;
        if (state == 0) {
            int tb = tM[AT(i, jm)];
            if (tb == 3) break;
            if (tb == 0) { if (s1[i - 1] == s2[j - 1]) matches++; else mism++; i--; j--; }
            else state = tb;
        } else if (state == 1) { int tb = tD[AT(i, jm)]; gaps++; i--; if (tb == 0) state = 0; }
        else { int tb = tI[AT(i, jm)]; gaps++; j--; if (tb == 0) state = 0; }
    }
#undef AT
    out8[0] = max_score; out8[1] = matches; out8[2] = mism; out8[3] = gaps;
    out8[4] = i; out8[5] = max_i; out8[6] = j; out8[7] = max_j;
    free(M); free(I); free(D); free(tM); free(tD); free(tI);
}

KPO_API void kpo_protein_align(const uint8_t *q, const int32_t *q_off, const int32_t *q_len, const uint8_t *t,
                               const int32_t *t_off, const int32_t *t_len, int n, int32_t *out /* [n][8] */) {
    if (!g_blosum_ready) blosum_init();
    for (int x = 0; x < n; x++)
        protein_pair(q + q_off[x], q_len[x], t + t_off[x], t_len[x], KP_PROT_K, KP_PROT_GAP_OPEN, KP_PROT_GAP_EXT, 0, 0,
                     out + 8 * (size_t)x);
}

/* the seeded mode (PairwiseAligner.align_seeds, pairwise.py:327-339): one diagonal offset per pair, band k */
KPO_API void kpo_protein_align_seeded(const uint8_t *q, const int32_t *q_off, const int32_t *q_len, const uint8_t *t,
                                      const int32_t *t_off, const int32_t *t_len, int n, const int32_t *offsets, int k,
                                      int32_t *out /* [n][8] */) {
    if (!g_blosum_ready) blosum_init();
    for (int x = 0; x < n; x++)
        protein_pair(q + q_off[x], q_len[x], t + t_off[x], t_len[x], k, KP_PROT_GAP_OPEN, KP_PROT_GAP_EXT, 1, offsets[x],
                     out + 8 * (size_t)x);
}

/* ================================================================================================================
 * Interval reductions (interval.py:595-639, 698-751) and ragged sequence kernels (seq.py:612-741)
 * ================================================================================================================ */

KPO_API void kpo_cull_overlaps(const int32_t *order, const int32_t *g1, const int32_t *g2, const int32_t *starts,
                               const int32_t *ends, double max_frac, int n, uint8_t *kept) {
    memset(kept, 0, (size_t)n);
    for (int i = 0; i < n; i++) {
        int idx = order[i], s = starts[idx], e = ends[idx], len = e - s;
        if (len <= 0) continue;
        int clash = 0;
        for (int j = 0; j < i && !clash; j++) {
            int p = order[j];
            if (!kept[p] || g1[p] != g1[idx] || g2[p] != g2[idx]) continue;
            int ov = imin(e, ends[p]) - imax(s, starts[p]);
            if (ov > 0 && (double)ov / (double)imin(len, ends[p] - starts[p]) > max_frac) clash = 1;
        }
        if (!clash) kept[idx] = 1;
    }
}

KPO_API void kpo_cluster(const int32_t *starts, const int32_t *ends, const int32_t *groups, int64_t tolerance,
                         const int32_t *order, int n, int32_t *ids) {
    if (n == 0) return;
    int cur = 0, f = order[0];
    int64_t cur_e = ends[f];
    int32_t cur_g = groups[f];
    ids[f] = 0;
    for (int i = 1; i < n; i++) {
        int idx = order[i];
        if (groups[idx] == cur_g && (int64_t)starts[idx] <= cur_e + tolerance) {
            if (ends[idx] > cur_e) cur_e = ends[idx];
        } else { cur++; cur_e = ends[idx]; cur_g = groups[idx]; }
        ids[idx] = cur;
    }
}

static uint8_t g_code[256], g_comp[256], g_codon[125];
static int g_tables_ready = 0;

static void tables_init(void) {
    memset(g_code, 4, sizeof g_code);
    const char *acgt = "ACGT";
    for (int i = 0; i < 4; i++) { g_code[(uint8_t)acgt[i]] = (uint8_t)i; g_code[(uint8_t)acgt[i] + 32] = (uint8_t)i; }
    g_code['U'] = g_code['u'] = 3;
    for (int i = 0; i < 256; i++) g_comp[i] = (uint8_t)i;
    const char *from = "ACGTUacgtu", *to = "TGCAAtgcaa";
    for (int i = 0; i < 10; i++) g_comp[(uint8_t)from[i]] = (uint8_t)to[i];
    /* table 11 in TCAG order, re-indexed to ACGT radix-5 (seq.py:418-499) */
    const char *aa = "FFLLSSSSYY**CC*WLLLLPPPPHHQQRRRRIIIMTTTTNNKKSSRRVVVVAAAADDEEGGGG";
    const int tcag[4] = {3, 1, 0, 2};
    memset(g_codon, 'X', sizeof g_codon);
    for (int a = 0; a < 4; a++)
        for (int b = 0; b < 4; b++)
            for (int c = 0; c < 4; c++) g_codon[tcag[a] * 25 + tcag[b] * 5 + tcag[c]] = (uint8_t)aa[a * 16 + b * 4 + c];
    g_tables_ready = 1;
}

/* returns total output length; out may be NULL to size */
KPO_API int64_t kpo_translate(const uint8_t *seqs, const int32_t *off, const int32_t *len, const int8_t *frames,
                              int n, int to_stop, uint8_t *out, int32_t *out_off, int32_t *out_len) {
    if (!g_tables_ready) tables_init();
    int64_t total = 0;
    for (int i = 0; i < n; i++) {
        int f = frames ? frames[i] : 0, nc = 0;
        if (len[i] > f) {
            int adj = len[i] - f, maxc = adj >= 3 ? adj / 3 : 0;
            const uint8_t *p = seqs + off[i] + f;
            for (int c = 0; c < maxc; c++, p += 3) {
                uint8_t a = g_codon[g_code[p[0]] * 25 + g_code[p[1]] * 5 + g_code[p[2]]];
                if (to_stop && a == 42) break;
                if (out) out[total + nc] = a;
                nc++;
            }
        }
        out_off[i] = (int32_t)total;
        out_len[i] = nc;
        total += nc;
    }
    return total;
}

KPO_API int64_t kpo_extract(const uint8_t *seqs, const int32_t *off, const int32_t *idx, const int32_t *starts,
                            const int32_t *ends, const int8_t *strands, int n, uint8_t *out, int32_t *out_off,
                            int32_t *out_len) {
    if (!g_tables_ready) tables_init();
    int64_t total = 0;
    for (int i = 0; i < n; i++) {
        int l = ends[i] - starts[i];
        out_off[i] = (int32_t)total;
        out_len[i] = l;
        if (out) {
            int64_t gs = (int64_t)off[idx[i]] + starts[i], ge = (int64_t)off[idx[i]] + ends[i];
            if (strands[i] >= 0) for (int c = 0; c < l; c++) out[total + c] = seqs[gs + c];
            else for (int c = 0; c < l; c++) out[total + c] = g_comp[seqs[ge - 1 - c]];
        }
        total += l > 0 ? l : 0;
    }
    return total;
}

/* ================================================================================================================
 * Nucleotide aligner "kp-align v3" (include/kp_spec.h) -- parity UNPINNED against the reference (see header)
 * ================================================================================================================ */

/* ---- seeds: minimap2's (10, 15) minimizers, kp_spec.h's state machine written out step by step ----------------------------
 * One call handles one sequence of codes (0..3, anything else ambiguous).  Every seed is reported once through `emit`
 * as (start of the 15-mer, strand bit z, x).  The ring `ring_x/ring_p/ring_z` holds the last KP_W steps. */
#define KPO_INF 0xFFFFFFFFu
typedef void (*kpo_seed_fn)(void *ud, int32_t start, int z, uint32_t x);

static void kpo_sketch(const uint8_t *c, int64_t len, kpo_seed_fn emit, void *ud) {
    uint32_t ring_x[KP_W];
    int32_t ring_p[KP_W];
    uint8_t ring_z[KP_W];
    for (int j = 0; j < KP_W; j++) { ring_x[j] = KPO_INF; ring_p[j] = -1; ring_z[j] = 0; }
    uint32_t fwd = 0, rev = 0, min_x = KPO_INF;
    int32_t min_p = -1;
    int min_z = 0, min_slot = 0, slot = 0;
    int64_t run = 0; /* l(i): unambiguous bases ending here */
    for (int64_t i = 0; i < len; i++) {
        uint32_t x = KPO_INF;
        int z = 0;
        if (c[i] < 4) {
            fwd = ((fwd << 2) | c[i]) & KP_KMER_MASK;
            rev = (rev >> 2) | ((uint32_t)(3 - c[i]) << (2 * (KP_K - 1)));
            run++;
            if (run >= KP_K) { z = fwd < rev ? 0 : 1; x = kp_hash30(z ? rev : fwd); }
        } else run = 0;
        const int32_t p = (int32_t)(i - (KP_K - 1));
        ring_x[slot] = x; ring_p[slot] = p; ring_z[slot] = (uint8_t)z;
        if (run == KP_W + KP_K - 1 && min_x != KPO_INF) /* (1) ties of the first full window, oldest first */
            for (int d = 1; d < KP_W; d++) {
                const int j = (slot + d) % KP_W;
                if (ring_x[j] == min_x && ring_p[j] != min_p) emit(ud, ring_p[j], ring_z[j], ring_x[j]);
            }
        if (x <= min_x) { /* (2) a new minimum takes over */
            if (run >= KP_W + KP_K && min_x != KPO_INF) emit(ud, min_p, min_z, min_x);
            min_x = x; min_p = p; min_z = z; min_slot = slot;
        } else if (slot == min_slot) { /* this step overwrote the minimum: it has left the window */
            if (run >= KP_W + KP_K - 1 && min_x != KPO_INF) emit(ud, min_p, min_z, min_x);
            min_x = KPO_INF;
            for (int d = 1; d <= KP_W; d++) { /* oldest to newest: the last smallest wins */
                const int j = (slot + d) % KP_W;
                if (ring_x[j] <= min_x) { min_x = ring_x[j]; min_p = ring_p[j]; min_z = ring_z[j]; min_slot = j; }
            }
            if (run >= KP_W + KP_K - 1 && min_x != KPO_INF)
                for (int d = 1; d <= KP_W; d++) {
                    const int j = (slot + d) % KP_W;
                    if (ring_x[j] == min_x && j != min_slot) emit(ud, ring_p[j], ring_z[j], ring_x[j]);
                }
        }
        slot = (slot + 1) % KP_W;
    }
    if (min_x != KPO_INF) emit(ud, min_p, min_z, min_x);
}

typedef struct { uint32_t key; uint32_t gene; uint32_t pos; uint32_t z; } kpo_posting; /* key = x of the gene seed */

typedef struct kpo_db {
    int n_genes;
    const uint8_t *codes; /* borrowed: one byte per base, 0..4 */
    const int32_t *off;   /* borrowed: n_genes+1 */
    uint8_t *rc;          /* reverse complements, same offsets */
    kpo_posting *post;    /* sorted by (key, gene, pos) */
    int64_t n_post, cap_post;
    uint32_t cur_gene;
} kpo_db;

static int cmp_posting(const void *a, const void *b) {
    const kpo_posting *x = a, *y = b;
    if (x->key != y->key) return x->key < y->key ? -1 : 1;
    if (x->gene != y->gene) return x->gene < y->gene ? -1 : 1;
    return x->pos < y->pos ? -1 : (x->pos > y->pos);
}

static void db_seed(void *ud, int32_t start, int z, uint32_t x) {
    kpo_db *db = ud;
    if (db->n_post == db->cap_post) {
        db->cap_post = db->cap_post ? 2 * db->cap_post : 4096;
        db->post = realloc(db->post, (size_t)db->cap_post * sizeof(kpo_posting));
    }
    kpo_posting *q = &db->post[db->n_post++];
    q->key = x; q->gene = db->cur_gene; q->pos = (uint32_t)start; q->z = (uint32_t)z;
}

KPO_API kpo_db *kpo_db_create(const uint8_t *codes, const int32_t *off, int n_genes) {
    kpo_db *db = calloc(1, sizeof *db);
    db->n_genes = n_genes; db->codes = codes; db->off = off;
    int64_t total = off[n_genes];
    db->rc = malloc((size_t)(total > 0 ? total : 1));
    for (int g = 0; g < n_genes; g++) {
        int len = off[g + 1] - off[g];
        for (int i = 0; i < len; i++) {
            uint8_t c = codes[off[g] + len - 1 - i];
            db->rc[off[g] + i] = c > 3 ? 4 : (uint8_t)(3 - c);
        }
        db->cur_gene = (uint32_t)g;
        kpo_sketch(codes + off[g], len, db_seed, db); /* the gene's forward strand, as minimap2 sketches a query */
    }
    if (!db->post) db->post = malloc(sizeof(kpo_posting));
    qsort(db->post, (size_t)db->n_post, sizeof(kpo_posting), cmp_posting);
    return db;
}

KPO_API void kpo_db_free(kpo_db *db) { if (db) { free(db->rc); free(db->post); free(db); } }
KPO_API int64_t kpo_db_n_postings(const kpo_db *db) { return db->n_post; }

static int64_t lower_bound_key(const kpo_db *db, uint32_t key) {
    int64_t lo = 0, hi = db->n_post;
    while (lo < hi) { int64_t mid = (lo + hi) >> 1; if (db->post[mid].key < key) lo = mid + 1; else hi = mid; }
    return lo;
}

typedef struct {
    const uint32_t *words; int64_t padded_len;
    const int32_t *ctg_start, *ctg_len; int n_ctg;
    const int32_t *n_runs; int n_nruns;
    uint8_t *codes; /* unpacked, N applied (code 4), pad = 0 */
} kpo_asm;

static void asm_unpack(kpo_asm *a) {
    a->codes = malloc((size_t)(a->padded_len > 0 ? a->padded_len : 1));
    for (int64_t i = 0; i < a->padded_len; i++) a->codes[i] = (uint8_t)((a->words[i >> 4] >> (2 * (i & 15))) & 3u);
    for (int r = 0; r < a->n_nruns; r++)
        for (int64_t i = a->n_runs[2 * r]; i < a->n_runs[2 * r + 1]; i++) a->codes[i] = 4;
}

static int cmp_u64(const void *a, const void *b) {
    uint64_t x = *(const uint64_t *)a, y = *(const uint64_t *)b;
    return x < y ? -1 : (x > y);
}

typedef struct {
    const kpo_db *db;
    int64_t ctg_start; /* of the contig being sketched, in the assembly's padded space */
    uint64_t *keys;
    int64_t n, cap;
    int64_t n_seeds;
} kpo_collect;

static void asm_seed(void *ud, int32_t start, int z, uint32_t x) {
    kpo_collect *k = ud;
    const kpo_db *db = k->db;
    k->n_seeds++;
    const int64_t t = k->ctg_start + start;
    for (int64_t i = lower_bound_key(db, x); i < db->n_post && db->post[i].key == x; i++) {
        if (k->n == k->cap) { k->cap *= 2; k->keys = realloc(k->keys, (size_t)k->cap * sizeof(uint64_t)); }
        const kpo_posting *q = &db->post[i];
        const int rev = (int)q->z != z; /* opposite strand bits: the contig carries the gene's reverse complement */
        const int glen = db->off[q->gene + 1] - db->off[q->gene];
        const uint32_t qpos = rev ? (uint32_t)(glen - KP_K - (int)q->pos) : q->pos;
        k->keys[k->n++] = KP_ANCHOR_KEY(2u * q->gene + (uint32_t)rev, (uint64_t)(t - qpos + KP_DIAG_BIAS), qpos);
    }
}

/* minimap2's mid_occ of an index over this assembly (mm_idx_cal_max_occ with -f 2e-4, then the floor of min_mid_occ = 10):
 * the occurrence counts of its DISTINCT minimizers (both strands: the seed value is that of the canonical 15-mer), sorted; the
 * count at position (int)((1 - 2e-4f) n), plus one. */
typedef struct { uint32_t *x; int64_t n, cap; } kpo_xs;
static void xs_seed(void *ud, int32_t start, int z, uint32_t x) {
    kpo_xs *v = ud; (void)start; (void)z;
    if (v->n == v->cap) { v->cap *= 2; v->x = realloc(v->x, (size_t)v->cap * sizeof(uint32_t)); }
    v->x[v->n++] = x;
}
static int cmp_u32v(const void *a, const void *b) { const uint32_t x = *(const uint32_t *)a, y = *(const uint32_t *)b; return x < y ? -1 : (x > y); }
static int asm_mid_occ(const kpo_asm *a) {
    kpo_xs v = {malloc(sizeof(uint32_t) << 16), 0, 1 << 16};
    for (int c = 0; c < a->n_ctg; c++) kpo_sketch(a->codes + a->ctg_start[c], a->ctg_len[c], xs_seed, &v);
    qsort(v.x, (size_t)v.n, sizeof(uint32_t), cmp_u32v);
    int64_t n = 0;
    uint32_t *cnt = malloc(sizeof(uint32_t) * (size_t)(v.n + 1));
    for (int64_t i = 0, j; i < v.n; i = j) {
        for (j = i + 1; j < v.n && v.x[j] == v.x[i]; j++) {}
        cnt[n++] = (uint32_t)(j - i);
    }
    int mid = KP_MID_OCC;
    if (n > 0) {
        qsort(cnt, (size_t)n, sizeof(uint32_t), cmp_u32v);
        int64_t kth = (int64_t)((1. - (double)KP_MID_OCC_FRAC) * (double)n);
        if (kth >= n) kth = n - 1;
        const uint32_t q = cnt[kth] < KP_MID_OCC_HIST - 1 ? cnt[kth] : KP_MID_OCC_HIST - 1; /* (counts are capped where the device's histogram ends) */
        if ((int)q + 1 > mid) mid = (int)q + 1;
    }
    free(cnt); free(v.x);
    return mid;
}

/* all anchors of one assembly, sorted by key; returns count (caller frees *out) */
static int64_t collect_anchors(const kpo_db *db, const kpo_asm *a, uint64_t **out) {
    kpo_collect k = {db, 0, NULL, 0, 1 << 16, 0};
    k.keys = malloc((size_t)k.cap * sizeof(uint64_t));
    for (int c = 0; c < a->n_ctg; c++) {
        k.ctg_start = a->ctg_start[c];
        kpo_sketch(a->codes + a->ctg_start[c], a->ctg_len[c], asm_seed, &k);
    }
    qsort(k.keys, (size_t)k.n, sizeof(uint64_t), cmp_u64);
    /* occurrence cut (kp_spec.h, KP_MID_OCC): the anchors of one gene seed -- (gene, position on the gene's forward strand) --
     * are the occurrences of its value among the assembly's minimizers, whichever strand they lie on; a seed with more than
     * KP_MID_OCC of them is dropped with all its anchors.  Anchors are sorted by gene/strand first, so a gene's anchors are one
     * stretch of the list. */
    int64_t kept = 0;
    /* the assembly's cut (kp_spec.h, OCCURRENCE CUT): minimap2's mid_occ when some gene has KP_MIN_ANCHORS seeds beyond the floor,
     * the floor otherwise */
    int mid_occ = KP_MID_OCC;
    for (int64_t i = 0; i < k.n;) {
        const uint32_t gene = KP_KEY_GS(k.keys[i]) >> 1;
        int64_t j = i;
        while (j < k.n && (KP_KEY_GS(k.keys[j]) >> 1) == gene) j++;
        if (j - i > KP_MID_OCC) {
            const int glen = db->off[gene + 1] - db->off[gene];
            int32_t *occ = calloc((size_t)glen + 1, sizeof(int32_t));
            int n_over = 0;
            for (int64_t u = i; u < j; u++) {
                const int q = (int)KP_KEY_QPOS(k.keys[u]);
                if (++occ[(KP_KEY_GS(k.keys[u]) & 1) ? glen - KP_K - q : q] == KP_MID_OCC + 1) n_over++;
            }
            free(occ);
            if (n_over >= KP_MIN_ANCHORS) { mid_occ = asm_mid_occ(a); break; }
        }
        i = j;
    }
    for (int64_t i = 0; i < k.n;) {
        const uint32_t gene = KP_KEY_GS(k.keys[i]) >> 1;
        int64_t j = i;
        while (j < k.n && (KP_KEY_GS(k.keys[j]) >> 1) == gene) j++;
        const int glen = db->off[gene + 1] - db->off[gene];
        if (j - i > KP_MID_OCC) {
            int32_t *occ = calloc((size_t)glen + 1, sizeof(int32_t));
            for (int64_t u = i; u < j; u++) {
                const int q = (int)KP_KEY_QPOS(k.keys[u]);
                occ[(KP_KEY_GS(k.keys[u]) & 1) ? glen - KP_K - q : q]++;
            }
            const int cut = mid_occ;
            for (int64_t u = i; u < j; u++) {
                const int q = (int)KP_KEY_QPOS(k.keys[u]);
                if (occ[(KP_KEY_GS(k.keys[u]) & 1) ? glen - KP_K - q : q] <= cut) k.keys[kept++] = k.keys[u];
            }
            free(occ);
        } else {
            for (int64_t u = i; u < j; u++) k.keys[kept++] = k.keys[u];
        }
        i = j;
    }
    k.n = kept;
    *out = k.keys;
    return k.n;
}

/* test hook: the seeds of one sequence of codes, in emission order; returns the count (writes at most cap) */
typedef struct { int32_t *start; uint8_t *z; uint32_t *x; int64_t n, cap; } kpo_seedbuf;
static void buf_seed(void *ud, int32_t start, int z, uint32_t x) {
    kpo_seedbuf *b = ud;
    if (b->n < b->cap) { b->start[b->n] = start; b->z[b->n] = (uint8_t)z; b->x[b->n] = x; }
    b->n++;
}
KPO_API int64_t kpo_seeds(const uint8_t *codes, int64_t len, int32_t *start, uint8_t *z, uint32_t *x, int64_t cap) {
    kpo_seedbuf b = {start, z, x, 0, cap};
    kpo_sketch(codes, len, buf_seed, &b);
    return b.n;
}

typedef struct kpo_task {
    int32_t gs;      /* gene*2 + strand */
    int32_t contig;
    int32_t lo;      /* lowest diagonal of the band, true value (tpos - qpos, assembly coordinates) */
    int32_t width;   /* 16, 32, 64 or 128 */
    int32_t n_anchors; /* anchors of the chain (kp_spec.h) */
    int32_t qmin, qmax;
    int32_t chain_score;
} kpo_task;

/* minimap2's chaining of a small anchor set (kp_spec.h): keys[0..n) are one cluster's anchors, n <= KP_CHAIN_DP_MAX.
 * Returns the chain's score and, through *cnt, its anchor count. */
static int chain_small(const uint64_t *keys, int n, int *cnt) {
    static const uint8_t pen[KP_CHAIN_PEN_SIZE] = KP_CHAIN_PEN_TABLE;
    int32_t t[KP_CHAIN_DP_MAX], q[KP_CHAIN_DP_MAX], f[KP_CHAIN_DP_MAX], p[KP_CHAIN_DP_MAX];
    for (int i = 0; i < n; i++) { /* insertion sort by (target, query) */
        int32_t qi = (int32_t)KP_KEY_QPOS(keys[i]), ti = (int32_t)KP_KEY_DIAG(keys[i]) - KP_DIAG_BIAS + qi;
        int j = i;
        while (j > 0 && (t[j - 1] > ti || (t[j - 1] == ti && q[j - 1] > qi))) { t[j] = t[j - 1]; q[j] = q[j - 1]; j--; }
        t[j] = ti; q[j] = qi;
    }
    int best = 0;
    for (int i = 0; i < n; i++) {
        int max_f = KP_K, max_j = -1;
        for (int j = i - 1; j >= 0; j--) {
            int dq = q[i] - q[j], dr = t[i] - t[j];
            if (dq <= 0 || dq > KP_CHAIN_MAX_DIST || dr == 0) continue;
            int dd = dr > dq ? dr - dq : dq - dr, dg = dr < dq ? dr : dq;
            int sc = dg < KP_K ? dg : KP_K;
            if (dd || dg > KP_K) sc -= pen[dd < KP_CHAIN_PEN_SIZE ? dd : KP_CHAIN_PEN_SIZE - 1];
            sc += f[j];
            if (sc > max_f) { max_f = sc; max_j = j; }
        }
        f[i] = max_f; p[i] = max_j;
        if (f[i] >= f[best]) best = i; /* the largest f, the later anchor on ties */
    }
    /* walk back; the chain is cut where the score counted from its end peaks */
    int i = best, max_s = 0, steps = 0, cut_steps = 0;
    do {
        i = p[i]; steps++;
        int sc = i < 0 ? f[best] : f[best] - f[i];
        if (sc > max_s) { max_s = sc; cut_steps = steps; }
    } while (i >= 0);
    *cnt = cut_steps;
    return max_s;
}

static int contig_of(const kpo_asm *a, int64_t t) { /* largest c with ctg_start[c] <= t */
    int lo = 0, hi = a->n_ctg - 1;
    while (lo < hi) { int mid = (lo + hi + 1) >> 1; if (a->ctg_start[mid] <= t) lo = mid; else hi = mid - 1; }
    return lo;
}

/* a cluster (kp_spec.h, v5): what the join stage needs of it -- every cluster is listed, weak ones (fewer than KP_MIN_ANCHORS
 * anchors, or fewer than KP_MIN_SEED_SPAN query bases) included */
typedef struct kpo_pcl {
    int32_t gs, contig;
    int32_t d0, dmax;       /* lowest / highest anchor diagonal (true values) */
    int32_t task;           /* index into the task list, -1 when it has no band task (weak, or rejected by its chain score) */
    int32_t provisional;    /* >= KP_MIN_ANCHORS anchors covering >= KP_MIN_SEED_SPAN query bases */
    int64_t first; int32_t cnt; /* its anchors in the sorted list */
} kpo_pcl;

static int64_t make_tasks_x(const kpo_asm *a, const uint64_t *keys, int64_t n, kpo_task **out, kpo_pcl **pcl_out, int64_t *n_pcl_out) {
    int64_t cap = 1024, nt = 0, pcap = 1024, np = 0;
    kpo_task *tasks = malloc((size_t)cap * sizeof(kpo_task));
    kpo_pcl *pcl = malloc((size_t)pcap * sizeof(kpo_pcl));
    int64_t i = 0;
    while (i < n) {
        uint32_t gs = KP_KEY_GS(keys[i]), d0 = KP_KEY_DIAG(keys[i]), q = KP_KEY_QPOS(keys[i]);
        int ctg = contig_of(a, (int64_t)d0 - KP_DIAG_BIAS + q);
        uint32_t dprev = d0, dmax = d0, qmin = q, qmax = q;
        int cnt = 1;
        int64_t j = i + 1;
        for (; j < n; j++) {
            uint32_t g2 = KP_KEY_GS(keys[j]), d2 = KP_KEY_DIAG(keys[j]), q2 = KP_KEY_QPOS(keys[j]);
            if (g2 != gs || d2 - dprev > KP_DIAG_GAP || d2 - d0 > KP_MAX_SPREAD) break;
            if (contig_of(a, (int64_t)d2 - KP_DIAG_BIAS + q2) != ctg) break;
            dprev = d2; dmax = d2; cnt++;
            if (q2 < qmin) qmin = q2;
            if (q2 > qmax) qmax = q2;
        }
        const int provisional = cnt >= KP_MIN_ANCHORS && (int)(qmax - qmin) + KP_K >= KP_MIN_SEED_SPAN;
        int chain_cnt = cnt, chain_sc = 0, ok;
        if (cnt <= KP_CHAIN_DP_MAX) {
            chain_sc = chain_small(keys + i, cnt, &chain_cnt);
            ok = chain_sc >= KP_MIN_CHAIN_SCORE;
        } else {
            const int span = (int)(qmax - qmin) + KP_K;
            ok = span >= KP_MIN_SEED_SPAN;
            chain_sc = KP_K * cnt < span ? KP_K * cnt : span;
        }
        {
            if (np == pcap) { pcap *= 2; pcl = realloc(pcl, (size_t)pcap * sizeof(kpo_pcl)); }
            kpo_pcl *c = &pcl[np++];
            c->gs = (int32_t)gs; c->contig = ctg;
            c->d0 = (int32_t)((int64_t)d0 - KP_DIAG_BIAS); c->dmax = (int32_t)((int64_t)dmax - KP_DIAG_BIAS);
            c->provisional = provisional;
            ok = ok && provisional;
            c->task = ok ? (int32_t)nt : -1;
            c->first = i; c->cnt = cnt;
        }
        if (ok) {
            int margin = KP_BAND_MARGIN_NARROW, need = (int)(dmax - d0) + 1 + 2 * KP_BAND_MARGIN_NARROW, w = 16;
            if (need > 16) {
                margin = KP_BAND_MARGIN;
                need = (int)(dmax - d0) + 1 + 2 * KP_BAND_MARGIN;
                w = need <= 32 ? 32 : (need <= 64 ? 64 : 128);
            }
            if (nt == cap) { cap *= 2; tasks = realloc(tasks, (size_t)cap * sizeof(kpo_task)); }
            kpo_task *t = &tasks[nt++];
            t->gs = (int32_t)gs; t->contig = ctg; t->width = w; t->n_anchors = chain_cnt; t->chain_score = chain_sc;
            t->lo = (int32_t)((int64_t)d0 - KP_DIAG_BIAS - margin - (w - need) / 2);
            t->qmin = (int32_t)qmin; t->qmax = (int32_t)qmax;
        }
        i = j;
    }
    *out = tasks;
    if (pcl_out) { *pcl_out = pcl; *n_pcl_out = np; } else free(pcl);
    return nt;
}

static int64_t make_tasks(const kpo_asm *a, const uint64_t *keys, int64_t n, kpo_task **out) {
    return make_tasks_x(a, keys, n, out, NULL, NULL);
}

/* Banded local alignment of one task with stored traceback.  out: score, q_start, q_end, t_start, t_end (query in the
 * orientation given, target in assembly coordinates), matches, block_len. */
static void sw_task(const uint8_t *q, int qlen, const uint8_t *tc, int64_t cstart, int64_t cend, int lo, int w,
                    int32_t *out7 /* [8] */) {
    const size_t cells = (size_t)qlen * (size_t)w;
    int32_t *H = malloc(cells * 4), *E = malloc(cells * 4), *F = malloc(cells * 4);
    uint8_t *tH = malloc(cells), *tE = malloc(cells), *tF = malloc(cells);
    const int oe = KP_GAP_OPEN + KP_GAP_EXT, ex = KP_GAP_EXT;
#define AT(r, b) ((size_t)(r) * (size_t)w + (size_t)(b))
#define VALID(r, b) ((r) >= 0 && (b) >= 0 && (b) < w && (int64_t)(r) + lo + (b) >= cstart && (int64_t)(r) + lo + (b) < cend)
    int best_s = 0, best_r = -1, best_b = -1;
    for (int r = 0; r < qlen; r++) {
        for (int b = 0; b < w; b++) {
            int64_t t = (int64_t)r + lo + b;
            if (t < cstart || t >= cend) { H[AT(r, b)] = 0; E[AT(r, b)] = F[AT(r, b)] = KP_NEG_INF; tH[AT(r, b)] = 3; continue; }
            int hl = 0, el = KP_NEG_INF, hu = 0, fu = KP_NEG_INF, hd = 0;
            if (VALID(r, b - 1)) { hl = H[AT(r, b - 1)]; el = E[AT(r, b - 1)]; }
            if (VALID(r - 1, b + 1)) { hu = H[AT(r - 1, b + 1)]; fu = F[AT(r - 1, b + 1)]; }
            if (VALID(r - 1, b)) hd = H[AT(r - 1, b)];
            int e_open = hl - oe, e_ext = el - ex, f_open = hu - oe, f_ext = fu - ex;
            int e = e_open >= e_ext ? e_open : e_ext, f = f_open >= f_ext ? f_open : f_ext;
            tE[AT(r, b)] = e_open >= e_ext ? 0 : 1;
            tF[AT(r, b)] = f_open >= f_ext ? 0 : 1;
            E[AT(r, b)] = e; F[AT(r, b)] = f;
            uint8_t qc = q[r], cc = tc[t];
            int s = (qc > 3 || cc > 3) ? KP_SC_N : (qc == cc ? KP_SC_MATCH : KP_SC_MISMATCH);
            int best = hd + s, tb = 0;
            if (e > best) { best = e; tb = 1; }
            if (f > best) { best = f; tb = 2; }
            if (best <= 0) { H[AT(r, b)] = 0; tH[AT(r, b)] = 3; }
            else {
                H[AT(r, b)] = best; tH[AT(r, b)] = (uint8_t)tb;
                if (best > best_s) { best_s = best; best_r = r; best_b = b; }
            }
        }
    }
    int r = best_r, b = best_b, state = 0, matches = 0, cols = 0;
    int sr = best_r, sb = best_b; /* first aligned cell seen so far */
    int gap = 0, credit = 0;      /* columns of the gap being walked; two-piece credit of the gaps closed so far */
    while (best_s > 0) {
        if (state == 0) {
            if (!VALID(r, b) || tH[AT(r, b)] == 3) break;
            int tb = tH[AT(r, b)];
            if (tb == 0) {
                sr = r; sb = b; cols++;
                if (q[r] <= 3 && q[r] == tc[(int64_t)r + lo + b]) matches++;
                r--; /* diagonal: same band index, previous row */
            } else state = tb;
        } else if (state == 1) { /* E: came from the left (same row, target - 1) */
            int tb = tE[AT(r, b)]; cols++; gap++; b--;
            if (tb == 0) { state = 0; if (gap > KP_GAP_LONG) credit += gap - KP_GAP_LONG; gap = 0; }
        } else { /* F: came from above (row - 1, same target => band index + 1) */
            int tb = tF[AT(r, b)]; cols++; gap++; r--; b++;
            if (tb == 0) { state = 0; if (gap > KP_GAP_LONG) credit += gap - KP_GAP_LONG; gap = 0; }
        }
    }
#undef AT
#undef VALID
    out7[0] = best_s; /* cell score of the one-piece recurrence: what KP_MIN_DP_SCORE is compared with */
    out7[7] = best_s + credit; /* reported score: the path under the two-piece gap cost (kp_spec.h) */
    if (best_s > 0) {
        out7[1] = sr; out7[2] = best_r + 1;
        out7[3] = (int32_t)((int64_t)sr + lo + sb); out7[4] = (int32_t)((int64_t)best_r + lo + best_b + 1);
    } else out7[1] = out7[2] = out7[3] = out7[4] = 0;
    out7[5] = matches; out7[6] = cols;
    free(H); free(E); free(F); free(tH); free(tE); free(tF);
}

/* ---- kp-align v5: chains of anchors across diagonal jumps, their pieces and the joined alignment (kp_spec.h) ---------------- */
typedef struct kpo_join {
    int32_t gs, contig, n_pieces, n_anchors, chain_score, width;
    int32_t lo[KP_JOIN_MAX_PIECES];    /* lowest diagonal of every piece's band, query order */
    int32_t r0[KP_JOIN_MAX_PIECES], r1[KP_JOIN_MAX_PIECES]; /* rows [r0, r1) of the piece (r1 is clipped to the gene's length by the fill) */
    int32_t cmask[KP_JOIN_MAX_PIECES]; /* bit c: the piece holds an anchor of the group's cluster c */
    int32_t n_members, member_task[KP_JOIN_GROUP_MAX]; /* the group's clusters: their band tasks (-1 = none) */
    int32_t weak_mask; /* bit k: piece k belongs to a WEAK END of the chain (kp_spec.h) */
    /* results: end_s[k] = the piece's best cell score (piece 0 included); per piece k > 0: res[k] = cell score, q_start, q_end,
     * t_start, t_end, matches, block_len, reported score, bonus of the order score; state[k]: 0 = no END / below the cut-off /
     * its path crosses no gap / on a path reported before, 1 = hit, 2 = rejected by the drop test; visited[k] = mask of the
     * pieces the path of piece k runs through; drop_mask = the group's clusters whose band tasks report no hit of their own */
    int32_t end_s[KP_JOIN_MAX_PIECES];
    int32_t res[KP_JOIN_MAX_PIECES][9];
    int32_t state[KP_JOIN_MAX_PIECES], visited[KP_JOIN_MAX_PIECES];
    int32_t drop_mask;
} kpo_join;

typedef struct { int32_t t, q, c; } kpo_ganchor;
static int cmp_ganchor(const void *a, const void *b) {
    const kpo_ganchor *x = a, *y = b;
    if (x->t != y->t) return x->t < y->t ? -1 : 1;
    return x->q < y->q ? -1 : (x->q > y->q);
}

/* One GROUP: minimap2's chaining DP over all its anchors (exact: every earlier anchor within KP_CHAIN_MAX_DIST is a candidate),
 * mg_chain_backtrack, every chain cut into PIECES; chains of 2..KP_JOIN_MAX_PIECES pieces become join records. */
static void chain_group(const uint64_t *keys, const kpo_pcl *pcl, const int *member, int n_members, kpo_join **joins, int64_t *nj, int64_t *cap) {
    static const uint8_t pen[KP_CHAIN_PEN_SIZE] = KP_CHAIN_PEN_TABLE;
    int64_t na = 0;
    for (int c = 0; c < n_members; c++) na += pcl[member[c]].cnt;
    if (na > KP_JOIN_ANCHOR_MAX || na < KP_MIN_ANCHORS) return;
    kpo_ganchor *a = malloc((size_t)na * sizeof(kpo_ganchor));
    int32_t *f = malloc((size_t)na * 4), *p = malloc((size_t)na * 4), *used = calloc((size_t)na, 4), *order = malloc((size_t)na * 4);
    int64_t n = 0;
    for (int c = 0; c < n_members; c++) {
        const kpo_pcl *cl = &pcl[member[c]];
        for (int64_t i = cl->first; i < cl->first + cl->cnt; i++) {
            const int32_t q = (int32_t)KP_KEY_QPOS(keys[i]);
            a[n].q = q; a[n].t = (int32_t)KP_KEY_DIAG(keys[i]) - KP_DIAG_BIAS + q; a[n].c = c; n++;
        }
    }
    qsort(a, (size_t)n, sizeof(kpo_ganchor), cmp_ganchor); /* (target, query): no two anchors of a gene/strand share both */
    for (int64_t i = 0; i < n; i++) {
        int max_f = KP_K; int64_t max_j = -1;
        for (int64_t j = i - 1; j >= 0; j--) {
            const int dr = a[i].t - a[j].t, dq = a[i].q - a[j].q;
            if (dr > KP_CHAIN_MAX_DIST) break;
            if (dq <= 0 || dq > KP_CHAIN_MAX_DIST || dr == 0) continue;
            const int dd = dr > dq ? dr - dq : dq - dr, dg = dr < dq ? dr : dq;
            if (dd > KP_JOIN_BW) continue;
            int sc = dg < KP_K ? dg : KP_K;
            if (dd || dg > KP_K) sc -= pen[dd];
            sc += f[j];
            if (sc > max_f) { max_f = sc; max_j = j; }
        }
        f[i] = max_f; p[i] = (int32_t)max_j;
    }
    /* ends by (f descending, the later anchor first) */
    int64_t n_ends = 0;
    for (int64_t i = 0; i < n; i++) if (f[i] >= KP_MIN_CHAIN_SCORE) order[n_ends++] = (int32_t)i;
    for (int64_t x = 1; x < n_ends; x++) { /* insertion sort (groups are small) */
        const int32_t v = order[x]; int64_t y = x;
        while (y > 0 && (f[order[y - 1]] < f[v] || (f[order[y - 1]] == f[v] && order[y - 1] < v))) { order[y] = order[y - 1]; y--; }
        order[y] = v;
    }
    int32_t *chain = malloc((size_t)n * 4);
    for (int64_t e = 0; e < n_ends; e++) {
        const int32_t end = order[e];
        if (used[end]) continue;
        /* walk back until a used anchor or the start; the chain is cut where the score counted from its end peaks */
        int32_t i = end, max_s = 0, steps = 0, cut = 0;
        do {
            i = p[i]; steps++;
            const int sc = i < 0 ? f[end] : f[end] - f[i];
            if (sc > max_s) { max_s = sc; cut = steps; }
            else if (max_s - sc > KP_JOIN_BW) break; /* (mg_chain_backtrack's max_drop = bw) */
        } while (i >= 0 && !used[i]);
        int32_t len = 0;
        for (i = end; len < cut; i = p[i]) { chain[len++] = i; used[i] = 1; }
        if (max_s < KP_MIN_CHAIN_SCORE || len < KP_MIN_ANCHORS) continue; /* (its anchors stay used, as in mg_chain_backtrack) */
        /* pieces, in query order (the chain was walked backwards): a new piece where the diagonal jumps by more than
         * KP_DIAG_GAP or the piece would span more than KP_MAX_SPREAD diagonals */
        int np = 0, dmin[KP_JOIN_MAX_PIECES + 1], dmax[KP_JOIN_MAX_PIECES + 1], cm[KP_JOIN_MAX_PIECES + 1];
        int qlo[KP_JOIN_MAX_PIECES + 1], qhi[KP_JOIN_MAX_PIECES + 1], jump_before[KP_JOIN_MAX_PIECES + 1];
        int prev_d = 0, over = 0;
        for (int32_t z = len - 1; z >= 0; z--) {
            const kpo_ganchor *x = &a[chain[z]];
            const int d = x->t - x->q;
            int fresh = np == 0;
            if (!fresh) {
                const int jump = d > prev_d ? d - prev_d : prev_d - d;
                const int lo2 = d < dmin[np - 1] ? d : dmin[np - 1], hi2 = d > dmax[np - 1] ? d : dmax[np - 1];
                fresh = jump > KP_DIAG_GAP || hi2 - lo2 > KP_MAX_SPREAD;
            }
            if (fresh) {
                if (np == KP_JOIN_MAX_PIECES) { over = 1; break; }
                dmin[np] = dmax[np] = d; cm[np] = 0; qlo[np] = x->q; jump_before[np] = np ? (d > prev_d ? d - prev_d : prev_d - d) : 0; np++;
            }
            if (d < dmin[np - 1]) dmin[np - 1] = d;
            if (d > dmax[np - 1]) dmax[np - 1] = d;
            cm[np - 1] |= 1 << x->c;
            qhi[np - 1] = x->q;
            prev_d = d;
        }
        if (over || np < 2) continue;
        if (*nj == *cap) { *cap *= 2; *joins = realloc(*joins, (size_t)*cap * sizeof(kpo_join)); }
        kpo_join *J = &(*joins)[(*nj)++];
        memset(J, 0, sizeof *J);
        J->gs = pcl[member[0]].gs; J->contig = pcl[member[0]].contig; J->n_pieces = np;
        J->n_anchors = len; J->chain_score = max_s;
        J->n_members = n_members;
        for (int c = 0; c < n_members; c++) J->member_task[c] = pcl[member[c]].task;
        int spread = 0; /* the widest piece's diagonal range */
        for (int k = 0; k < np; k++) if (dmax[k] - dmin[k] > spread) spread = dmax[k] - dmin[k];
        const int margin = kp_piece_margin(spread);
        J->width = kp_piece_width(spread);
        for (int k = 0; k < np; k++) {
            const int need = dmax[k] - dmin[k] + 1 + 2 * margin;
            J->lo[k] = dmin[k] - margin - (J->width - need) / 2;
            J->cmask[k] = cm[k];
        }
        J->weak_mask = kp_weak_ends(np, qlo, qhi, jump_before);
        for (int k = 0; k < np; k++) {
            J->r0[k] = k ? qhi[k - 1] & ~7 : 0;
            J->r1[k] = k + 1 < np ? qlo[k + 1] + KP_K : KP_MAX_GENE_LEN + 1;
        }
    }
    free(chain); free(a); free(f); free(p); free(used); free(order);
}

/* groups of clusters -> chains of anchors -> join records */
static int64_t make_joins(const uint64_t *keys, const kpo_pcl *pcl, int64_t np, kpo_join **out) {
    int64_t cap = 64, nj = 0;
    kpo_join *joins = malloc((size_t)cap * sizeof(kpo_join));
    /* GROUPS (kp_spec.h): per gene/strand and contig, the clusters in list order; a cluster continues the open
     * sequence of its contig iff its lowest diagonal is within KP_JOIN_BW of that sequence's last cluster's highest one and
     * the sequence holds fewer than KP_JOIN_GROUP_MAX clusters.  At most KP_JOIN_OPEN sequences of a gene/strand are open
     * at a time: a new contig's cluster closes the open sequence whose last cluster ends lowest (lowest contig on ties).
     * A closed sequence of two or more clusters, one of them provisional, is a group. */
    typedef struct { int n; int member[KP_JOIN_GROUP_MAX]; } kpo_seq;
    kpo_seq open[KP_JOIN_OPEN];
    int n_open = 0;
#define KPO_CLOSE(slot) do { if (open[slot].n >= 2) { int any = 0; for (int z = 0; z < open[slot].n; z++) any |= pcl[open[slot].member[z]].provisional; \
                                 if (any) chain_group(keys, pcl, open[slot].member, open[slot].n, &joins, &nj, &cap); } \
                             open[slot] = open[--n_open]; } while (0)
    for (int64_t c = 0; c <= np; c++) {
        if (c == np || (n_open > 0 && pcl[open[0].member[0]].gs != pcl[c].gs))
            while (n_open > 0) KPO_CLOSE(0);
        if (c == np) break;
        int found = -1;
        for (int s2 = 0; s2 < n_open; s2++)
            if (pcl[open[s2].member[0]].contig == pcl[c].contig) found = s2;
        if (found >= 0) {
            const kpo_pcl *last = &pcl[open[found].member[open[found].n - 1]];
            if (pcl[c].d0 - last->dmax <= KP_JOIN_BW && open[found].n < KP_JOIN_GROUP_MAX) { open[found].member[open[found].n++] = (int)c; continue; }
            KPO_CLOSE(found);
        } else if (n_open == KP_JOIN_OPEN) {
            int ev = 0;
            for (int s2 = 1; s2 < n_open; s2++) {
                const kpo_pcl *x = &pcl[open[s2].member[open[s2].n - 1]], *y = &pcl[open[ev].member[open[ev].n - 1]];
                if (x->dmax < y->dmax || (x->dmax == y->dmax && x->contig < y->contig)) ev = s2;
            }
            KPO_CLOSE(ev);
        }
        open[n_open].n = 1; open[n_open].member[0] = (int)c; n_open++;
    }
#undef KPO_CLOSE
    *out = joins;
    return nj;
}

/* Joined fill and walk-back of one join (kp_spec.h).  q: the gene in the join's orientation. */
enum { XT_DIAG = 0, XT_E = 1, XT_F = 2, XT_RESTART = 3, XT_X1 = 4, XT_X2 = 5 };
#define KPO_DEAD(v) ((v) < KP_NEG_INF / 2)
typedef struct {
    int32_t *H; uint8_t *tH, *tE, *tF; /* [qlen * w] */
    int64_t *x1, *x2;                  /* exports towards the next piece: best key per row (or column), INT64_MIN = none */
    int32_t *a1, *a2;                  /* ... and the coordinate (t' or r') of the cell that holds it */
    int64_t xbase; int xlen, horizontal; /* index = row (horizontal) or t - xbase (vertical) */
    int end_s, end_r, end_b;
} kpo_xpiece;

static void join_run(kpo_join *J, const uint8_t *q, int qlen, const uint8_t *tc, int64_t cstart, int64_t cend) {
    const int m = J->n_pieces, w = J->width;
    const int oe = KP_GAP_OPEN + KP_GAP_EXT, ex = KP_GAP_EXT;
    const size_t cells = (size_t)qlen * (size_t)w;
    kpo_xpiece P[KP_JOIN_MAX_PIECES];
    memset(P, 0, sizeof P);
#define AT(r, b) ((size_t)(r) * (size_t)w + (size_t)(b))
    for (int k = 0; k < m; k++) {
        kpo_xpiece *X = &P[k];
        const int lo = J->lo[k];
        /* every piece is a local alignment (H >= 0, restarts); pieces after the first have the cross gaps from the piece
         * before them as two more sources of H */
        const int none = 0;
        X->H = malloc(cells * 4); X->tH = malloc(cells); X->tE = malloc(cells); X->tF = malloc(cells);
        int32_t *E = malloc(cells * 4), *F = malloc(cells * 4);
        const kpo_xpiece *prev = k > 0 ? &P[k - 1] : NULL;
        if (k + 1 < m) { /* this piece exports towards piece k + 1 */
            X->horizontal = J->lo[k + 1] > lo;
            X->xbase = lo; X->xlen = X->horizontal ? qlen : qlen + w;
            X->x1 = malloc((size_t)X->xlen * 8); X->x2 = malloc((size_t)X->xlen * 8);
            X->a1 = malloc((size_t)X->xlen * 4); X->a2 = malloc((size_t)X->xlen * 4);
            for (int i = 0; i < X->xlen; i++) { X->x1[i] = X->x2[i] = INT64_MIN; X->a1[i] = X->a2[i] = 0; }
        }
        X->end_s = 0; X->end_r = X->end_b = -1;
        const int R0 = J->r0[k], R1 = J->r1[k] < qlen ? J->r1[k] : qlen;
#define VALID(r, b) ((r) >= R0 && (r) < R1 && (b) >= 0 && (b) < w && (int64_t)(r) + lo + (b) >= cstart && (int64_t)(r) + lo + (b) < cend)
        for (int r = 0; r < qlen; r++) {
            for (int b = 0; b < w; b++) {
                const int64_t t = (int64_t)r + lo + b;
                if (t < cstart || t >= cend || r < R0 || r >= R1) { X->H[AT(r, b)] = none; E[AT(r, b)] = F[AT(r, b)] = KP_NEG_INF; X->tH[AT(r, b)] = XT_RESTART; X->tE[AT(r, b)] = X->tF[AT(r, b)] = 0; continue; }
                int hl = none, el = KP_NEG_INF, hu = none, fu = KP_NEG_INF, hd = none;
                if (VALID(r, b - 1)) { hl = X->H[AT(r, b - 1)]; el = E[AT(r, b - 1)]; }
                if (VALID(r - 1, b + 1)) { hu = X->H[AT(r - 1, b + 1)]; fu = F[AT(r - 1, b + 1)]; }
                if (VALID(r - 1, b)) hd = X->H[AT(r - 1, b)];
                const int e_open = hl - oe, e_ext = el - ex, f_open = hu - oe, f_ext = fu - ex;
                int e = e_open >= e_ext ? e_open : e_ext, f = f_open >= f_ext ? f_open : f_ext;
                X->tE[AT(r, b)] = e_open >= e_ext ? 0 : 1;
                X->tF[AT(r, b)] = f_open >= f_ext ? 0 : 1;
                if (KPO_DEAD(e)) e = KP_NEG_INF;
                if (KPO_DEAD(f)) f = KP_NEG_INF;
                E[AT(r, b)] = e; F[AT(r, b)] = f;
                const uint8_t qc = q[r], cc = tc[t];
                const int s = (qc > 3 || cc > 3) ? KP_SC_N : (qc == cc ? KP_SC_MATCH : KP_SC_MISMATCH);
                int64_t best = (int64_t)hd + s; int tb = XT_DIAG;
                if (e > best) { best = e; tb = XT_E; }
                if (f > best) { best = f; tb = XT_F; }
                if (prev && r < (J->r1[k - 1] < qlen ? J->r1[k - 1] : qlen)) { /* cross gaps from piece k - 1: taken in rows of the junction zone */
                    const int64_t xi = prev->horizontal ? r : t - prev->xbase;
                    if (xi >= 0 && xi < prev->xlen && prev->x1[xi] != INT64_MIN) {
                        const int64_t pos = prev->horizontal ? t : r;
                        const int64_t c1 = prev->x1[xi] - KP_GAP_OPEN - (int64_t)KP_GAP_EXT * pos;
                        const int64_t c2 = prev->x2[xi] - KP_GAP_OPEN2 - (int64_t)KP_GAP_EXT2 * pos;
                        if (c1 > best) { best = c1; tb = XT_X1; }
                        if (c2 > best) { best = c2; tb = XT_X2; }
                    }
                }
                const int live = best > 0;
                if (!live) { X->H[AT(r, b)] = 0; X->tH[AT(r, b)] = XT_RESTART; }
                if (live) {
                    X->H[AT(r, b)] = (int)best; X->tH[AT(r, b)] = (uint8_t)tb;
                    if (best > X->end_s) { X->end_s = (int)best; X->end_r = r; X->end_b = b; }
                    if (X->x1 && r >= J->r0[k + 1]) { /* export (cells of this piece, in rows of the junction zone, that lie before / above the next piece's band) */
                        const int lo_next = J->lo[k + 1];
                        if (X->horizontal ? (b < lo_next - lo) : (lo + b > lo_next + w - 1)) {
                            const int64_t xi = X->horizontal ? r : t - X->xbase, pos = X->horizontal ? t : r;
                            const int64_t k1 = best + (int64_t)KP_GAP_EXT * pos, k2 = best + (int64_t)KP_GAP_EXT2 * pos;
                            /* first maximum in increasing t' (rows are visited left to right) / r' (rows in increasing order) */
                            if (k1 > X->x1[xi]) { X->x1[xi] = k1; X->a1[xi] = (int32_t)pos; }
                            if (k2 > X->x2[xi]) { X->x2[xi] = k2; X->a2[xi] = (int32_t)pos; }
                        }
                    }
                }
            }
        }
#undef VALID
        free(E); free(F);
    }
    /* THE JOINED PATH (kp_spec.h): the pieces are tried in the order of their best cells' scores (the earlier piece on ties); a
     * piece whose path the drop test rejects is set aside and the next one tried; the first path that is not rejected settles
     * the chain -- a hit if it crosses at least one gap, otherwise the band task of that piece's cluster stands for it */
    int settled = 0, hit_k = -1, alone_k = -1, any_rejected = 0;
    for (int k = 0; k < m; k++) { J->end_s[k] = P[k].end_s; J->state[k] = 0; J->visited[k] = 0; }
    for (;;) {
        int k = -1;
        for (int z = 0; z < m; z++)
            if (!((settled >> z) & 1) && (k < 0 || P[z].end_s > P[k].end_s)) k = z;
        if (k < 0) break;
        const kpo_xpiece *X = &P[k];
        if (X->end_r < 0 || X->end_s < KP_MIN_DP_SCORE) { alone_k = k; break; } /* (nothing to report: the best piece keeps its own task) */
        int pk = k, r = X->end_r, b = X->end_b, state = 0, matches = 0, cols = 0, gap = 0, credit = 0;
        int sr = r, sb = b, spk = k;
        int suf = 0, sufmax = 0, gsum = 0, rejected = 0, visited = 1 << k, bonus = 0;
        for (;;) {
            const kpo_xpiece *Y = &P[pk];
            const int lo = J->lo[pk];
            if (state == 0) {
                const int64_t t = (int64_t)r + lo + b;
                if (r < J->r0[pk] || b < 0 || b >= w || t < cstart || t >= cend) break;
                const int tb = Y->tH[AT(r, b)];
                if (tb == XT_RESTART) break;
                /* the drop test: the path's score behind this cell, cross-gap costs left out, against its largest value so far --
                 * at every cross gap, and at every cell once a gap has been crossed */
                if (suf + gsum > sufmax) sufmax = suf + gsum;
                else if ((tb >= XT_X1 || visited != 1 << k) && sufmax - (suf + gsum) > KP_JOIN_DROP) { rejected = 1; break; }
                if (tb == XT_DIAG) {
                    sr = r; sb = b; spk = pk; cols++;
                    const uint8_t qc = q[r], cc = tc[t];
                    if (qc <= 3 && qc == cc) matches++;
                    suf += (qc > 3 || cc > 3) ? KP_SC_N : (qc == cc ? KP_SC_MATCH : KP_SC_MISMATCH);
                    r--;
                } else if (tb == XT_E || tb == XT_F) state = tb;
                else { /* a cross gap: on to the cell of piece pk - 1 it came from */
                    const kpo_xpiece *Z = &P[pk - 1];
                    const int64_t xi = Z->horizontal ? r : t - Z->xbase;
                    const int32_t src = tb == XT_X1 ? Z->a1[xi] : Z->a2[xi];
                    const int n = Z->horizontal ? (int)(t - src) : r - src;
                    cols += n;
                    const int cost = tb == XT_X1 ? KP_GAP_OPEN + KP_GAP_EXT * n : KP_GAP_OPEN2 + KP_GAP_EXT2 * n;
                    suf -= cost; gsum += cost;
                    if (cost > KP_GAP_OPEN + kp_log2x2((uint32_t)n)) bonus += cost - (KP_GAP_OPEN + kp_log2x2((uint32_t)n));
                    if (Z->horizontal) b = (int)(src - r - J->lo[pk - 1]);       /* same row, column src */
                    else { b = (int)(t - src - J->lo[pk - 1]); r = src; }      /* same column, row src */
                    pk--; visited |= 1 << pk;
                }
            } else if (state == XT_E) {
                const int tb = Y->tE[AT(r, b)]; cols++; gap++; b--; suf -= ex;
                if (tb == 0) { state = 0; suf -= KP_GAP_OPEN; if (gap > KP_GAP_LONG) credit += gap - KP_GAP_LONG; gap = 0; }
            } else {
                const int tb = Y->tF[AT(r, b)]; cols++; gap++; r--; b++; suf -= ex;
                if (tb == 0) { state = 0; suf -= KP_GAP_OPEN; if (gap > KP_GAP_LONG) credit += gap - KP_GAP_LONG; gap = 0; }
            }
        }
        J->visited[k] = visited;
        if (rejected) { J->state[k] = 2; any_rejected = 1; settled |= 1 << k; continue; }
        if (visited == 1 << k) { alone_k = k; break; } /* crosses no gap: the band task of the piece's cluster covers it */
        J->state[k] = 1; hit_k = k;
        J->res[k][0] = X->end_s; J->res[k][1] = sr; J->res[k][2] = X->end_r + 1;
        J->res[k][3] = (int32_t)((int64_t)sr + J->lo[spk] + sb); J->res[k][4] = (int32_t)((int64_t)X->end_r + J->lo[k] + X->end_b + 1);
        J->res[k][5] = matches; J->res[k][6] = cols; J->res[k][7] = X->end_s + credit;
        J->res[k][8] = bonus < KP_HIT_BONUS_MAX ? bonus : KP_HIT_BONUS_MAX;
        break;
    }
    /* CONSUMED PIECES (kp_spec.h): the clusters of the pieces the joined hit runs through, and those of the chain's weak ends
     * that no reported path reaches, lose the hit of their own band task */
    {
        const int on = hit_k >= 0 ? J->visited[hit_k] : 0, keep = alone_k >= 0 ? 1 << alone_k : 0;
        int drop = 0;
        for (int k = 0; k < m; k++) {
            if ((on >> k) & 1) drop |= J->cmask[k];
            else if (!any_rejected && ((J->weak_mask >> k) & 1) && !((keep >> k) & 1)) drop |= J->cmask[k];
        }
        if (alone_k >= 0) drop &= ~J->cmask[alone_k];
        J->drop_mask = drop;
    }
#undef AT
    for (int k = 0; k < m; k++) {
        free(P[k].H); free(P[k].tH); free(P[k].tE); free(P[k].tF);
        free(P[k].x1); free(P[k].x2); free(P[k].a1); free(P[k].a2);
    }
}

static const float *ln_half_table(void) {
    static float *t = NULL;
    if (!t) {
        t = malloc(sizeof(float) * KP_MAPQ_LN_HALF_SIZE);
        for (int i = 0; i < KP_MAPQ_LN_HALF_SIZE; i++) t[i] = i ? kp_mapq_ln((double)i / 2.0) : 0.0f;
    }
    return t;
}
static const float *ln_int_table(void) {
    static float *t = NULL;
    if (!t) {
        t = malloc(sizeof(float) * KP_MAPQ_LN_INT_SIZE);
        for (int i = 0; i < KP_MAPQ_LN_INT_SIZE; i++) t[i] = i ? kp_mapq_ln((double)i) : 0.0f;
    }
    return t;
}

/* primary / secondary and mapping quality of one gene's hits (emission order), kp_spec.h: minimap2's mm_set_parent
 * (mask level 1/2 with the uncovered-length correction) and mm_set_mapq on the finished hits.  On entry a hit's chain
 * score sits in its (mapq, pad_) bytes (low byte in mapq), clamped to 65535. */
static int hit_chain_score(const kp_hit *h) { return (int)h->mapq | ((int)h->pad_ << 8); }

static void assign_mapq(kp_hit *h, int n) {
    const size_t m = (size_t)(n > 0 ? n : 1);
    int *parent = malloc(sizeof(int) * m), *subsc = calloc(m, sizeof(int)), *dp2 = calloc(m, sizeof(int)), *n_sub = calloc(m, sizeof(int));
    for (int i = 0; i < n; i++) {
        parent[i] = i;
        const int si = h[i].q_start, ei = h[i].q_end;
        /* bases of [si, ei) that no earlier primary hit covers */
        int uncov = 0, any = 0;
        for (int x = si; x < ei;) {
            int reach = x, next = ei;
            for (int j = 0; j < i; j++) {
                if (parent[j] != j || h[j].q_end <= si || h[j].q_start >= ei) continue;
                any = 1;
                const int sj = h[j].q_start > si ? h[j].q_start : si, ej = h[j].q_end < ei ? h[j].q_end : ei;
                if (sj <= x) { if (ej > reach) reach = ej; }
                else if (sj < next) next = sj;
            }
            if (reach > x) x = reach;
            else { uncov += next - x; x = next; }
        }
        if (!any) continue;
        for (int j = 0; j < i; j++) {
            if (parent[j] != j || h[j].q_end <= si || h[j].q_start >= ei) continue;
            const int sj = h[j].q_start, ej = h[j].q_end;
            const int mn = ej - sj < ei - si ? ej - sj : ei - si, mx = ej - sj > ei - si ? ej - sj : ei - si;
            const int ol = (ei < ej ? ei : ej) - (si > sj ? si : sj);
            if ((float)ol / (float)mn - (float)uncov / (float)mx > 0.5f) {
                int cnt_sub = h[i].n_seeds >= h[j].n_seeds;
                const int sci = hit_chain_score(&h[i]);
                parent[i] = j;
                if (sci > subsc[j]) subsc[j] = sci;
                if (h[j].contig != h[i].contig || h[j].t_start != h[i].t_start || h[j].t_end != h[i].t_end || ol != mn) {
                    if (KP_HIT_OSCORE(h[i].score) > dp2[j]) dp2[j] = KP_HIT_OSCORE(h[i].score);
                    if (KP_HIT_OSCORE(h[j].score) - KP_HIT_OSCORE(h[i].score) <= 2 * KP_SC_MATCH - KP_SC_MISMATCH) cnt_sub = 1;
                }
                if (cnt_sub) n_sub[j]++;
                break;
            }
        }
    }
    for (int i = 0; i < n; i++) {
        const int cs = hit_chain_score(&h[i]);
        h[i].pad_ = 0;
        h[i].mapq = parent[i] != i ? 0
                                    : (uint8_t)kp_mapq_value(KP_HIT_OSCORE(h[i].score), cs, h[i].n_seeds, h[i].matches, h[i].block_len, subsc[i], dp2[i],
                                                             n_sub[i], ln_half_table(), ln_int_table());
    }
    for (int i = 0; i < n; i++) h[i].score = KP_HIT_SCORE(h[i].score); /* the finished record holds the plain score */
    free(parent); free(subsc); free(dp2); free(n_sub);
}

static int cmp_hit(const void *a, const void *b) {
    const kp_hit *x = a, *y = b;
    if (x->gene != y->gene) return x->gene < y->gene ? -1 : 1;
    if (KP_HIT_OSCORE(x->score) != KP_HIT_OSCORE(y->score)) return KP_HIT_OSCORE(x->score) > KP_HIT_OSCORE(y->score) ? -1 : 1;
    if (x->contig != y->contig) return x->contig < y->contig ? -1 : 1;
    if (x->t_start != y->t_start) return x->t_start < y->t_start ? -1 : 1;
    if (x->strand != y->strand) return x->strand > y->strand ? -1 : 1;
    if (x->q_start != y->q_start) return x->q_start < y->q_start ? -1 : 1;
    if (x->q_end != y->q_end) return x->q_end < y->q_end ? -1 : 1;
    if (x->t_end != y->t_end) return x->t_end < y->t_end ? -1 : 1;
    if (x->matches != y->matches) return x->matches > y->matches ? -1 : 1;
    if (x->block_len != y->block_len) return x->block_len < y->block_len ? -1 : 1;
    if (x->n_seeds != y->n_seeds) return x->n_seeds > y->n_seeds ? -1 : 1;
    return hit_chain_score(x) > hit_chain_score(y) ? -1 : (hit_chain_score(x) < hit_chain_score(y)); /* (still in mapq, pad_) */
}

static int same_span(const kp_hit *x, const kp_hit *y) {
    return x->gene == y->gene && x->contig == y->contig && x->strand == y->strand && x->q_start == y->q_start &&
           x->q_end == y->q_end && x->t_start == y->t_start && x->t_end == y->t_end;
}

static void asm_init(kpo_asm *a, const uint32_t *words, int64_t padded_len, const int32_t *ctg_start,
                     const int32_t *ctg_len, int n_ctg, const int32_t *n_runs, int n_nruns) {
    a->words = words; a->padded_len = padded_len; a->ctg_start = ctg_start; a->ctg_len = ctg_len; a->n_ctg = n_ctg;
    a->n_runs = n_runs; a->n_nruns = n_nruns;
    asm_unpack(a);
}

/* Stage entry points (for stage-by-stage parity tests).  Each returns the count; when the buffer is too small the
 * first `cap` items are written and the full count is still returned. */
KPO_API int64_t kpo_anchors(const kpo_db *db, const uint32_t *words, int64_t padded_len, const int32_t *ctg_start,
                            const int32_t *ctg_len, int n_ctg, const int32_t *n_runs, int n_nruns, uint64_t *out,
                            int64_t cap) {
    kpo_asm a; asm_init(&a, words, padded_len, ctg_start, ctg_len, n_ctg, n_runs, n_nruns);
    uint64_t *keys; int64_t n = n_ctg ? collect_anchors(db, &a, &keys) : (keys = NULL, 0);
    if (out) memcpy(out, keys, (size_t)(n < cap ? n : cap) * sizeof(uint64_t));
    free(keys); free(a.codes);
    return n;
}

KPO_API int64_t kpo_tasks(const kpo_db *db, const uint32_t *words, int64_t padded_len, const int32_t *ctg_start,
                          const int32_t *ctg_len, int n_ctg, const int32_t *n_runs, int n_nruns, kpo_task *out,
                          int64_t cap) {
    kpo_asm a; asm_init(&a, words, padded_len, ctg_start, ctg_len, n_ctg, n_runs, n_nruns);
    uint64_t *keys; int64_t n = n_ctg ? collect_anchors(db, &a, &keys) : (keys = NULL, 0);
    kpo_task *tasks; int64_t nt = make_tasks(&a, keys, n, &tasks);
    if (out) memcpy(out, tasks, (size_t)(nt < cap ? nt : cap) * sizeof(kpo_task));
    free(tasks); free(keys); free(a.codes);
    return nt;
}

/* raw DP result per task, before the score filter: [n][7] = score, q_start, q_end (query as aligned: reverse
 * complement coordinates for strand -), t_start, t_end (assembly coordinates), matches, block_len */
KPO_API int64_t kpo_sw(const kpo_db *db, const uint32_t *words, int64_t padded_len, const int32_t *ctg_start,
                       const int32_t *ctg_len, int n_ctg, const int32_t *n_runs, int n_nruns, const kpo_task *tasks,
                       int64_t n_tasks, int32_t *out7) {
    kpo_asm a; asm_init(&a, words, padded_len, ctg_start, ctg_len, n_ctg, n_runs, n_nruns);
    for (int64_t i = 0; i < n_tasks; i++) {
        const kpo_task *t = &tasks[i];
        int g = t->gs >> 1, qlen = db->off[g + 1] - db->off[g];
        const uint8_t *q = ((t->gs & 1) ? db->rc : db->codes) + db->off[g];
        int64_t cs = ctg_start[t->contig];
        int32_t r8[8];
        sw_task(q, qlen, a.codes, cs, cs + ctg_len[t->contig], t->lo, t->width, r8);
        memcpy(out7 + 7 * i, r8, 7 * sizeof(int32_t));
    }
    free(a.codes);
    return n_tasks;
}

/* joins of one assembly, run: caller frees *out */
static int64_t run_joins(const kpo_db *db, const kpo_asm *a, const uint64_t *keys, const kpo_pcl *pcl, int64_t np, kpo_join **out) {
    kpo_join *joins; const int64_t nj = make_joins(keys, pcl, np, &joins);
    for (int64_t j = 0; j < nj; j++) {
        kpo_join *J = &joins[j];
        const int g = J->gs >> 1, qlen = db->off[g + 1] - db->off[g];
        const uint8_t *q = ((J->gs & 1) ? db->rc : db->codes) + db->off[g];
        const int64_t cs = a->ctg_start[J->contig];
        join_run(J, q, qlen, a->codes, cs, cs + a->ctg_len[J->contig]);
    }
    *out = joins;
    return nj;
}

/* stage entry point: the joins of one assembly as rows of KPO_JOIN_ROW int32 --
 * gs, contig, n_pieces, n_anchors, chain_score, width, lo[8], then per piece state, visited, res[9] */
#define KPO_JOIN_ROW (6 + KP_JOIN_MAX_PIECES + 11 * KP_JOIN_MAX_PIECES)
KPO_API int64_t kpo_joins(const kpo_db *db, const uint32_t *words, int64_t padded_len, const int32_t *ctg_start,
                          const int32_t *ctg_len, int n_ctg, const int32_t *n_runs, int n_nruns, int32_t *out, int64_t cap) {
    kpo_asm a; asm_init(&a, words, padded_len, ctg_start, ctg_len, n_ctg, n_runs, n_nruns);
    uint64_t *keys; int64_t n = n_ctg ? collect_anchors(db, &a, &keys) : (keys = NULL, 0);
    kpo_task *tasks; kpo_pcl *pcl; int64_t np = 0;
    make_tasks_x(&a, keys, n, &tasks, &pcl, &np);
    kpo_join *joins; const int64_t nj = run_joins(db, &a, keys, pcl, np, &joins);
    for (int64_t j = 0; j < nj && j < cap; j++) {
        const kpo_join *J = &joins[j];
        int32_t *o = out + KPO_JOIN_ROW * j;
        memset(o, 0, KPO_JOIN_ROW * sizeof(int32_t));
        o[0] = J->gs; o[1] = J->contig; o[2] = J->n_pieces; o[3] = J->n_anchors; o[4] = J->chain_score; o[5] = J->width;
        for (int k = 0; k < J->n_pieces; k++) {
            o[6 + k] = J->lo[k];
            int32_t *pr = o + 6 + KP_JOIN_MAX_PIECES + 11 * k;
            pr[0] = J->state[k]; pr[1] = J->visited[k];
            if (J->state[k] == 1) memcpy(pr + 2, J->res[k], 9 * sizeof(int32_t));
        }
    }
    free(joins); free(pcl); free(tasks); free(keys); free(a.codes);
    return nj;
}

/* Full aligner for one assembly: hits in emission order.  Returns the number of hits (writes at most cap). */
KPO_API int64_t kpo_align(const kpo_db *db, const uint32_t *words, int64_t padded_len, const int32_t *ctg_start,
                          const int32_t *ctg_len, int n_ctg, const int32_t *n_runs, int n_nruns, kp_hit *out,
                          int64_t cap, int64_t *stats /* optional [3]: anchors, tasks, dp cells */,
                          int32_t *chain_out /* optional [cap]: the chain score behind every hit (| order-score bonus << 16) */) {
    kpo_asm a; asm_init(&a, words, padded_len, ctg_start, ctg_len, n_ctg, n_runs, n_nruns);
    uint64_t *keys; int64_t n = n_ctg ? collect_anchors(db, &a, &keys) : (keys = NULL, 0);
    kpo_task *tasks; kpo_pcl *pcl; int64_t np = 0;
    int64_t nt = make_tasks_x(&a, keys, n, &tasks, &pcl, &np);
    kpo_join *joins; const int64_t nj = run_joins(db, &a, keys, pcl, np, &joins);
    uint8_t *dropped = calloc((size_t)(nt > 0 ? nt : 1), 1); /* band tasks whose hit a joined path replaces */
    int64_t n_join_hits = 0;
    for (int64_t j = 0; j < nj; j++) {
        for (int k = 1; k < joins[j].n_pieces; k++)
            if (joins[j].state[k] == 1) n_join_hits++;
        for (int c = 0; c < joins[j].n_members; c++)
            if (((joins[j].drop_mask >> c) & 1) && joins[j].member_task[c] >= 0) dropped[joins[j].member_task[c]] = 1;
    }
    kp_hit *hits = malloc((size_t)(nt + n_join_hits > 0 ? nt + n_join_hits : 1) * sizeof(kp_hit));
    int64_t nh = 0, cells = 0;
    for (int64_t i = 0; i < nt; i++) {
        const kpo_task *t = &tasks[i];
        int g = t->gs >> 1, rev = t->gs & 1, qlen = db->off[g + 1] - db->off[g];
        const uint8_t *q = (rev ? db->rc : db->codes) + db->off[g];
        int64_t cs = ctg_start[t->contig];
        int32_t r[8];
        sw_task(q, qlen, a.codes, cs, cs + ctg_len[t->contig], t->lo, t->width, r);
        cells += (int64_t)qlen * t->width;
        if (r[0] < KP_MIN_DP_SCORE || dropped[i]) continue;
        kp_hit *h = &hits[nh++];
        memset(h, 0, sizeof *h);
        h->gene = g; h->contig = t->contig; h->strand = rev ? -1 : 1;
        h->q_start = rev ? qlen - r[2] : r[1];
        h->q_end = rev ? qlen - r[1] : r[2];
        h->t_start = (int32_t)(r[3] - cs); h->t_end = (int32_t)(r[4] - cs);
        h->score = r[7]; h->matches = r[5]; h->block_len = r[6];
        h->n_seeds = (uint8_t)(t->n_anchors < 255 ? t->n_anchors : 255);
        { const int cs = t->chain_score < 65535 ? t->chain_score : 65535; h->mapq = (uint8_t)(cs & 255); h->pad_ = (uint8_t)(cs >> 8); }
    }
    for (int64_t j = 0; j < nj; j++) { /* the joined paths */
        const kpo_join *J = &joins[j];
        const int g = J->gs >> 1, rev = J->gs & 1, qlen = db->off[g + 1] - db->off[g];
        const int64_t cs = ctg_start[J->contig];
        for (int k = 0; k < J->n_pieces; k++) { const int R1 = J->r1[k] < qlen ? J->r1[k] : qlen; if (R1 > J->r0[k]) cells += (int64_t)(R1 - J->r0[k]) * J->width; }
        for (int k = 1; k < J->n_pieces; k++) {
            if (J->state[k] != 1) continue;
            const int32_t *r = J->res[k];
            kp_hit *h = &hits[nh++];
            memset(h, 0, sizeof *h);
            h->gene = g; h->contig = J->contig; h->strand = rev ? -1 : 1;
            h->q_start = rev ? qlen - r[2] : r[1];
            h->q_end = rev ? qlen - r[1] : r[2];
            h->t_start = (int32_t)(r[3] - cs); h->t_end = (int32_t)(r[4] - cs);
            h->score = (int32_t)((uint32_t)r[7] | ((uint32_t)r[8] << KP_HIT_BONUS_SHIFT)); h->matches = r[5]; h->block_len = r[6];
            h->n_seeds = (uint8_t)(J->n_anchors < 255 ? J->n_anchors : 255);
            { const int c2 = J->chain_score < 65535 ? J->chain_score : 65535; h->mapq = (uint8_t)(c2 & 255); h->pad_ = (uint8_t)(c2 >> 8); }
        }
    }
    free(dropped); free(joins); free(pcl);
    qsort(hits, (size_t)nh, sizeof(kp_hit), cmp_hit);
    int64_t m = 0;
    for (int64_t i = 0; i < nh; i++) { /* hits with the same span are emitted once */
        if (m > 0 && same_span(&hits[m - 1], &hits[i])) continue;
        hits[m++] = hits[i];
    }
    if (chain_out) /* (with the order-score bonus of a joined hit above bit 16: the other input of its mapq that the record does not keep) */
        for (int64_t i = 0; i < m && i < cap; i++) chain_out[i] = hit_chain_score(&hits[i]) | (int32_t)(((uint32_t)hits[i].score >> KP_HIT_BONUS_SHIFT) << 16);
    for (int64_t i = 0; i < m;) { /* mapping qualities, gene by gene */
        int64_t j = i;
        while (j < m && hits[j].gene == hits[i].gene) j++;
        assign_mapq(hits + i, (int)(j - i));
        i = j;
    }
    if (out) memcpy(out, hits, (size_t)(m < cap ? m : cap) * sizeof(kp_hit));
    if (stats) { stats[0] = n; stats[1] = nt; stats[2] = cells; }
    free(hits); free(tasks); free(keys); free(a.codes);
    return m;
}
