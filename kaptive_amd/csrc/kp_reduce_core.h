// kp_reduce_core.h -- per-assembly pieces of the batched reduction, written once as plain functions.
//
// They restate, for one assembly at a time and on flat arrays, the reduction the reference runs in numpy/numba after
// its aligner returns (src/kaptive/serotyping/core.py:157-396); kaptive_amd/serotyping/core.py::Serotyper.reduce is
// the golden-pinned statement of the same steps and is what these functions are tested against.  The HIP kernels in
// kp_reduce.hip call them (block-parallel where the work is quadratic, one lane where it is inherently sequential and
// tiny); tests/native_harness compiles the same header with g++ so the logic is checked in the GPU-less container.
// Nothing here is a product CPU path: the product only ever loads the HIP build.
#pragma once

#include <stdint.h>

#include "../../include/kp_spec.h"

#if defined(__HIPCC__)
#define KP_HD __host__ __device__ __forceinline__
#else
#define KP_HD inline
#endif
#define KP_MAPQ_FN KP_HD
#include "../../include/kp_mapq.h"

// ---- records (public layouts live in include/kp_spec.h) -----------------------------------------------------------------
typedef kp_kept KpKept;
typedef kp_piece KpPiece;
typedef kp_asm_summary KpAsmSummary;
typedef kp_typing_params KpTypingParams;

typedef struct KpTypingDb {  // device-resident views of the Database arrays the reduction reads
    const uint16_t *gene_locus;     // db.gene_locus_indices
    const uint8_t *gene_extra;      // db.extra_genes
    const uint16_t *gene_pos;       // db.gene_positions
    const int8_t *gene_strand;      // db.gene_intervals.strands
    const int32_t *gene_len;        // db.genes.lengths
    const int32_t *locus_gene_off;  // db.locus_gene_offsets
    const int32_t *locus_gene_len;  // db.locus_gene_lengths
    const uint8_t *prot;            // db.translations (bytes, stop codons kept)
    const int32_t *prot_off, *prot_len;
    int32_t n_genes, n_loci;
} KpTypingDb;

// ---- hit finalisation -----------------------------------------------------------------------------------------------
// task result -> hit record (strand flip, contig-local coordinates); mapq is filled after sorting -- until then the
// (mapq, pad_) bytes carry the chain's score (low byte in mapq, clamped to 65535), which only the mapq computation reads
KP_HD int kp_hit_chain_score(const kp_hit &h) { return (int)h.mapq | ((int)h.pad_ << 8); }
KP_HD kp_hit kp_make_hit(int gs, int contig, int32_t ctg_start, int qlen, int score, int q_start, int q_end,
                         int t_start, int t_end, int matches, int block_len, int n_anchors, int chain_score) {
    kp_hit h;
    const int rev = gs & 1;
    h.gene = gs >> 1; h.contig = contig;
    h.q_start = rev ? qlen - q_end : q_start;
    h.q_end = rev ? qlen - q_start : q_end;
    h.t_start = t_start - ctg_start; h.t_end = t_end - ctg_start;
    h.score = score; h.matches = matches; h.block_len = block_len;
    const int cs = chain_score < 65535 ? chain_score : 65535;
    h.strand = rev ? -1 : 1; h.mapq = (uint8_t)(cs & 255); h.n_seeds = (uint8_t)(n_anchors < 255 ? n_anchors : 255); h.pad_ = (uint8_t)(cs >> 8);
    return h;
}

// emission order of kp_spec.h as three ascending 64-bit keys
KP_HD void kp_hit_keys(const kp_hit &h, uint64_t k[3]) {
    k[0] = ((uint64_t)(uint32_t)h.gene << 40) | ((uint64_t)(0xFFFFFu - (uint32_t)KP_HIT_OSCORE(h.score)) << 20) | (uint32_t)h.contig;
    k[1] = ((uint64_t)(uint32_t)h.t_start << 32) | ((uint64_t)(h.strand < 0 ? 1u : 0u) << 31) |
           ((uint64_t)(uint32_t)h.q_start << 16) | (uint32_t)h.q_end;
    k[2] = ((uint64_t)(uint32_t)h.t_end << 32) | ((uint64_t)(0xFFFFu - (uint32_t)h.matches) << 16) | (uint32_t)h.block_len;
}

// (seeds: the last criteria of the emission order -- more seeds first, then the higher chain score; `seeds_*` carry both,
// kp_hit_seeds_key; the records' own indices break what is left, which are hits equal in every field)
KP_HD uint32_t kp_hit_seeds_key(const kp_hit &h) { return ((uint32_t)h.n_seeds << 16) | (uint32_t)kp_hit_chain_score(h); }
KP_HD bool kp_keys_less(const uint64_t a[3], uint32_t seeds_a, uint32_t ia, const uint64_t b[3], uint32_t seeds_b, uint32_t ib) {
    if (a[0] != b[0]) return a[0] < b[0];
    if (a[1] != b[1]) return a[1] < b[1];
    if (a[2] != b[2]) return a[2] < b[2];
    if (seeds_a != seeds_b) return seeds_a > seeds_b;
    return ia < ib;
}

// Primary / secondary and mapping quality of one gene's hits, in emission order (kp_spec.h): minimap2's mm_set_parent
// (mask level 1/2 with the uncovered-length correction) and mm_set_mapq on the finished hits.  `parent`, `subsc`, `dp2`,
// `n_sub`: scratch of n ints each.  Cost: a hit is compared with the PRIMARY hits before it and stops at the first one
// that masks it.  Primaries of one gene overlap each other little on the query, and a hit spans at least
// KP_MIN_CHAIN_SCORE query bases, so a gene has a handful of them (the gene and its fragments); the n copies of a
// multi-copy gene (IS elements, hundreds of hits) all cover the same query span, so every one after the first stops at
// j = 0: n steps, not n^2.
KP_HD void kp_assign_mapq(kp_hit *h, int n, int32_t *parent, int32_t *subsc, int32_t *dp2, int32_t *n_sub, const float *ln_half,
                          const float *ln_int) {
#if defined(__clang__)
#pragma clang fp contract(off)
#endif
    for (int i = 0; i < n; ++i) { subsc[i] = 0; dp2[i] = 0; n_sub[i] = 0; }
    for (int i = 0; i < n; ++i) {
        parent[i] = i;
        const int si = h[i].q_start, ei = h[i].q_end;
        // bases of [si, ei) that no earlier primary hit covers: sweep from si, jumping to the furthest end any interval that
        // has started reaches, or -- over a hole -- to the next start
        int uncov = 0;
        bool any = false;
        for (int x = si; x < ei;) {
            int reach = x, next = ei;
            for (int j = 0; j < i; ++j) {
                if (parent[j] != j || h[j].q_end <= si || h[j].q_start >= ei) continue;
                any = true;
                const int sj = h[j].q_start > si ? h[j].q_start : si, ej = h[j].q_end < ei ? h[j].q_end : ei;
                if (sj <= x) { if (ej > reach) reach = ej; }
                else if (sj < next) next = sj;
            }
            if (reach > x) x = reach;
            else { uncov += next - x; x = next; }
        }
        if (!any) continue;
        for (int j = 0; j < i; ++j) {
            if (parent[j] != j || h[j].q_end <= si || h[j].q_start >= ei) continue;
            const int sj = h[j].q_start, ej = h[j].q_end;
            const int mn = ej - sj < ei - si ? ej - sj : ei - si, mx = ej - sj > ei - si ? ej - sj : ei - si;
            const int ol = (ei < ej ? ei : ej) - (si > sj ? si : sj);
            const float lhs = (float)ol / (float)mn, rhs = (float)uncov / (float)mx;
            if (lhs - rhs > 0.5f) {  // KP_MASK_LEVEL
                bool cnt_sub = h[i].n_seeds >= h[j].n_seeds;
                const int sci = kp_hit_chain_score(h[i]);
                parent[i] = j;
                if (sci > subsc[j]) subsc[j] = sci;
                if (h[j].contig != h[i].contig || h[j].t_start != h[i].t_start || h[j].t_end != h[i].t_end || ol != mn) {
                    if (KP_HIT_OSCORE(h[i].score) > dp2[j]) dp2[j] = KP_HIT_OSCORE(h[i].score);
                    if (KP_HIT_OSCORE(h[j].score) - KP_HIT_OSCORE(h[i].score) <= 2 * KP_SC_MATCH - KP_SC_MISMATCH) cnt_sub = true;
                }
                if (cnt_sub) n_sub[j]++;
                break;
            }
        }
    }
    for (int i = 0; i < n; ++i) {
        const int cs = kp_hit_chain_score(h[i]);
        h[i].pad_ = 0;
        h[i].mapq = parent[i] != i ? (uint8_t)0
                                    : (uint8_t)kp_mapq_value(KP_HIT_OSCORE(h[i].score), cs, h[i].n_seeds, h[i].matches, h[i].block_len, subsc[i], dp2[i],
                                                             n_sub[i], ln_half, ln_int);
    }
    for (int i = 0; i < n; ++i) h[i].score = KP_HIT_SCORE(h[i].score);  // (kp_spec.h, order score: the finished record holds the plain score)
}

KP_HD bool kp_same_span(const kp_hit &x, const kp_hit &y) {
    return x.gene == y.gene && x.contig == y.contig && x.strand == y.strand && x.q_start == y.q_start &&
           x.q_end == y.q_end && x.t_start == y.t_start && x.t_end == y.t_end;
}

// ---- scoring (core.py:164-201) ----------------------------------------------------------------------------------------
// hits are in emission order (gene ascending).  Best hit of gene g = highest query coverage among hits with coverage >=
// min_cov (core.py:174-182 breaks ties by score, but only the coverage enters the locus score).
KP_HD int kp_lower_bound_gene(const kp_hit *hits, int n, int gene) {
    int lo = 0, hi = n;
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (hits[mid].gene < gene) lo = mid + 1; else hi = mid;
    }
    return lo;
}

KP_HD bool kp_gene_best_cov(const kp_hit *hits, int n, int gene, int gene_len, double min_cov, double *cov_out) {
    bool found = false;
    double best = 0.0;
    for (int i = kp_lower_bound_gene(hits, n, gene); i < n && hits[i].gene == gene; ++i) {
        const double cov = gene_len > 0 ? (double)(hits[i].q_end - hits[i].q_start) / (double)gene_len : 0.0;
        if (cov >= min_cov && (!found || cov > best)) { best = cov; found = true; }
    }
    *cov_out = best;
    return found;
}

// locus l: sum of best coverages of its non-extra genes in gene order (float64, same order as np.add.at), and count
KP_HD void kp_locus_score(const kp_hit *hits, int n, const KpTypingDb &db, int locus, double min_cov, double *score,
                          int32_t *count) {
    double s = 0.0;
    int c = 0;
    const int g0 = db.locus_gene_off[locus], g1 = g0 + db.locus_gene_len[locus];
    for (int g = g0; g < g1; ++g) {
        if (db.gene_extra[g]) continue;
        double cov;
        if (kp_gene_best_cov(hits, n, g, db.gene_len[g], min_cov, &cov)) { s += cov; ++c; }
    }
    *score = s;
    *count = c;
}

// ---- overlap cull (alignment.py:643-686, interval.py:698-751) ------------------------------------------------------------
// visit order: (score + 1e9 * priority) desc, matches desc, uint8-wrapped (-mapq) asc, emission index asc (stable lexsort)
KP_HD uint64_t kp_cull_key(const kp_hit &h, bool priority, uint32_t idx) {
    const uint32_t s = ((priority ? 1u : 0u) << 20) | (uint32_t)h.score;  // scores are < 2^20
    const uint32_t negq = (uint32_t)(uint8_t)(0u - (uint32_t)h.mapq);
    return ((uint64_t)(0x1FFFFFu - s) << 43) | ((uint64_t)(0xFFFFu - (uint32_t)h.matches) << 27) |
           ((uint64_t)negq << 19) | (uint64_t)idx;  // idx < 2^19
}

// greedy pass over `order`; kept_flag[i] set for survivors.  kept_* are scratch of capacity cap; returns survivors
// (or -1 when cap is too small).
KP_HD int kp_cull_sequential(const kp_hit *hits, int n, const uint32_t *order, uint8_t *kept_flag, int32_t *kept_ctg,
                             int32_t *kept_s, int32_t *kept_e, int cap) {
    int nk = 0;
    for (int p = 0; p < n; ++p) {
        const kp_hit &h = hits[order[p]];
        const int s = h.t_start, e = h.t_end, len = e - s;
        kept_flag[order[p]] = 0;
        if (len <= 0) continue;
        bool clash = false;
        for (int j = 0; j < nk && !clash; ++j) {
            if (kept_ctg[j] != h.contig) continue;
            const int ov = (e < kept_e[j] ? e : kept_e[j]) - (s > kept_s[j] ? s : kept_s[j]);
            const int klen = kept_e[j] - kept_s[j];
            // ov / min(len, klen) > 0.1 in float64 <=> 10 * ov > min(len, klen) for these integer ranges
            if (ov > 0 && (int64_t)ov * 10 > (int64_t)(len < klen ? len : klen)) clash = true;
        }
        if (clash) continue;
        if (nk >= cap) return -1;
        kept_ctg[nk] = h.contig; kept_s[nk] = s; kept_e[nk] = e;
        ++nk;
        kept_flag[order[p]] = 1;
    }
    return nk;
}

// ---- clustering, pieces, inside, missing (core.py:219-301) ---------------------------------------------------------------
// What the database says about a kept hit's gene, next to the hit (optional): on the device one lane runs this function
// and every look-up by gene index would be a trip to global memory on its own.
typedef struct KpGeneInfo {
    uint16_t locus, pos;
    int8_t strand;
    uint8_t extra;
} KpGeneInfo;

// kept[] is in emission order.  scratch: perm[nk]; info (optional): nk entries; piece_tmp (optional): 3 * piece_cap ints,
// the pieces' contig / start / end where the caller can read them back fast.
KP_HD void kp_cluster_and_pieces(KpKept *kept, int nk, const KpTypingDb &db, int best_locus, int64_t tolerance,
                                 int32_t *perm, KpPiece *pieces, int piece_cap, KpAsmSummary *sum,
                                 const KpGeneInfo *info = nullptr, int32_t *piece_tmp = nullptr) {
    // stable order by (contig, t_start, t_end): insertion sort of indices (nk is small)
    for (int i = 0; i < nk; ++i) {
        int j = i;
        while (j > 0) {
            const KpKept &a = kept[perm[j - 1]], &b = kept[i];
            const bool gt = a.contig != b.contig ? a.contig > b.contig
                            : (a.t_start != b.t_start ? a.t_start > b.t_start : a.t_end > b.t_end);
            if (!gt) break;
            perm[j] = perm[j - 1];
            --j;
        }
        perm[j] = i;
    }
    int cur = 0;
    int64_t cur_e = 0;
    int cur_c = -1;
    for (int p = 0; p < nk; ++p) {  // single linkage with tolerance (interval.py:626-637)
        KpKept &k = kept[perm[p]];
        if (p == 0) { cur = 0; cur_e = k.t_end; cur_c = k.contig; }
        else if (k.contig == cur_c && (int64_t)k.t_start <= cur_e + tolerance) { if (k.t_end > cur_e) cur_e = k.t_end; }
        else { ++cur; cur_e = k.t_end; cur_c = k.contig; }
        k.cluster = cur;
    }
    const int n_clusters = nk ? cur + 1 : 0;
    // flags; the primary hit of an expected gene is its top-scoring kept hit, the earliest in emission order on ties
    // (core.py:236-245: first of lexsort((-score, gene))).  Emission order is gene ascending but NOT score descending: a joined
    // hit is ranked by its order score (kp_spec.h), which can put it before a hit with a higher alignment score.
    for (int i = 0; i < nk; ++i) {
        KpKept &k = kept[i];
        uint8_t f = 0;
        if (info ? info[i].extra : db.gene_extra[k.gene]) f |= KP_F_EXTRA;
        else if ((int)(info ? info[i].locus : db.gene_locus[k.gene]) == best_locus) f |= KP_F_EXPECTED;
        k.flags = f;
    }
    for (int i = 0; i < nk;) {
        int j = i, top = i;
        for (; j < nk && kept[j].gene == kept[i].gene; ++j)
            if (kept[j].score > kept[top].score) top = j;
        if (kept[top].flags & KP_F_EXPECTED) kept[top].flags |= KP_F_PRIMARY;
        i = j;
    }
    // one piece per cluster (ascending id) that holds a primary hit
    int np = 0;
    bool piece_overflow = false;
    for (int c = 0; c < n_clusters; ++c) {
        int ctg = -1, smin = 0, emax = 0, nprim = 0, vote = 0;
        int64_t pos_sum = 0;
        for (int i = 0; i < nk; ++i) {
            const KpKept &k = kept[i];
            if (k.cluster != c) continue;
            if (ctg < 0) ctg = k.contig;
            if (!(k.flags & KP_F_PRIMARY)) continue;
            if (nprim == 0 || k.t_start < smin) smin = k.t_start;
            if (nprim == 0 || k.t_end > emax) emax = k.t_end;
            pos_sum += info ? info[i].pos : db.gene_pos[k.gene];
            vote += (int)k.strand * (int)(info ? info[i].strand : db.gene_strand[k.gene]);
            ++nprim;
        }
        if (nprim == 0) continue;
        if (np >= piece_cap) { piece_overflow = true; break; }
        pieces[np].contig = ctg; pieces[np].start = smin; pieces[np].end = emax;
        pieces[np].strand = vote < 0 ? -1 : 1;
        pieces[np].mean_pos = (double)pos_sum / (double)nprim;
        if (piece_tmp) { piece_tmp[3 * np] = ctg; piece_tmp[3 * np + 1] = smin; piece_tmp[3 * np + 2] = emax; }
        ++np;
    }
    for (int i = 0; i < nk; ++i) {  // closed-interval overlap with any piece on the same contig
        KpKept &k = kept[i];
        for (int p = 0; p < np; ++p) {
            const int pc = piece_tmp ? piece_tmp[3 * p] : pieces[p].contig, ps = piece_tmp ? piece_tmp[3 * p + 1] : pieces[p].start,
                      pe = piece_tmp ? piece_tmp[3 * p + 2] : pieces[p].end;
            if (k.contig == pc && k.t_start <= pe && k.t_end >= ps) {
                k.flags |= KP_F_INSIDE;
                break;
            }
        }
    }
    // expected genes of the best locus that were not found inside
    const int g0 = db.locus_gene_off[best_locus], gl = db.locus_gene_len[best_locus];
    for (int w = 0; w < KP_MAX_LOCUS_GENES / 64; ++w) sum->missing_mask[w] = 0;
    int n_exp = 0, n_missing = 0;
    for (int j = 0; j < gl; ++j) {
        const int g = g0 + j;
        if (db.gene_extra[g]) continue;  // an "Extra genes" record has no expected genes
        ++n_exp;
        bool found = false;
        for (int i = 0; i < nk && !found; ++i)
            found = kept[i].gene == g && (kept[i].flags & (KP_F_EXPECTED | KP_F_INSIDE)) == (KP_F_EXPECTED | KP_F_INSIDE);
        if (!found) {
            ++n_missing;
            if (j < KP_MAX_LOCUS_GENES) sum->missing_mask[j >> 6] |= (uint64_t)1 << (j & 63);
        }
    }
    sum->n_pieces = np;
    sum->n_expected = n_exp;
    sum->n_missing = n_missing;
    if (piece_overflow) sum->overflow |= 2;
    if (gl > KP_MAX_LOCUS_GENES) sum->overflow |= 4;
}

// ---- extraction + translation from the packed stream (seq.py:612-741; models.py:252-259) -----------------------------------
// base code at assembly position t: 0..3, or 4 inside an N run
KP_HD int kp_code_at(const uint32_t *asm_words, const int32_t *runs, int n_runs, int32_t t) {
    if (n_runs > 0) {
        int lo = 0, hi = n_runs;
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (runs[2 * mid + 1] <= t) lo = mid + 1; else hi = mid;
        }
        if (lo < n_runs && runs[2 * lo] <= t) return 4;
    }
    return (int)((asm_words[t >> 4] >> (2 * (t & 15))) & 3u);
}

// amino acid of codon `c` (0-based) of the hit's extracted strand; the table is NCBI 11 indexed a*25+b*5+c with N=4
KP_HD uint8_t kp_codon_aa(const uint32_t *asm_words, const int32_t *runs, int n_runs, int32_t abs_start,
                          int32_t abs_end, int strand, int frame, int c, const uint8_t *codon_table) {
    int code[3];
    for (int x = 0; x < 3; ++x) {
        const int off = frame + 3 * c + x;  // offset in the extracted (strand-corrected) sequence
        if (strand >= 0) code[x] = kp_code_at(asm_words, runs, n_runs, abs_start + off);
        else {
            const int v = kp_code_at(asm_words, runs, n_runs, abs_end - 1 - off);
            code[x] = v > 3 ? 4 : 3 - v;
        }
    }
    return codon_table[code[0] * 25 + code[1] * 5 + code[2]];
}

KP_HD void kp_fill_codon_table(uint8_t *t /*125*/) {  // seq.py:418-499, amino acids in TCAG order
    const char *aa = "FFLLSSSSYY**CC*WLLLLPPPPHHQQRRRRIIIMTTTTNNKKSSRRVVVVAAAADDEEGGGG";
    const int tcag[4] = {3, 1, 0, 2};
    for (int i = 0; i < 125; ++i) t[i] = 'X';
    for (int a = 0; a < 4; ++a)
        for (int b = 0; b < 4; ++b)
            for (int c = 0; c < 4; ++c) t[tcag[a] * 25 + tcag[b] * 5 + tcag[c]] = (uint8_t)aa[a * 16 + b * 4 + c];
}

// ---- float32 sum with numpy's association (np.add.reduce on a contiguous float32 array) -----------------------------------
// np.mean(float32 array) = float32(float64(reduce) / n) where reduce is numpy's blocked pairwise summation over the
// whole array (checked against np.add.reduce for every length class in tests/test_reduce_core_cpu.py): < 8 items sequential from 0, <= 128 items eight running sums combined as a tree, else split in two
// (multiple of 8) halves.  Reproduced here so that the batched mean identity has the reference's bits (core.py:395-396).
KP_HD float kp_np_pairwise_f32(const float *a, int n) {
    if (n < 8) {
        float res = 0.f;
        for (int i = 0; i < n; ++i) res += a[i];
        return res;
    }
    if (n <= 128) {
        float r[8];
        for (int j = 0; j < 8; ++j) r[j] = a[j];
        int i = 8;
        for (; i < n - (n % 8); i += 8)
            for (int j = 0; j < 8; ++j) r[j] += a[i + j];
        float res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
        for (; i < n; ++i) res += a[i];
        return res;
    }
    int n2 = n / 2;
    n2 -= n2 % 8;
    return kp_np_pairwise_f32(a, n2) + kp_np_pairwise_f32(a + n2, n - n2);
}

KP_HD float kp_np_sum_f32(const float *a, int n) { return n <= 0 ? 0.f : kp_np_pairwise_f32(a, n); }

// ---- gene states (core.py:363-394; alignment.py:774-809) ------------------------------------------------------------------
KP_HD void kp_gene_state(KpKept *k, int gene_len, int contig_len, const KpTypingParams &prm) {
    const int total = k->dp[1] + k->dp[2] + k->dp[3];
    const double pid = total > 0 ? ((double)k->dp[1] * 100.0) / (double)total : 0.0;
    k->pident = (float)pid;
    const double prot_cov = ((double)k->prot_len * 3.0) / (double)gene_len;
    double c = prot_cov * 100.0;
    c = c < 0.0 ? 0.0 : (c > 100.0 ? 100.0 : c);
    k->coverage = (float)c;
    const bool fwd = k->strand == 1;
    const bool left = k->t_start <= prm.edge_tolerance && (fwd ? k->q_start > 0 : k->q_end < gene_len);
    const bool right = k->t_end >= contig_len - prm.edge_tolerance && (fwd ? k->q_end < gene_len : k->q_start > 0);
    int state = KP_STATE_NORMAL;
    if (left || right) { state = KP_STATE_PARTIAL; k->flags |= KP_F_PARTIAL; }
    else if (prot_cov < 0.90) state = KP_STATE_TRUNCATED;
    if (!(k->flags & KP_F_INSIDE) && k->pident < prm.id_threshold) k->flags |= KP_F_SPURIOUS;
    if (state == KP_STATE_NORMAL && k->pident < prm.id_threshold) state = KP_STATE_NOVEL;
    k->state = (int8_t)state;
}
