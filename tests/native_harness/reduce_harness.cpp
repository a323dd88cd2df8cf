// reduce_harness.cpp -- TEST-ONLY: compiles kaptive_amd/csrc/kp_reduce_core.h with g++ and walks one assembly through
// the same sequence of core calls the HIP kernels of kp_reduce.hip make, serially.  It lets the GPU-less container
// check the reduction logic against the golden vectors.  It is not part of the product and is never loaded by it.
#include <algorithm>
#include <cmath>
#include <cstring>
#include <vector>

#include "../../kaptive_amd/csrc/kp_reduce_core.h"

extern "C" {

// hits: unsorted raw hits in, emission-ordered + deduplicated + mapq out (returns the new count)
int kph_finalise_hits(kp_hit *hits, int n) {
    std::vector<kp_hit> raw(hits, hits + n), sorted((size_t)n);
    std::vector<uint64_t> keys(3 * (size_t)n);
    for (int i = 0; i < n; ++i) kp_hit_keys(raw[i], &keys[3 * (size_t)i]);
    for (int i = 0; i < n; ++i) {  // rank sort, as the kernel does
        int rank = 0;
        for (int j = 0; j < n; ++j)
            rank += kp_keys_less(&keys[3 * (size_t)j], kp_hit_seeds_key(raw[j]), j, &keys[3 * (size_t)i], kp_hit_seeds_key(raw[i]), i);
        sorted[(size_t)rank] = raw[i];
    }
    int m = 0;
    for (int i = 0; i < n; ++i) {
        if (m > 0 && kp_same_span(hits[m - 1], sorted[i])) continue;
        hits[m++] = sorted[i];
    }
    std::vector<float> ln_half(KP_MAPQ_LN_HALF_SIZE), ln_int(KP_MAPQ_LN_INT_SIZE);
    for (int i = 0; i < KP_MAPQ_LN_HALF_SIZE; ++i) ln_half[i] = i ? kp_mapq_ln((double)i / 2.0) : 0.0f;
    for (int i = 0; i < KP_MAPQ_LN_INT_SIZE; ++i) ln_int[i] = i ? kp_mapq_ln((double)i) : 0.0f;
    std::vector<int32_t> scratch(4 * (size_t)(m > 0 ? m : 1));
    for (int i = 0; i < m;) {
        int j = i;
        while (j < m && hits[j].gene == hits[i].gene) ++j;
        kp_assign_mapq(hits + i, j - i, scratch.data(), scratch.data() + m, scratch.data() + 2 * (size_t)m, scratch.data() + 3 * (size_t)m,
                       ln_half.data(), ln_int.data());
        i = j;
    }
    return m;
}

void kph_locus_scores(const kp_hit *hits, int n, const KpTypingDb *db, double min_cov, double *scores, int32_t *counts) {
    for (int l = 0; l < db->n_loci; ++l) kp_locus_score(hits, n, *db, l, min_cov, &scores[l], &counts[l]);
}

// cull + kept list + clustering + pieces + translation; returns n_kept (or -1 on kept overflow)
int kph_reduce(const kp_hit *hits, int n, const KpTypingDb *db, const KpTypingParams *prm, int best_locus,
               const uint32_t *asm_words, const int32_t *ctg_start, const int32_t *n_runs, int n_nruns, KpKept *kept,
               int kept_cap, KpPiece *pieces, int piece_cap, KpAsmSummary *sum, uint8_t *prot, int prot_cap) {
    std::memset(sum, 0, sizeof *sum);
    sum->n_hits = n;
    sum->best_locus = best_locus;
    std::vector<uint64_t> keys((size_t)n);
    std::vector<uint32_t> order((size_t)n);
    for (int i = 0; i < n; ++i) keys[(size_t)i] = kp_cull_key(hits[i], (int)db->gene_locus[hits[i].gene] == best_locus, (uint32_t)i);
    for (int i = 0; i < n; ++i) {
        int rank = 0;
        for (int j = 0; j < n; ++j) rank += keys[(size_t)j] < keys[(size_t)i];
        order[(size_t)rank] = (uint32_t)i;
    }
    std::vector<uint8_t> flag((size_t)std::max(n, 1));
    std::vector<int32_t> kc((size_t)kept_cap), ks((size_t)kept_cap), ke((size_t)kept_cap), perm((size_t)kept_cap);
    int nk;
    if (n < 2) {  // the reference returns the table unchanged below two hits (alignment.py:665-666)
        nk = n;
        if (n) flag[0] = 1;
    } else nk = kp_cull_sequential(hits, n, order.data(), flag.data(), kc.data(), ks.data(), ke.data(), kept_cap);
    if (nk < 0) { sum->overflow |= 1; return -1; }
    int k = 0;
    for (int i = 0; i < n; ++i) {
        if (!flag[(size_t)i]) continue;
        KpKept &o = kept[k++];
        std::memset(&o, 0, sizeof o);
        o.gene = hits[i].gene; o.contig = hits[i].contig; o.q_start = hits[i].q_start; o.q_end = hits[i].q_end;
        o.t_start = hits[i].t_start; o.t_end = hits[i].t_end; o.score = hits[i].score; o.strand = hits[i].strand;
    }
    sum->n_kept = nk;
    kp_cluster_and_pieces(kept, nk, *db, best_locus, prm->max_locus_length, perm.data(), pieces, piece_cap, sum);
    uint8_t table[125];
    kp_fill_codon_table(table);
    int used = 0;
    for (int i = 0; i < nk; ++i) {
        KpKept &o = kept[i];
        const int frame = (3 - o.q_start % 3) % 3, len = o.t_end - o.t_start;
        const int max_codons = len > frame ? (len - frame) / 3 : 0;
        if (used + max_codons > prot_cap) { sum->overflow |= 8; return -1; }
        const int32_t a0 = ctg_start[o.contig] + o.t_start, a1 = ctg_start[o.contig] + o.t_end;
        int nc = 0;
        for (; nc < max_codons; ++nc) {
            const uint8_t aa = kp_codon_aa(asm_words, n_runs, n_nruns, a0, a1, o.strand, frame, nc, table);
            if (aa == '*') break;
            prot[used + nc] = aa;
        }
        o.prot_off = used; o.prot_len = nc;
        used += max_codons;
    }
    return nk;
}

void kph_states(KpKept *kept, int nk, const KpTypingDb *db, const KpTypingParams *prm, const int32_t *ctg_len,
                const int32_t *dp8, KpAsmSummary *sum) {
    std::vector<float> vals;
    for (int i = 0; i < nk; ++i) {
        std::memcpy(kept[i].dp, dp8 + 8 * (size_t)i, 8 * sizeof(int32_t));
        kp_gene_state(&kept[i], db->gene_len[kept[i].gene], ctg_len[kept[i].contig], *prm);
        if (!(kept[i].flags & KP_F_SPURIOUS) && kept[i].state == KP_STATE_NORMAL) vals.push_back(kept[i].pident);
    }
    sum->n_normal = (int32_t)vals.size();
    sum->ident_sum = kp_np_sum_f32(vals.data(), (int)vals.size());
}

float kph_np_sum_f32(const float *a, int n) { return kp_np_sum_f32(a, n); }

int kph_sizes(int *out) {
    out[0] = (int)sizeof(kp_hit); out[1] = (int)sizeof(KpKept); out[2] = (int)sizeof(KpPiece);
    out[3] = (int)sizeof(KpAsmSummary); out[4] = (int)sizeof(KpTypingDb); out[5] = (int)sizeof(KpTypingParams);
    return 6;
}

}  // extern "C"
