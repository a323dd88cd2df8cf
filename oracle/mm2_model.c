/* mm2_model.c -- an INDEPENDENT CPU model of the published minimap2 mapping pipeline.  TEST INFRASTRUCTURE ONLY.
 *
 * Why it exists.  The reference aligns every database gene against the assembly with the closed third-party wheel
 * rammappy 0.1.3 (src/kaptive/serotyping/core.py:147-155: Index.build over the contigs, Aligner(preset=None),
 * best_n=50000, pri_ratio=0.0, map_batch over the genes), which docs/serotyping/method.md:23-28 describes as a
 * "minimap2-based banded dynamic programming aligner" and whose hit objects carry minimap2's mm_reg1_t fields
 * (src/kaptive/core/alignment.py:415-446).  The wheel is neither in the reference tree nor installable here, so the
 * product's aligner (include/kp_spec.h, "kp-align") cannot be compared with it.  This file restates the PUBLISHED
 * algorithm of minimap2 (Li 2018, Bioinformatics 34:3094; Li 2021, Bioinformatics 37:4572 for the chaining gap cost;
 * Suzuki & Kasahara 2018 for the DP formulation) with minimap2's documented defaults for "no preset", so that the
 * two can be set side by side: tools/concordance.py replays this model's hits and kp-align's hits through the
 * reference's own Serotyper and reports how often a KaptiveRow differs (profiles/concordance_r3.md).
 *
 * PARITY UNPINNED: nothing here has been checked against rammappy or a minimap2 binary (neither exists in the build
 * container; there is no network).  It was written from the papers and from knowledge of minimap2's sources
 * (2.24-2.28 behaviour: sketch.c, seed.c, lchain.c, hit.c, align.c, ksw2_extd2) without access to them; it is a
 * MODEL, not a port.  It shares no code, data structure or parameter header with the product or with kp_oracle.c.
 * Only tests/ and tools/concordance.py use it; nothing under kaptive_amd/ does.
 *
 * What is modelled (function names are minimap2's):
 *   mm_sketch            (w=10,k=15) minimizers of hash64(min(fwd,rev)), symmetric k-mers skipped, ties reported
 *   mm_idx_cal_max_occ   -f 2e-4 occurrence cut, clamped to [min_mid_occ=10, max_mid_occ=1e6]
 *   mm_collect_matches   seeds above mid_occ dropped; anchors (strand|rid|rpos, span|qpos) sorted
 *   mm_chain_dp          max_gap 5000, bw 500, max_skip 25, max_iter 5000, -n 3, -m 40, gap cost 0.12*dd + 0.5*log2(dd+1)
 *   mg_chain_backtrack   peak-score backtracking with max_drop = bw
 *   mm_gen_regs / mm_set_parent (-M 0.5, uncovered-length correction) / mm_select_sub (pri_ratio 0: keeps all)
 *   mm_align1            extension limits from neighbouring seeds, left z-drop extension, global fills between anchors
 *                        >= 200 apart (min_ksw_len), mm_test_zdrop + split at a z-drop, right z-drop extension,
 *                        scores 2/-4, gaps min(4+2n, 24+n), N = -1, z-drop 400, band 1.5*bw+1
 *   mm_update_extra      blen/mlen, dp_max with the log-gap cost; leading/trailing gaps trimmed (mm_fix_cigar)
 *   mm_filter_regs       mlen >= 40 and dp_max >= 80 (-s 80); mm_hit_sort by dp_max; mm_set_parent; mm_set_mapq
 *   (round 6) mm_seed_select   high-occurrence seeds: per streak of them up to (query extent / occ_dist 500) of the least frequent are
 *                        kept (max_max_occ 4095); rep_len of the dropped ones scales the mapping quality (uniq_ratio)
 *   (round 6) mm_fix_bad_ends  a chain end whose anchors cover less than twice the gap that follows them is cut off before the
 *                        alignment (the extension may still reach across); mm_filter_bad_seeds / mm_filter_bad_seeds_alt:
 *                        anchors between two long gaps that nearly cancel, or that lie closer together than they are long, are
 *                        skipped as fill boundaries (MM_SEED_IGNORE / MM_SEED_LONG_JOIN); mm_adjust_minier: fills start and end
 *                        in the middle of an anchor's k-mer; extension windows reach back to the uncut chain's ends
 * What is NOT modelled (stated, not hidden): the re-chaining passes (max_occ rescue, RMQ long-join with bw_long), MM_SEED_TANDEM
 * (equal neighbouring minimizers of the query), inversion detection, the SSE kernel's approximate-max shortcut (the exact z-drop
 * test is always used), hash-based tie order of equal scores (stable order instead).  For 0.6-1.8 kb gene queries against
 * bacterial contigs none of these is expected to act.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define MM_API __attribute__((visibility("default")))

/* ---- options: minimap2's defaults without a preset ------------------------------------------------------------------ */
enum { MM_K = 15, MM_W = 10, MM_A = 2, MM_B = 4, MM_Q = 4, MM_E = 2, MM_Q2 = 24, MM_E2 = 1, MM_SC_AMBI = 1 };
enum { MM_ZDROP = 400, MM_BW = 500, MM_MAX_GAP = 5000, MM_MAX_SKIP = 25, MM_MAX_ITER = 5000, MM_MIN_CNT = 3 };
enum { MM_MIN_CHAIN_SC = 40, MM_MIN_DP_MAX = 80, MM_MIN_KSW_LEN = 200, MM_MIN_MID_OCC = 10, MM_MAX_MID_OCC = 1000000 };
enum { MM_MAX_MAX_OCC = 4095, MM_OCC_DIST = 500, MM_MAX_MAX_HIGH_OCC = 128 };
#define MM_SEED_LONG_JOIN (1ULL << 40)
#define MM_SEED_IGNORE (1ULL << 41)
static int g_end_flt = 1, g_seed_select = 1; /* test switches (mm2_set_options): the round-5 model had neither */
static const float MM_MID_OCC_FRAC = 2e-4f, MM_MASK_LEVEL = 0.5f, MM_CHAIN_GAP_SCALE = 0.8f;

typedef struct {
    uint64_t x, y;
} mm128;

typedef struct { /* the hit record shared with the rest of the test infrastructure (numpy HIT_DTYPE) */
    int32_t gene, contig, q_start, q_end, t_start, t_end, score, matches, block_len;
    int8_t strand;
    uint8_t mapq, n_seeds, pad_;
} mm2_hit;

typedef struct {
    int n_ctg;
    const uint8_t *seq; /* borrowed: ASCII contigs back to back */
    int64_t *off;
    int32_t *len;
    uint8_t **code; /* per contig: 0..3, 4 = ambiguous */
    mm128 *mz;      /* all minimizers sorted by hash (x>>8), then by y */
    int64_t n_mz;
    int32_t mid_occ;
} mm2_index;

static const uint8_t *nt4(void) {
    static uint8_t t[256];
    static int ready = 0;
    if (!ready) {
        memset(t, 4, 256);
        t['A'] = t['a'] = 0, t['C'] = t['c'] = 1, t['G'] = t['g'] = 2, t['T'] = t['t'] = 3, t['U'] = t['u'] = 3;
        ready = 1;
    }
    return t;
}

static inline uint64_t hash64(uint64_t key, uint64_t mask) { /* Thomas Wang's invertible integer hash, as in sketch.c */
    key = (~key + (key << 21)) & mask;
    key = key ^ key >> 24;
    key = ((key + (key << 3)) + (key << 8)) & mask;
    key = key ^ key >> 14;
    key = ((key + (key << 2)) + (key << 4)) & mask;
    key = key ^ key >> 28;
    key = (key + (key << 31)) & mask;
    return key;
}

typedef struct {
    mm128 *a;
    int64_t n, m;
} vec128;
static void push(vec128 *v, mm128 x) {
    if (v->n == v->m) {
        v->m = v->m ? v->m * 2 : 256;
        v->a = (mm128 *)realloc(v->a, (size_t)v->m * sizeof(mm128));
    }
    v->a[v->n++] = x;
}

/* mm_sketch: x = hash<<8 | span, y = rid<<32 | last_pos<<1 | strand */
static void sketch(const uint8_t *code, int len, uint32_t rid, vec128 *p) {
    const int w = MM_W, k = MM_K;
    const uint64_t shift1 = 2 * (k - 1), mask = (1ULL << 2 * k) - 1;
    uint64_t kmer[2] = {0, 0};
    int i, j, l, buf_pos, min_pos, kmer_span = 0;
    mm128 buf[256], min = {UINT64_MAX, UINT64_MAX};
    memset(buf, 0xff, (size_t)w * 16);
    for (i = l = buf_pos = min_pos = 0; i < len; ++i) {
        int c = code[i];
        mm128 info = {UINT64_MAX, UINT64_MAX};
        if (c < 4) {
            int z;
            kmer_span = l + 1 < k ? l + 1 : k;
            kmer[0] = (kmer[0] << 2 | (uint64_t)c) & mask;
            kmer[1] = (kmer[1] >> 2) | (3ULL ^ (uint64_t)c) << shift1;
            if (kmer[0] == kmer[1]) continue; /* symmetric k-mer: strand unknown */
            z = kmer[0] < kmer[1] ? 0 : 1;
            ++l;
            if (l >= k && kmer_span < 256) {
                info.x = hash64(kmer[z], mask) << 8 | (uint64_t)kmer_span;
                info.y = (uint64_t)rid << 32 | (uint32_t)i << 1 | (uint32_t)z;
            }
        } else
            l = 0, kmer_span = 0;
        buf[buf_pos] = info;
        if (l == w + k - 1 && min.x != UINT64_MAX) { /* first window: identical k-mers were not stored yet */
            for (j = buf_pos + 1; j < w; ++j)
                if (min.x == buf[j].x && buf[j].y != min.y) push(p, buf[j]);
            for (j = 0; j < buf_pos; ++j)
                if (min.x == buf[j].x && buf[j].y != min.y) push(p, buf[j]);
        }
        if (info.x <= min.x) { /* a new minimum; write the old one */
            if (l >= w + k && min.x != UINT64_MAX) push(p, min);
            min = info, min_pos = buf_pos;
        } else if (buf_pos == min_pos) { /* the old minimum left the window */
            if (l >= w + k - 1 && min.x != UINT64_MAX) push(p, min);
            for (j = buf_pos + 1, min.x = UINT64_MAX; j < w; ++j)
                if (min.x >= buf[j].x) min = buf[j], min_pos = j;
            for (j = 0; j <= buf_pos; ++j)
                if (min.x >= buf[j].x) min = buf[j], min_pos = j;
            if (l >= w + k - 1 && min.x != UINT64_MAX) {
                for (j = buf_pos + 1; j < w; ++j)
                    if (min.x == buf[j].x && min.y != buf[j].y) push(p, buf[j]);
                for (j = 0; j <= buf_pos; ++j)
                    if (min.x == buf[j].x && min.y != buf[j].y) push(p, buf[j]);
            }
        }
        if (++buf_pos == w) buf_pos = 0;
    }
    if (min.x != UINT64_MAX) push(p, min);
}

static int cmp128(const void *pa, const void *pb) {
    const mm128 *a = (const mm128 *)pa, *b = (const mm128 *)pb;
    if (a->x != b->x) return a->x < b->x ? -1 : 1;
    return a->y < b->y ? -1 : a->y > b->y;
}
static int cmp_u32(const void *a, const void *b) {
    uint32_t x = *(const uint32_t *)a, y = *(const uint32_t *)b;
    return x < y ? -1 : x > y;
}

MM_API void *mm2_index_build(const uint8_t *seq, const int64_t *off, const int32_t *len, int n_ctg) {
    mm2_index *mi = (mm2_index *)calloc(1, sizeof(mm2_index));
    const uint8_t *t = nt4();
    vec128 v = {0, 0, 0};
    mi->n_ctg = n_ctg, mi->seq = seq;
    mi->off = (int64_t *)malloc(sizeof(int64_t) * (size_t)(n_ctg + 1));
    mi->len = (int32_t *)malloc(sizeof(int32_t) * (size_t)(n_ctg + 1));
    mi->code = (uint8_t **)calloc((size_t)n_ctg + 1, sizeof(uint8_t *));
    for (int c = 0; c < n_ctg; ++c) {
        mi->off[c] = off[c], mi->len[c] = len[c];
        mi->code[c] = (uint8_t *)malloc((size_t)len[c] + 1);
        for (int32_t i = 0; i < len[c]; ++i) mi->code[c][i] = t[seq[off[c] + i]];
        if (len[c] > 0) sketch(mi->code[c], len[c], (uint32_t)c, &v);
    }
    /* index key = hash (the span is always k here); occurrences sorted by position */
    qsort(v.a, (size_t)v.n, sizeof(mm128), cmp128);
    mi->mz = v.a, mi->n_mz = v.n;
    { /* mm_idx_cal_max_occ(mi, 2e-4) then the [min_mid_occ, max_mid_occ] clamp of mm_mapopt_update */
        int64_t n = 0, i, j;
        uint32_t *cnt = (uint32_t *)malloc(sizeof(uint32_t) * (size_t)(v.n + 1));
        for (i = 0; i < v.n; i = j) {
            for (j = i + 1; j < v.n && v.a[j].x >> 8 == v.a[i].x >> 8; ++j) {}
            cnt[n++] = (uint32_t)(j - i);
        }
        int32_t thres = INT32_MAX;
        if (n > 0) {
            qsort(cnt, (size_t)n, sizeof(uint32_t), cmp_u32);
            int64_t kth = (int64_t)((1. - (double)MM_MID_OCC_FRAC) * (double)n);
            if (kth >= n) kth = n - 1;
            thres = (int32_t)cnt[kth] + 1;
        }
        free(cnt);
        if (thres < MM_MIN_MID_OCC) thres = MM_MIN_MID_OCC;
        if (thres > MM_MAX_MID_OCC) thres = MM_MAX_MID_OCC;
        mi->mid_occ = thres;
    }
    return mi;
}

MM_API void mm2_index_free(void *p) {
    mm2_index *mi = (mm2_index *)p;
    if (!mi) return;
    for (int c = 0; c < mi->n_ctg; ++c) free(mi->code[c]);
    free(mi->code), free(mi->off), free(mi->len), free(mi->mz), free(mi);
}

MM_API int32_t mm2_index_mid_occ(void *p) { return ((mm2_index *)p)->mid_occ; }
MM_API int64_t mm2_index_n_minimizers(void *p) { return ((mm2_index *)p)->n_mz; }

/* occurrences of one minimizer hash: [lo, hi) in mi->mz */
static void idx_get(const mm2_index *mi, uint64_t h, int64_t *lo_, int64_t *hi_) {
    int64_t lo = 0, hi = mi->n_mz;
    while (lo < hi) {
        int64_t mid = (lo + hi) >> 1;
        if (mi->mz[mid].x >> 8 < h) lo = mid + 1;
        else hi = mid;
    }
    *lo_ = lo;
    while (lo < mi->n_mz && mi->mz[lo].x >> 8 == h) ++lo;
    *hi_ = lo;
}

/* ---- chaining (lchain.c: mm_chain_dp) --------------------------------------------------------------------------------- */
static inline float mg_log2(float x) { /* fast approximate log2 for x >= 2, as minimap2 uses in the gap cost */
    union {
        float f;
        uint32_t i;
    } z = {x};
    float log_2 = (float)((int)((z.i >> 23) & 255) - 128);
    z.i &= ~(255U << 23);
    z.i += 127U << 23;
    log_2 += (-0.34484843f * z.f + 2.02466578f) * z.f - 0.67487759f;
    return log_2;
}

static inline int32_t comput_sc(const mm128 *ai, const mm128 *aj, int32_t max_dist_x, int32_t max_dist_y, int32_t bw,
                                float chn_pen_gap) {
    int32_t dq = (int32_t)ai->y - (int32_t)aj->y, dr, dd, dg, q_span, sc;
    if (dq <= 0 || dq > max_dist_x) return INT32_MIN;
    dr = (int32_t)(ai->x - aj->x);
    if (dr == 0 || dq > max_dist_y) return INT32_MIN;
    dd = dr > dq ? dr - dq : dq - dr;
    if (dd > bw) return INT32_MIN;
    dg = dr < dq ? dr : dq;
    q_span = (int32_t)(aj->y >> 32 & 0xff);
    sc = q_span < dg ? q_span : dg;
    if (dd || dg > q_span) {
        float lin_pen = chn_pen_gap * (float)dd; /* chain_skip_scale = 0 */
        float log_pen = dd >= 1 ? mg_log2((float)(dd + 1)) : 0.0f;
        sc -= (int)(lin_pen + .5f * log_pen);
    }
    return sc;
}

typedef struct {
    int32_t score, cnt; /* chain score, anchors */
    int64_t as;         /* first anchor in the compacted array */
} chain_t;

static int cmp_z_desc(const void *pa, const void *pb) { /* (f, index) descending: highest score first, later index first */
    const mm128 *a = (const mm128 *)pa, *b = (const mm128 *)pb;
    if (a->x != b->x) return a->x > b->x ? -1 : 1;
    return a->y > b->y ? -1 : a->y < b->y;
}

static int64_t chain_bk_end(int32_t max_drop, const mm128 *z, const int32_t *f, const int64_t *p, int32_t *t, int64_t k) {
    int64_t i = (int64_t)z[k].y, end_i = -1, max_i = i;
    int32_t max_s = 0;
    if (i < 0 || t[i] != 0) return i;
    do {
        int32_t s;
        t[i] = 2;
        end_i = i = p[i];
        s = i < 0 ? (int32_t)z[k].x : (int32_t)z[k].x - f[i];
        if (s > max_s) max_s = s, max_i = i;
        else if (max_s - s > max_drop) break;
    } while (i >= 0 && t[i] == 0);
    for (i = (int64_t)z[k].y; i >= 0 && i != end_i; i = p[i]) t[i] = 0;
    return max_i;
}

/* a[] sorted by x; returns chains and the compacted anchors (b, ascending inside a chain; chains by first target pos) */
static chain_t *chain_dp(int64_t n, const mm128 *a, int *n_chains, mm128 **b_out) {
    const int32_t max_dist_x = MM_MAX_GAP, max_dist_y = MM_MAX_GAP, bw = MM_BW, max_skip = MM_MAX_SKIP, max_iter = MM_MAX_ITER;
    const float chn_pen_gap = MM_CHAIN_GAP_SCALE * 0.01f * (float)MM_K;
    *n_chains = 0, *b_out = 0;
    if (n == 0) return 0;
    int32_t *f = (int32_t *)malloc(sizeof(int32_t) * (size_t)n), *t = (int32_t *)calloc((size_t)n, sizeof(int32_t));
    int32_t *v = (int32_t *)malloc(sizeof(int32_t) * (size_t)n);
    int64_t *p = (int64_t *)malloc(sizeof(int64_t) * (size_t)n);
    int64_t i, j, st = 0, max_ii = -1;
    for (i = 0; i < n; ++i) {
        int64_t max_j = -1, end_j;
        int32_t max_f = (int32_t)(a[i].y >> 32 & 0xff), n_skip = 0;
        while (st < i && (a[i].x >> 32 != a[st].x >> 32 || a[i].x > a[st].x + (uint64_t)max_dist_x)) ++st;
        if (i - st > max_iter) st = i - max_iter;
        for (j = i - 1; j >= st; --j) {
            int32_t sc = comput_sc(&a[i], &a[j], max_dist_x, max_dist_y, bw, chn_pen_gap);
            if (sc == INT32_MIN) continue;
            sc += f[j];
            if (sc > max_f) {
                max_f = sc, max_j = j;
                if (n_skip > 0) --n_skip;
            } else if (t[j] == (int32_t)i) {
                if (++n_skip > max_skip) break;
            }
            if (p[j] >= 0) t[p[j]] = (int32_t)i;
        }
        end_j = j;
        if (max_ii < 0 || a[i].x - a[max_ii].x > (uint64_t)max_dist_x) {
            int32_t max = INT32_MIN;
            max_ii = -1;
            for (j = i - 1; j >= st; --j)
                if (max < f[j]) max = f[j], max_ii = j;
        }
        if (max_ii >= 0 && max_ii < end_j) {
            int32_t tmp = comput_sc(&a[i], &a[max_ii], max_dist_x, max_dist_y, bw, chn_pen_gap);
            if (tmp != INT32_MIN && max_f < tmp + f[max_ii]) max_f = tmp + f[max_ii], max_j = max_ii;
        }
        f[i] = max_f, p[i] = max_j;
        v[i] = max_j >= 0 && v[max_j] > max_f ? v[max_j] : max_f;
        if (max_ii < 0 || (a[i].x - a[max_ii].x <= (uint64_t)max_dist_x && f[max_ii] < f[i])) max_ii = i;
    }
    /* mg_chain_backtrack(min_cnt=3, min_sc=40, max_drop=bw) */
    int64_t n_z = 0, k, n_v = 0;
    for (i = 0; i < n; ++i)
        if (f[i] >= MM_MIN_CHAIN_SC) ++n_z;
    chain_t *ch = 0;
    int n_u = 0;
    if (n_z > 0) {
        mm128 *z = (mm128 *)malloc(sizeof(mm128) * (size_t)n_z);
        for (i = 0, k = 0; i < n; ++i)
            if (f[i] >= MM_MIN_CHAIN_SC) z[k].x = (uint64_t)f[i], z[k++].y = (uint64_t)i;
        qsort(z, (size_t)n_z, sizeof(mm128), cmp_z_desc);
        memset(t, 0, sizeof(int32_t) * (size_t)n);
        ch = (chain_t *)malloc(sizeof(chain_t) * (size_t)n_z);
        for (k = 0; k < n_z; ++k) {
            if (t[z[k].y] != 0) continue;
            int64_t n_v0 = n_v, end_i = chain_bk_end(bw, z, f, p, t, k);
            for (i = (int64_t)z[k].y; i != end_i; i = p[i]) v[n_v++] = (int32_t)i, t[i] = 1;
            int32_t sc = i < 0 ? (int32_t)z[k].x : (int32_t)z[k].x - f[i];
            if (sc >= MM_MIN_CHAIN_SC && n_v > n_v0 && n_v - n_v0 >= MM_MIN_CNT) {
                ch[n_u].score = sc, ch[n_u].cnt = (int32_t)(n_v - n_v0), ch[n_u].as = n_v0;
                ++n_u;
            } else
                n_v = n_v0;
        }
        free(z);
    }
    /* compact_a: anchors of a chain in ascending order, chains ordered by their first anchor's target position */
    mm128 *b = (mm128 *)malloc(sizeof(mm128) * (size_t)(n_v + 1));
    if (n_u > 0) {
        mm128 *w = (mm128 *)malloc(sizeof(mm128) * (size_t)n_u);
        for (int c = 0; c < n_u; ++c) w[c].x = a[v[ch[c].as + ch[c].cnt - 1]].x, w[c].y = (uint64_t)c;
        qsort(w, (size_t)n_u, sizeof(mm128), cmp128);
        chain_t *ch2 = (chain_t *)malloc(sizeof(chain_t) * (size_t)n_u);
        int64_t kk = 0;
        for (int c = 0; c < n_u; ++c) {
            const chain_t *s = &ch[w[c].y];
            ch2[c] = *s, ch2[c].as = kk;
            for (int32_t q = 0; q < s->cnt; ++q) b[kk++] = a[v[s->as + (s->cnt - q - 1)]];
        }
        free(w), free(ch), ch = ch2;
    }
    free(f), free(t), free(v), free(p);
    *n_chains = n_u, *b_out = b;
    return ch;
}

/* ---- regions ------------------------------------------------------------------------------------------------------------- */
typedef struct {
    int32_t id, cnt, rid, score; /* score = chain score */
    int32_t qs, qe, rs, re;      /* query on the forward strand of the gene */
    int32_t parent, subsc, n_sub;
    int64_t as;
    int32_t mlen, blen, dp_score, dp_max, dp_max2;
    int32_t fuzzy_mlen; /* mm_cal_fuzzy_len on the chain: what mm_fix_bad_ends measures an end against */
    int32_t order_key;  /* stable tie order */
    uint8_t rev, has_p, mapq, dead;
} reg_t;

static void set_parent(int n, reg_t *r, int sub_diff) { /* hit.c: mm_set_parent, mask_level 0.5, soft masking */
    if (n <= 0) return;
    int i, j, k, *w = (int *)malloc(sizeof(int) * (size_t)n);
    uint64_t *cov = (uint64_t *)malloc(sizeof(uint64_t) * (size_t)n);
    for (i = 0; i < n; ++i) r[i].id = i;
    w[0] = 0, r[0].parent = 0;
    for (i = 1, k = 1; i < n; ++i) {
        reg_t *ri = &r[i];
        int si = ri->qs, ei = ri->qe, n_cov = 0, uncov_len = 0;
        for (j = 0; j < k; ++j) {
            reg_t *rp = &r[w[j]];
            int sj = rp->qs, ej = rp->qe;
            if (ej <= si || sj >= ei) continue;
            if (sj < si) sj = si;
            if (ej > ei) ej = ei;
            cov[n_cov++] = (uint64_t)sj << 32 | (uint32_t)ej;
        }
        if (n_cov > 0) {
            int x = si;
            for (int a1 = 1; a1 < n_cov; ++a1) { /* insertion sort: n_cov is tiny */
                uint64_t key = cov[a1];
                int b1 = a1 - 1;
                while (b1 >= 0 && cov[b1] > key) cov[b1 + 1] = cov[b1], --b1;
                cov[b1 + 1] = key;
            }
            for (j = 0; j < n_cov; ++j) {
                if ((int)(cov[j] >> 32) > x) uncov_len += (int)(cov[j] >> 32) - x;
                x = (int32_t)cov[j] > x ? (int32_t)cov[j] : x;
            }
            if (ei > x) uncov_len += ei - x;
            for (j = 0; j < k; ++j) {
                reg_t *rp = &r[w[j]];
                int sj = rp->qs, ej = rp->qe, min, max, ol;
                if (ej <= si || sj >= ei) continue;
                min = ej - sj < ei - si ? ej - sj : ei - si;
                max = ej - sj > ei - si ? ej - sj : ei - si;
                ol = si < sj ? (ei < sj ? 0 : ei < ej ? ei - sj : ej - sj) : (ej < si ? 0 : ej < ei ? ej - si : ei - si);
                if ((float)ol / (float)min - (float)uncov_len / (float)max > MM_MASK_LEVEL) {
                    int cnt_sub = 0, sci = ri->score;
                    ri->parent = rp->parent;
                    rp->subsc = rp->subsc > sci ? rp->subsc : sci;
                    if (ri->cnt >= rp->cnt) cnt_sub = 1;
                    if (rp->has_p && ri->has_p && (rp->rid != ri->rid || rp->rs != ri->rs || rp->re != ri->re || ol != min)) {
                        sci = ri->dp_max;
                        rp->dp_max2 = rp->dp_max2 > sci ? rp->dp_max2 : sci;
                        if (rp->dp_max - ri->dp_max <= sub_diff) cnt_sub = 1;
                    }
                    if (cnt_sub) ++rp->n_sub;
                    break;
                }
            }
            if (j < k) continue;
        }
        w[k++] = i, ri->parent = i, ri->n_sub = 0;
    }
    free(w), free(cov);
}

static void set_mapq(int n, reg_t *r, int rep_len) { /* hit.c: mm_set_mapq2; rep_len = query bases under filtered seeds */
    int64_t sum_sc = 0;
    for (int i = 0; i < n; ++i)
        if (r[i].parent == r[i].id) sum_sc += r[i].score;
    const float uniq_ratio = sum_sc + rep_len > 0 ? (float)sum_sc / (float)(sum_sc + rep_len) : 1.0f;
    for (int i = 0; i < n; ++i) {
        reg_t *g = &r[i];
        if (g->parent != g->id) {
            g->mapq = 0;
            continue;
        }
        int mapq, subsc;
        float pen_s1 = (g->score > 100 ? 1.0f : 0.01f * (float)g->score) * uniq_ratio;
        float pen_cm = g->cnt > 10 ? 1.0f : 0.1f * (float)g->cnt;
        pen_cm = pen_s1 < pen_cm ? pen_s1 : pen_cm;
        subsc = g->subsc > MM_MIN_CHAIN_SC ? g->subsc : MM_MIN_CHAIN_SC;
        float identity = (float)g->mlen / (float)g->blen;
        if (g->dp_max2 > 0 && g->dp_max > 0) {
            float x = (float)g->dp_max2 * (float)subsc / (float)g->dp_max / (float)g->score;
            mapq = (int)(identity * pen_cm * 40.0f * (1.0f - x * x) * logf((float)g->dp_max / (float)MM_A));
            int mapq_alt = (int)(6.02f * identity * identity * (float)(g->dp_max - g->dp_max2) / (float)MM_A + .499f);
            mapq = mapq < mapq_alt ? mapq : mapq_alt;
        } else {
            float x = (float)subsc / (float)g->score;
            mapq = (int)(identity * pen_cm * 40.0f * (1.0f - x) * logf((float)g->dp_max / (float)MM_A));
        }
        mapq -= (int)(4.343f * logf((float)(g->n_sub + 1)) + .499f);
        mapq = mapq > 0 ? mapq : 0;
        g->mapq = (uint8_t)(mapq < 60 ? mapq : 60);
        if (g->dp_max > g->dp_max2 && g->mapq == 0) g->mapq = 1;
    }
}

/* ---- ksw2_extd2 semantics, scalar and unrotated ----------------------------------------------------------------------------
 * H(i,j) = max{H(i-1,j-1)+s, E, F, E2, F2};  E(i+1,j) = max{H(i,j)-q, E(i,j)} - e  (gap consuming the target),
 * F(i,j+1) = max{H(i,j)-q, F(i,j)} - e (gap consuming the query); the same with (q2,e2).  i = target, j = query.
 * Boundary: H(-1,-1) = 0, H(-1,j) = H(j,-1) = -min(q+e(j+1), q2+e2(j+1)).  Band |i-j| <= w.  The diagonal wins ties
 * against the gap states, the first piece against the second, opening against extending; with `right` (KSW_EZ_RIGHT,
 * used on the reversed left extension so that gaps end up left-aligned on the forward strand) the ties go the other
 * way.  Extension mode stops at a z-drop (tested on each anti-diagonal's maximum against the running maximum with
 * the e2 slope) and reports the first global maximum; global mode reports H(tlen-1,qlen-1). */
typedef struct {
    int32_t max, max_t, max_q, score, zdropped;
    uint32_t *cigar; /* BAM encoding: len<<4 | op, op 0=M 1=I 2=D */
    int n_cigar, m_cigar;
} ez_t;

static void cigar_push(ez_t *ez, int op, int len) {
    if (ez->n_cigar > 0 && (int)(ez->cigar[ez->n_cigar - 1] & 0xf) == op) {
        ez->cigar[ez->n_cigar - 1] += (uint32_t)len << 4;
        return;
    }
    if (ez->n_cigar == ez->m_cigar) {
        ez->m_cigar = ez->m_cigar ? ez->m_cigar * 2 : 16;
        ez->cigar = (uint32_t *)realloc(ez->cigar, sizeof(uint32_t) * (size_t)ez->m_cigar);
    }
    ez->cigar[ez->n_cigar++] = (uint32_t)len << 4 | (uint32_t)op;
}

#define NEG_INF (-0x40000000)

static inline int8_t sub_score(int a, int b) { return (a > 3 || b > 3) ? -MM_SC_AMBI : (a == b ? MM_A : -MM_B); }

/* ext_only: extension with z-drop; otherwise global.  rev_cigar: leave the CIGAR in traceback (end-to-start) order. */
static void ksw_extd2(int qlen, const uint8_t *query, int tlen, const uint8_t *target, int w, int zdrop, int ext_only,
                      int right, int rev_cigar, ez_t *ez) {
    ez->max = 0, ez->max_t = ez->max_q = -1, ez->score = NEG_INF, ez->zdropped = 0, ez->n_cigar = 0;
    if (qlen <= 0 || tlen <= 0) return;
    const int64_t n_cells = (int64_t)tlen * qlen;
    /* direction byte per cell: bits 0-2 source of H (0 diag, 1 E, 2 F, 3 E2, 4 F2); bit 3 E extended, 4 F ext, 5 E2 ext, 6 F2 ext */
    uint8_t *dir = (uint8_t *)malloc((size_t)n_cells);
    int32_t *Hrow = (int32_t *)malloc(sizeof(int32_t) * (size_t)(qlen + 1)); /* H(i-1, j) for j = -1..qlen-1 at [j+1] */
    int32_t *Erow = (int32_t *)malloc(sizeof(int32_t) * (size_t)qlen);       /* E(i, j) entering row i */
    int32_t *E2row = (int32_t *)malloc(sizeof(int32_t) * (size_t)qlen);
    int32_t *diagmax = 0, *diagmax_t = 0;
    if (ext_only) {
        diagmax = (int32_t *)malloc(sizeof(int32_t) * (size_t)(tlen + qlen));
        diagmax_t = (int32_t *)malloc(sizeof(int32_t) * (size_t)(tlen + qlen));
        for (int r = 0; r < tlen + qlen; ++r) diagmax[r] = NEG_INF, diagmax_t[r] = -1;
    }
#define GAPC(n) ((MM_Q + MM_E * (n)) < (MM_Q2 + MM_E2 * (n)) ? (MM_Q + MM_E * (n)) : (MM_Q2 + MM_E2 * (n)))
    Hrow[0] = 0;
    for (int j = 0; j < qlen; ++j) {
        Hrow[j + 1] = j < w ? -GAPC(j + 1) : NEG_INF; /* row -1 */
        Erow[j] = NEG_INF, E2row[j] = NEG_INF;
    }
    /* E entering row 0 from the boundary row: open from H(-1,j) */
    for (int j = 0; j < qlen; ++j) {
        if (Hrow[j + 1] > NEG_INF) Erow[j] = Hrow[j + 1] - MM_Q - MM_E, E2row[j] = Hrow[j + 1] - MM_Q2 - MM_E2;
    }
    for (int i = 0; i < tlen; ++i) {
        int jlo = i - w > 0 ? i - w : 0, jhi = i + w < qlen - 1 ? i + w : qlen - 1;
        if (jlo > jhi) break; /* the band has left the matrix */
        int32_t hleft = (i <= w && jlo == 0) ? -GAPC(i + 1) : NEG_INF; /* H(i, jlo-1) */
        int32_t hdiag = jlo == 0 ? (i == 0 ? 0 : (i - 1 < w ? -GAPC(i) : NEG_INF)) : Hrow[jlo]; /* H(i-1, jlo-1) */
        int32_t F = hleft > NEG_INF ? hleft - MM_Q - MM_E : NEG_INF, F2 = hleft > NEG_INF ? hleft - MM_Q2 - MM_E2 : NEG_INF;
        uint8_t *drow = dir + (int64_t)i * qlen;
        const int tc = target[i];
        for (int j = jlo; j <= jhi; ++j) {
            int32_t hup = Hrow[j + 1]; /* H(i-1, j): already folded into Erow */
            int32_t E = Erow[j], E2 = E2row[j];
            if (j > i - 1 + w) E = NEG_INF, E2 = NEG_INF; /* (i-1, j) is outside the band */
            int32_t h = hdiag > NEG_INF ? hdiag + sub_score(tc, query[j]) : NEG_INF;
            uint8_t d = 0;
            if (!right) {
                if (E > h) h = E, d = 1;
                if (F > h) h = F, d = 2;
                if (E2 > h) h = E2, d = 3;
                if (F2 > h) h = F2, d = 4;
            } else {
                if (E >= h) h = E, d = 1;
                if (F >= h) h = F, d = 2;
                if (E2 >= h) h = E2, d = 3;
                if (F2 >= h) h = F2, d = 4;
            }
            /* next gap states */
            int32_t ho = h - MM_Q, ho2 = h - MM_Q2, t1;
            t1 = E;
            if (!right ? (t1 > ho) : (t1 >= ho)) d |= 0x08; else t1 = ho;
            Erow[j] = t1 - MM_E;
            t1 = F;
            if (!right ? (t1 > ho) : (t1 >= ho)) d |= 0x10; else t1 = ho;
            F = t1 - MM_E;
            t1 = E2;
            if (!right ? (t1 > ho2) : (t1 >= ho2)) d |= 0x20; else t1 = ho2;
            E2row[j] = t1 - MM_E2;
            t1 = F2;
            if (!right ? (t1 > ho2) : (t1 >= ho2)) d |= 0x40; else t1 = ho2;
            F2 = t1 - MM_E2;
            drow[j] = d;
            hdiag = hup;
            Hrow[j + 1] = h;
            if (ext_only && h > diagmax[i + j]) diagmax[i + j] = h, diagmax_t[i + j] = i;
        }
        /* cells right of the band in this row are unreachable for the next row */
        if (jhi + 1 < qlen) Hrow[jhi + 2] = NEG_INF;
        Hrow[0] = i < w ? -GAPC(i + 1) : NEG_INF;
        if (jlo > 0) Hrow[jlo] = NEG_INF; /* H(i, jlo-1) outside the band for the next row's diagonal */
    }
    int end_t, end_q;
    if (ext_only) {
        /* anti-diagonal order, exactly as the kernel walks: running maximum, z-drop against it */
        for (int r = 0; r < tlen + qlen - 1; ++r) {
            int32_t H = diagmax[r];
            int t = diagmax_t[r];
            if (t < 0) continue;
            if (H > ez->max) ez->max = H, ez->max_t = t, ez->max_q = r - t;
            else if (t >= ez->max_t && r - t >= ez->max_q) {
                int tl = t - ez->max_t, ql = (r - t) - ez->max_q, l = tl > ql ? tl - ql : ql - tl;
                if (zdrop >= 0 && ez->max - H > zdrop + l * MM_E2) {
                    ez->zdropped = 1;
                    break;
                }
            }
        }
        end_t = ez->max_t, end_q = ez->max_q;
    } else {
        ez->score = (tlen - 1 - w <= qlen - 1) ? Hrow[qlen] : NEG_INF;
        end_t = tlen - 1, end_q = qlen - 1;
        if (ez->score <= NEG_INF / 2) end_t = -1;
    }
    /* NOTE for ext_only: cells beyond a z-drop anti-diagonal were filled but cannot hold the reported maximum, because
     * the scan above stops there. */
    if (end_t >= 0 && end_q >= 0) {
        int i = end_t, j = end_q, state = 0;
        while (i >= 0 && j >= 0) {
            uint8_t d = dir[(int64_t)i * qlen + j];
            if (state == 0) state = d & 7;
            if (state == 0) cigar_push(ez, 0, 1), --i, --j;
            else if (state == 1) { cigar_push(ez, 2, 1); --i; state = (i >= 0 && (dir[(int64_t)i * qlen + j] & 0x08)) ? 1 : 0; }
            else if (state == 2) { cigar_push(ez, 1, 1); --j; state = (j >= 0 && (dir[(int64_t)i * qlen + j] & 0x10)) ? 2 : 0; }
            else if (state == 3) { cigar_push(ez, 2, 1); --i; state = (i >= 0 && (dir[(int64_t)i * qlen + j] & 0x20)) ? 3 : 0; }
            else { cigar_push(ez, 1, 1); --j; state = (j >= 0 && (dir[(int64_t)i * qlen + j] & 0x40)) ? 4 : 0; }
        }
        if (i >= 0) cigar_push(ez, 2, i + 1);
        if (j >= 0) cigar_push(ez, 1, j + 1);
        if (!rev_cigar)
            for (int a1 = 0, b1 = ez->n_cigar - 1; a1 < b1; ++a1, --b1) {
                uint32_t tmp = ez->cigar[a1];
                ez->cigar[a1] = ez->cigar[b1], ez->cigar[b1] = tmp;
            }
    }
    free(dir), free(Hrow), free(Erow), free(E2row);
    if (ext_only) free(diagmax), free(diagmax_t);
}

/* The gap-state bookkeeping above stores in dir[(i,j)] bit 3 whether E LEAVING (i,j) (i.e. E(i+1,j)) extends E(i,j).
 * During traceback a deletion step from (i,j) to (i-1,j) that was in state E must therefore consult the bit of the
 * cell it arrives at: E(i,j) was produced at cell (i-1,j).  The loop above does exactly that. */

/* global fill with an exact z-drop test (the second pass minimap2 runs when mm_test_zdrop fires): anti-diagonal maxima
 * of the GLOBAL matrix against the running maximum; returns through ez->zdropped/max_t/max_q where it stopped */
static void global_zdrop_probe(int qlen, const uint8_t *query, int tlen, const uint8_t *target, int w, int zdrop, ez_t *ez) {
    /* identical recurrence; only H is needed */
    int32_t *Hrow = (int32_t *)malloc(sizeof(int32_t) * (size_t)(qlen + 1));
    int32_t *Erow = (int32_t *)malloc(sizeof(int32_t) * (size_t)qlen), *E2row = (int32_t *)malloc(sizeof(int32_t) * (size_t)qlen);
    int32_t *dm = (int32_t *)malloc(sizeof(int32_t) * (size_t)(tlen + qlen)), *dt = (int32_t *)malloc(sizeof(int32_t) * (size_t)(tlen + qlen));
    for (int r = 0; r < tlen + qlen; ++r) dm[r] = NEG_INF, dt[r] = -1;
    Hrow[0] = 0;
    for (int j = 0; j < qlen; ++j) {
        Hrow[j + 1] = j < w ? -GAPC(j + 1) : NEG_INF;
        Erow[j] = Hrow[j + 1] > NEG_INF ? Hrow[j + 1] - MM_Q - MM_E : NEG_INF;
        E2row[j] = Hrow[j + 1] > NEG_INF ? Hrow[j + 1] - MM_Q2 - MM_E2 : NEG_INF;
    }
    for (int i = 0; i < tlen; ++i) {
        int jlo = i - w > 0 ? i - w : 0, jhi = i + w < qlen - 1 ? i + w : qlen - 1;
        if (jlo > jhi) break;
        int32_t hleft = (i <= w && jlo == 0) ? -GAPC(i + 1) : NEG_INF;
        int32_t hdiag = jlo == 0 ? (i == 0 ? 0 : (i - 1 < w ? -GAPC(i) : NEG_INF)) : Hrow[jlo];
        int32_t F = hleft > NEG_INF ? hleft - MM_Q - MM_E : NEG_INF, F2 = hleft > NEG_INF ? hleft - MM_Q2 - MM_E2 : NEG_INF;
        for (int j = jlo; j <= jhi; ++j) {
            int32_t hup = Hrow[j + 1], E = Erow[j], E2 = E2row[j];
            if (j > i - 1 + w) E = NEG_INF, E2 = NEG_INF;
            int32_t h = hdiag > NEG_INF ? hdiag + sub_score(target[i], query[j]) : NEG_INF;
            if (E > h) h = E;
            if (F > h) h = F;
            if (E2 > h) h = E2;
            if (F2 > h) h = F2;
            int32_t ho = h - MM_Q, ho2 = h - MM_Q2;
            Erow[j] = (E > ho ? E : ho) - MM_E, F = (F > ho ? F : ho) - MM_E;
            E2row[j] = (E2 > ho2 ? E2 : ho2) - MM_E2, F2 = (F2 > ho2 ? F2 : ho2) - MM_E2;
            hdiag = hup, Hrow[j + 1] = h;
            if (h > dm[i + j]) dm[i + j] = h, dt[i + j] = i;
        }
        if (jhi + 1 < qlen) Hrow[jhi + 2] = NEG_INF;
        Hrow[0] = i < w ? -GAPC(i + 1) : NEG_INF;
        if (jlo > 0) Hrow[jlo] = NEG_INF;
    }
    ez->max = 0, ez->max_t = ez->max_q = -1, ez->zdropped = 0;
    for (int r = 0; r < tlen + qlen - 1; ++r) {
        int32_t H = dm[r];
        int t = dt[r];
        if (t < 0) continue;
        if (H > ez->max) ez->max = H, ez->max_t = t, ez->max_q = r - t;
        else if (t >= ez->max_t && r - t >= ez->max_q) {
            int tl = t - ez->max_t, ql = (r - t) - ez->max_q, l = tl > ql ? tl - ql : ql - tl;
            if (ez->max - H > zdrop + l * MM_E2) {
                ez->zdropped = 1;
                break;
            }
        }
    }
    free(Hrow), free(Erow), free(E2row), free(dm), free(dt);
}

/* align.c: mm_test_zdrop without the inversion test */
static int test_zdrop(const uint8_t *qseq, const uint8_t *tseq, int n_cigar, const uint32_t *cigar) {
    int32_t score = 0, max = INT32_MIN, max_i = -1, max_j = -1, i = 0, j = 0, max_zdrop = 0;
    for (int k = 0; k < n_cigar; ++k) {
        uint32_t op = cigar[k] & 0xf, len = cigar[k] >> 4;
        if (op == 0) {
            for (uint32_t l = 0; l < len; ++l) {
                score += sub_score(tseq[i + (int)l], qseq[j + (int)l]);
                if (score < max) {
                    int li = i + (int)l - max_i, lj = j + (int)l - max_j, diff = li > lj ? li - lj : lj - li;
                    int z = max - score - diff * MM_E;
                    if (z > max_zdrop) max_zdrop = z;
                } else
                    max = score, max_i = i + (int)l, max_j = j + (int)l;
            }
            i += (int)len, j += (int)len;
        } else {
            score -= MM_Q + MM_E * (int)len;
            if (op == 1) j += (int)len;
            else i += (int)len;
            if (score < max) {
                int li = i - max_i, lj = j - max_j, diff = li > lj ? li - lj : lj - li;
                int z = max - score - diff * MM_E;
                if (z > max_zdrop) max_zdrop = z;
            } else
                max = score, max_i = i, max_j = j;
        }
    }
    return max_zdrop > MM_ZDROP;
}

typedef struct {
    uint32_t *c;
    int n, m;
} cig_t;
static void cig_append(cig_t *g, int n_cigar, const uint32_t *cigar) {
    for (int k = 0; k < n_cigar; ++k) {
        if (g->n > 0 && (g->c[g->n - 1] & 0xf) == (cigar[k] & 0xf)) {
            g->c[g->n - 1] += cigar[k] >> 4 << 4;
            continue;
        }
        if (g->n == g->m) {
            g->m = g->m ? g->m * 2 : 32;
            g->c = (uint32_t *)realloc(g->c, sizeof(uint32_t) * (size_t)g->m);
        }
        g->c[g->n++] = cigar[k];
    }
}

/* ---- align.c: what happens to a chain before it is aligned ------------------------------------------------------------------- */
static void cal_fuzzy_len(reg_t *r, const mm128 *a) { /* hit.c: mm_cal_fuzzy_len -- matching / aligned bases the anchors imply */
    r->fuzzy_mlen = 0;
    if (r->cnt <= 0) return;
    int32_t mlen = (int32_t)(a[r->as].y >> 32 & 0xff);
    for (int64_t i = r->as + 1; i < r->as + r->cnt; ++i) {
        int32_t span = (int32_t)(a[i].y >> 32 & 0xff);
        int32_t tl = (int32_t)a[i].x - (int32_t)a[i - 1].x, ql = (int32_t)a[i].y - (int32_t)a[i - 1].y;
        mlen += tl > span && ql > span ? span : tl < ql ? tl : ql;
    }
    r->fuzzy_mlen = mlen;
}

/* mm_fix_bad_ends: walking in from either end of the chain, an anchor that follows a gap longer than half the length
 * covered so far becomes the new end; the walk stops once enough of the chain has been seen */
static void fix_bad_ends(const reg_t *r, const mm128 *a, int bw, int min_match, int64_t *as, int32_t *cnt) {
    int64_t i;
    int32_t l, m;
    *as = r->as, *cnt = r->cnt;
    if (r->cnt < 3) return;
    m = l = (int32_t)(a[r->as].y >> 32 & 0xff);
    for (i = r->as + 1; i < r->as + r->cnt - 1; ++i) {
        int32_t lq, lr, min, max, q_span = (int32_t)(a[i].y >> 32 & 0xff);
        if (a[i].y & MM_SEED_LONG_JOIN) break;
        lr = (int32_t)a[i].x - (int32_t)a[i - 1].x;
        lq = (int32_t)a[i].y - (int32_t)a[i - 1].y;
        min = lr < lq ? lr : lq, max = lr > lq ? lr : lq;
        if (max - min > l >> 1) *as = i;
        l += min;
        m += min < q_span ? min : q_span;
        if (l >= bw << 1 || (m >= min_match && m >= bw) || m >= r->fuzzy_mlen >> 1) break;
    }
    *cnt = (int32_t)(r->as + r->cnt - *as);
    m = l = (int32_t)(a[r->as + r->cnt - 1].y >> 32 & 0xff);
    for (i = r->as + r->cnt - 2; i > *as; --i) {
        int32_t lq, lr, min, max, q_span = (int32_t)(a[i + 1].y >> 32 & 0xff);
        if (a[i + 1].y & MM_SEED_LONG_JOIN) break;
        lr = (int32_t)a[i + 1].x - (int32_t)a[i].x;
        lq = (int32_t)a[i + 1].y - (int32_t)a[i].y;
        min = lr < lq ? lr : lq, max = lr > lq ? lr : lq;
        if (max - min > l >> 1) *cnt = (int32_t)(i + 1 - *as);
        l += min;
        m += min < q_span ? min : q_span;
        if (l >= bw << 1 || (m >= min_match && m >= bw) || m >= r->fuzzy_mlen >> 1) break;
    }
}

static int *collect_long_gaps(int64_t as1, int32_t cnt1, const mm128 *a, int min_gap, int *n_) {
    int i, n, *K;
    *n_ = 0;
    for (i = 1, n = 0; i < cnt1; ++i) {
        int gap = ((int32_t)a[as1 + i].y - (int32_t)a[as1 + i - 1].y) - ((int32_t)a[as1 + i].x - (int32_t)a[as1 + i - 1].x);
        if (gap < -min_gap || gap > min_gap) ++n;
    }
    if (n <= 1) return 0;
    K = (int *)malloc(sizeof(int) * (size_t)n);
    for (i = 1, n = 0; i < cnt1; ++i) {
        int gap = ((int32_t)a[as1 + i].y - (int32_t)a[as1 + i - 1].y) - ((int32_t)a[as1 + i].x - (int32_t)a[as1 + i - 1].x);
        if (gap < -min_gap || gap > min_gap) K[n++] = i;
    }
    *n_ = n;
    return K;
}

/* mm_filter_bad_seeds: between two long gaps whose insertions and deletions mostly cancel, the anchors are not to be trusted */
static void filter_bad_seeds(int64_t as1, int32_t cnt1, mm128 *a, int min_gap, int diff_thres, int max_ext_len, int max_ext_cnt) {
    int max_st, max_en, n, i, k, max, *K;
    K = collect_long_gaps(as1, cnt1, a, min_gap, &n);
    if (K == 0) return;
    max = 0, max_st = max_en = -1;
    for (k = 0;; ++k) {
        int gap, l, n_ins = 0, n_del = 0, qs, rs, max_diff = 0, max_diff_l = -1;
        if (k == n || k >= max_en) {
            if (max_en > 0)
                for (i = K[max_st]; i < K[max_en]; ++i) a[as1 + i].y |= MM_SEED_IGNORE;
            max = 0, max_st = max_en = -1;
            if (k == n) break;
        }
        i = K[k];
        gap = ((int32_t)a[as1 + i].y - (int32_t)a[as1 + i - 1].y) - (int32_t)(a[as1 + i].x - a[as1 + i - 1].x);
        if (gap > 0) n_ins += gap;
        else n_del += -gap;
        qs = (int32_t)a[as1 + i - 1].y;
        rs = (int32_t)a[as1 + i - 1].x;
        for (l = k + 1; l < n && l <= k + max_ext_cnt; ++l) {
            int j = K[l], diff;
            if ((int32_t)a[as1 + j].y - qs > max_ext_len || (int32_t)a[as1 + j].x - rs > max_ext_len) break;
            gap = ((int32_t)a[as1 + j].y - (int32_t)a[as1 + j - 1].y) - (int32_t)(a[as1 + j].x - a[as1 + j - 1].x);
            if (gap > 0) n_ins += gap;
            else n_del += -gap;
            diff = n_ins + n_del - abs(n_ins - n_del);
            if (max_diff < diff) max_diff = diff, max_diff_l = l;
        }
        if (max_diff > diff_thres && max_diff > max) max = max_diff, max_st = k, max_en = max_diff_l;
    }
    free(K);
}

/* mm_filter_bad_seeds_alt: long gaps with less matching sequence between them than they are long become one long join */
static void filter_bad_seeds_alt(int64_t as1, int32_t cnt1, mm128 *a, int min_gap, int max_ext) {
    int n, k, *K;
    K = collect_long_gaps(as1, cnt1, a, min_gap, &n);
    if (K == 0) return;
    for (k = 0; k < n;) {
        int i = K[k], l;
        int gap1 = ((int32_t)a[as1 + i].y - (int32_t)a[as1 + i - 1].y) - ((int32_t)a[as1 + i].x - (int32_t)a[as1 + i - 1].x);
        int re1 = (int32_t)a[as1 + i].x, qe1 = (int32_t)a[as1 + i].y;
        gap1 = gap1 > 0 ? gap1 : -gap1;
        for (l = k + 1; l < n; ++l) {
            int j = K[l], gap2, q_span_pre, rs2, qs2, m;
            if ((int32_t)a[as1 + j].y - qe1 > max_ext || (int32_t)a[as1 + j].x - re1 > max_ext) break;
            gap2 = ((int32_t)a[as1 + j].y - (int32_t)a[as1 + j - 1].y) - (int32_t)(a[as1 + j].x - a[as1 + j - 1].x);
            q_span_pre = (int)(a[as1 + j - 1].y >> 32 & 0xff);
            rs2 = (int32_t)a[as1 + j - 1].x + q_span_pre;
            qs2 = (int32_t)a[as1 + j - 1].y + q_span_pre;
            m = rs2 - re1 < qs2 - qe1 ? rs2 - re1 : qs2 - qe1;
            gap2 = gap2 > 0 ? gap2 : -gap2;
            if (m > gap1 + gap2) break;
            re1 = (int32_t)a[as1 + j].x, qe1 = (int32_t)a[as1 + j].y;
            gap1 = gap2;
        }
        if (l > k + 1) {
            int j, end = K[l - 1];
            for (j = K[k]; j < end; ++j) a[as1 + j].y |= MM_SEED_IGNORE;
            a[as1 + end].y |= MM_SEED_LONG_JOIN;
        }
        k = l;
    }
    free(K);
}

/* align.c: mm_align1 for one region (anchors a[as .. as+cnt) of the compacted array, all chains of the query).
 * qcode[0] = gene, qcode[1] = its reverse complement.  On a z-drop inside a global fill the region is truncated and, if
 * at least min_cnt anchors remain, *r2 receives the rest (r2->cnt > 0). */
static void align1(const mm2_index *mi, int qlen, uint8_t *const qcode[2], reg_t *r, reg_t *r2, int64_t n_a, mm128 *a,
                   ez_t *ez) {
    const int bw = (int)(MM_BW * 1.5 + 1.);
    const int rid = r->rid, rev = r->rev;
    const uint8_t *tcode = mi->code[rid];
    const int32_t tlen = mi->len[rid];
    int64_t as1 = r->as;
    int32_t cnt1 = r->cnt;
    int32_t rs, re, qs, qe, rs0, re0, qs0, qe0, rs1, re1, qs1, qe1, l, i;
    cig_t cg = {0, 0, 0};
    r2->cnt = 0;
    if (cnt1 == 0) return;
    if (g_end_flt) { /* (the default: MM_F_NO_END_FLT is off) */
        fix_bad_ends(r, a, MM_BW, MM_MIN_CHAIN_SC * 2, &as1, &cnt1);
        filter_bad_seeds(as1, cnt1, a, 10, 40, MM_MAX_GAP >> 1, 10);
        filter_bad_seeds_alt(as1, cnt1, a, 30, MM_MAX_GAP >> 1);
        /* mm_adjust_minier without homopolymer compression: fills begin and end in the middle of the anchor's k-mer */
        rs = (int32_t)a[as1].x - (MM_K >> 1), qs = (int32_t)a[as1].y - (MM_K >> 1);
        re = (int32_t)a[as1 + cnt1 - 1].x - (MM_K >> 1), qe = (int32_t)a[as1 + cnt1 - 1].y - (MM_K >> 1);
    } else {
        rs = (int32_t)a[as1].x + 1 - (int32_t)(a[as1].y >> 32 & 0xff);
        qs = (int32_t)a[as1].y + 1 - (int32_t)(a[as1].y >> 32 & 0xff);
        re = (int32_t)a[as1 + cnt1 - 1].x + 1;
        qe = (int32_t)a[as1 + cnt1 - 1].y + 1;
    }
    /* extension limits: from the UNCUT chain's first and last anchor outwards */
    rs0 = (int32_t)a[r->as].x + 1 - (int32_t)(a[r->as].y >> 32 & 0xff);
    qs0 = (int32_t)a[r->as].y + 1 - (int32_t)(a[r->as].y >> 32 & 0xff);
    if (rs0 < 0) rs0 = 0;
    rs1 = qs1 = 0;
    for (int64_t ii = r->as - 1, cntl = 0; ii >= 0 && a[ii].x >> 32 == a[r->as].x >> 32; --ii) {
        int32_t x = (int32_t)a[ii].x + 1 - (int32_t)(a[ii].y >> 32 & 0xff), y = (int32_t)a[ii].y + 1 - (int32_t)(a[ii].y >> 32 & 0xff);
        if (x < rs0 && y < qs0) {
            if (++cntl > MM_MIN_CNT) {
                l = rs0 - x > qs0 - y ? rs0 - x : qs0 - y;
                rs1 = rs0 - l, qs1 = qs0 - l;
                if (rs1 < 0) rs1 = 0;
                break;
            }
        }
    }
    if (qs > 0 && rs > 0) {
        l = qs < MM_MAX_GAP ? qs : MM_MAX_GAP;
        qs1 = qs1 > qs - l ? qs1 : qs - l;
        qs0 = qs0 < qs1 ? qs0 : qs1; /* at least include qs0 */
        l += l * MM_A > MM_Q ? (l * MM_A - MM_Q) / MM_E : 0;
        l = l < MM_MAX_GAP ? l : MM_MAX_GAP;
        l = l < rs ? l : rs;
        rs1 = rs1 > rs - l ? rs1 : rs - l;
        rs0 = rs0 < rs1 ? rs0 : rs1;
        rs0 = rs0 < rs ? rs0 : rs;
    } else
        rs0 = rs, qs0 = qs;
    re0 = (int32_t)a[r->as + r->cnt - 1].x + 1, qe0 = (int32_t)a[r->as + r->cnt - 1].y + 1;
    re1 = tlen, qe1 = qlen;
    for (int64_t ii = r->as + r->cnt, cntl = 0; ii < n_a && a[ii].x >> 32 == a[r->as].x >> 32; ++ii) {
        int32_t x = (int32_t)a[ii].x + 1, y = (int32_t)a[ii].y + 1;
        if (x > re0 && y > qe0) {
            if (++cntl > MM_MIN_CNT) {
                l = x - re0 > y - qe0 ? x - re0 : y - qe0;
                re1 = re0 + l, qe1 = qe0 + l;
                break;
            }
        }
    }
    if (qe < qlen && re < tlen) {
        l = qlen - qe < MM_MAX_GAP ? qlen - qe : MM_MAX_GAP;
        qe1 = qe1 < qe + l ? qe1 : qe + l;
        qe0 = qe0 > qe1 ? qe0 : qe1; /* at least include qe0 */
        l += l * MM_A > MM_Q ? (l * MM_A - MM_Q) / MM_E : 0;
        l = l < MM_MAX_GAP ? l : MM_MAX_GAP;
        l = l < tlen - re ? l : tlen - re;
        re1 = re1 < re + l ? re1 : re + l;
        re0 = re0 > re1 ? re0 : re1;
    } else
        re0 = re, qe0 = qe;
    if (qe0 > qlen) qe0 = qlen;
    if (re0 > tlen) re0 = tlen;

    const uint8_t *qseq0 = qcode[rev];
    uint8_t *tbuf = (uint8_t *)malloc((size_t)(re0 - rs0 + 1)), *qbuf = (uint8_t *)malloc((size_t)qlen + 1);
    int32_t dp_score = 0;
    rs1 = rs, qs1 = qs, re1 = re, qe1 = qe; /* updated below */
    if (qs > 0 && rs > 0 && qs > qs0 && rs > rs0) { /* left extension on reversed sequences */
        int ql = qs - qs0, tl = rs - rs0;
        for (i = 0; i < ql; ++i) qbuf[i] = qseq0[qs - 1 - i];
        for (i = 0; i < tl; ++i) tbuf[i] = tcode[rs - 1 - i];
        ksw_extd2(ql, qbuf, tl, tbuf, bw, MM_ZDROP, 1, 1, 1, ez); /* EXTZ_ONLY | RIGHT | REV_CIGAR */
        if (ez->n_cigar > 0) cig_append(&cg, ez->n_cigar, ez->cigar), dp_score += ez->max;
        rs1 = rs - (ez->max_t + 1), qs1 = qs - (ez->max_q + 1);
    }
    re1 = rs, qe1 = qs;
    int dropped = 0;
    for (i = 1; i < cnt1; ++i) { /* gap filling */
        if ((a[as1 + i].y & MM_SEED_IGNORE) && i != cnt1 - 1) continue;
        if (g_end_flt) re = (int32_t)a[as1 + i].x - (MM_K >> 1), qe = (int32_t)a[as1 + i].y - (MM_K >> 1);
        else re = (int32_t)a[as1 + i].x + 1, qe = (int32_t)a[as1 + i].y + 1;
        re1 = re, qe1 = qe;
        if (i == cnt1 - 1 || (a[as1 + i].y & MM_SEED_LONG_JOIN) || (qe - qs >= MM_MIN_KSW_LEN && re - rs >= MM_MIN_KSW_LEN)) {
            const uint8_t *qseq = qseq0 + qs, *tseq = tcode + rs;
            const int bw1 = (a[as1 + i].y & MM_SEED_LONG_JOIN) ? (qe - qs > re - rs ? qe - qs : re - rs) : bw;
            ksw_extd2(qe - qs, qseq, re - rs, tseq, bw1, -1, 0, 0, 0, ez);
            if (test_zdrop(qseq, tseq, ez->n_cigar, ez->cigar)) {
                ez_t probe = {0, 0, 0, 0, 0, 0, 0, 0};
                global_zdrop_probe(qe - qs, qseq, re - rs, tseq, bw1, MM_ZDROP, &probe);
                if (probe.zdropped) { /* truncated by the z-drop: keep the path up to the maximum, split the rest off */
                    ez_t part = {0, 0, 0, 0, 0, 0, 0, 0};
                    ksw_extd2(probe.max_q + 1, qseq, probe.max_t + 1, tseq, bw1, -1, 0, 0, 0, &part);
                    if (part.n_cigar > 0) cig_append(&cg, part.n_cigar, part.cigar);
                    free(part.cigar);
                    int32_t j;
                    for (j = i - 1; j >= 0; --j)
                        if ((int32_t)a[as1 + j].x <= rs + probe.max_t) break;
                    dropped = 1;
                    if (j < 0) j = 0;
                    dp_score += probe.max;
                    re1 = rs + (probe.max_t + 1), qe1 = qs + (probe.max_q + 1);
                    if (cnt1 - (j + 1) >= MM_MIN_CNT) { /* mm_split_reg at anchor as1 + j + 1, counted from the uncut chain's start */
                        const int32_t n_first = (int32_t)(as1 + j + 1 - r->as);
                        *r2 = *r;
                        r2->as = r->as + n_first, r2->cnt = r->cnt - n_first;
                        r2->score = (int32_t)((double)r->score * ((float)r2->cnt / (float)r->cnt) + .499);
                        r->cnt = n_first, r->score -= r2->score;
                        r2->has_p = 0, r2->parent = -1;
                        cal_fuzzy_len(r2, a);
                    }
                    break;
                }
            }
            if (ez->n_cigar > 0) cig_append(&cg, ez->n_cigar, ez->cigar);
            dp_score += ez->score;
            rs = re, qs = qe;
        }
    }
    if (!dropped && qe < qe0 && re < re0) { /* right extension */
        ksw_extd2(qe0 - qe, qseq0 + qe, re0 - re, tcode + re, bw, MM_ZDROP, 1, 0, 0, ez);
        if (ez->n_cigar > 0) cig_append(&cg, ez->n_cigar, ez->cigar), dp_score += ez->max;
        re1 = re + (ez->max_t + 1), qe1 = qe + (ez->max_q + 1);
    }
    /* mm_fix_cigar: leading/trailing gaps go (the coordinates shrink); then mm_update_extra */
    int k0 = 0, k1 = cg.n;
    while (k0 < k1 && (cg.c[k0] & 0xf) != 0) {
        if ((cg.c[k0] & 0xf) == 1) qs1 += (int32_t)(cg.c[k0] >> 4);
        else rs1 += (int32_t)(cg.c[k0] >> 4);
        ++k0;
    }
    while (k1 > k0 && (cg.c[k1 - 1] & 0xf) != 0) {
        if ((cg.c[k1 - 1] & 0xf) == 1) qe1 -= (int32_t)(cg.c[k1 - 1] >> 4);
        else re1 -= (int32_t)(cg.c[k1 - 1] >> 4);
        --k1;
    }
    r->has_p = 0;
    if (k1 > k0) {
        double s = 0.0, max = 0.0;
        int32_t toff = rs1, qoff = qs1, blen = 0, mlen = 0;
        for (int k = k0; k < k1; ++k) {
            uint32_t op = cg.c[k] & 0xf, len = cg.c[k] >> 4;
            if (op == 0) {
                int n_ambi = 0, n_diff = 0;
                for (uint32_t q = 0; q < len; ++q) {
                    int cq = qseq0[qoff + (int)q], ct = tcode[toff + (int)q];
                    if (ct > 3 || cq > 3) ++n_ambi;
                    else if (ct != cq) ++n_diff;
                    s += sub_score(ct, cq);
                    if (s < 0) s = 0;
                    else max = max > s ? max : s;
                }
                blen += (int32_t)len - n_ambi, mlen += (int32_t)len - (n_ambi + n_diff);
                toff += (int32_t)len, qoff += (int32_t)len;
            } else {
                int n_ambi = 0;
                for (uint32_t q = 0; q < len; ++q)
                    if ((op == 1 ? qseq0[qoff + (int)q] : tcode[toff + (int)q]) > 3) ++n_ambi;
                blen += (int32_t)len - n_ambi;
                s -= MM_Q + (double)MM_E * mg_log2(1.0f + (float)len);
                if (s < 0) s = 0;
                if (op == 1) qoff += (int32_t)len;
                else toff += (int32_t)len;
            }
        }
        r->has_p = 1, r->blen = blen, r->mlen = mlen, r->dp_score = dp_score, r->dp_max = (int32_t)(max + .499), r->dp_max2 = 0;
        r->rs = rs1, r->re = re1;
        if (rev) r->qs = qlen - qe1, r->qe = qlen - qs1;
        else r->qs = qs1, r->qe = qe1;
    }
    free(cg.c), free(tbuf), free(qbuf);
}

static int cmp_reg_chain(const void *pa, const void *pb) { /* mm_gen_regs: chain score descending, stable */
    const reg_t *a = (const reg_t *)pa, *b = (const reg_t *)pb;
    if (a->score != b->score) return a->score > b->score ? -1 : 1;
    return a->order_key < b->order_key ? -1 : a->order_key > b->order_key;
}
static int cmp_reg_dp(const void *pa, const void *pb) { /* mm_hit_sort: dp_max descending, stable */
    const reg_t *a = (const reg_t *)pa, *b = (const reg_t *)pb;
    if (a->dp_max != b->dp_max) return a->dp_max > b->dp_max ? -1 : 1;
    return a->order_key < b->order_key ? -1 : a->order_key > b->order_key;
}

/* map one gene; hits appended to out (at most cap); returns the number of hits of this gene (even beyond cap) */
static int64_t map_one(const mm2_index *mi, int gene, const uint8_t *gseq, int qlen, mm2_hit *out, int64_t cap, int64_t n_out,
                       int score_kind) {
    const uint8_t *t4 = nt4();
    if (qlen <= 0) return 0;
    uint8_t *qc[2];
    qc[0] = (uint8_t *)malloc((size_t)qlen), qc[1] = (uint8_t *)malloc((size_t)qlen);
    for (int i = 0; i < qlen; ++i) {
        qc[0][i] = t4[gseq[i]];
        qc[1][qlen - 1 - i] = qc[0][i] < 4 ? (uint8_t)(3 - qc[0][i]) : 4;
    }
    vec128 mv = {0, 0, 0}, av = {0, 0, 0};
    sketch(qc[0], qlen, 0, &mv);
    /* mm_seed_collect_all: the query minimizers that occur in the index, in query order; mm_seed_select (or the plain cut);
     * rep_len = query bases under the seeds that were dropped */
    int32_t n_m0 = 0, rep_len = 0;
    int64_t *m_lo = (int64_t *)malloc(sizeof(int64_t) * (size_t)(mv.n + 1)), *m_hi = (int64_t *)malloc(sizeof(int64_t) * (size_t)(mv.n + 1));
    int32_t *m_ix = (int32_t *)malloc(sizeof(int32_t) * (size_t)(mv.n + 1));
    uint8_t *m_flt = (uint8_t *)calloc((size_t)mv.n + 1, 1);
    for (int64_t m = 0; m < mv.n; ++m) {
        int64_t lo, hi;
        idx_get(mi, mv.a[m].x >> 8, &lo, &hi);
        if (hi - lo == 0) continue;
        m_lo[n_m0] = lo, m_hi[n_m0] = hi, m_ix[n_m0] = (int32_t)m, ++n_m0;
    }
    if (g_seed_select) {
        int n_high = 0;
        for (int i = 0; i < n_m0; ++i) n_high += m_hi[i] - m_lo[i] > mi->mid_occ;
        if (n_m0 > 1 && n_high > 0) {
            for (int i = 0, last0 = -1; i <= n_m0; ++i) {
                if (i == n_m0 || m_hi[i] - m_lo[i] <= mi->mid_occ) {
                    if (i - last0 > 1) { /* a streak of high-occurrence seeds: (last0, i) */
                        const int32_t ps = last0 < 0 ? 0 : (int32_t)((uint32_t)mv.a[m_ix[last0]].y >> 1);
                        const int32_t pe = i == n_m0 ? qlen : (int32_t)((uint32_t)mv.a[m_ix[i]].y >> 1);
                        const int st = last0 + 1, en = i;
                        int max_high_occ = (int)((double)(pe - ps) / MM_OCC_DIST + .499);
                        if (max_high_occ > 0) { /* the max_high_occ least frequent of the streak are kept */
                            if (max_high_occ > MM_MAX_MAX_HIGH_OCC) max_high_occ = MM_MAX_MAX_HIGH_OCC;
                            uint64_t bsel[MM_MAX_MAX_HIGH_OCC];
                            int k = 0, j;
                            for (j = st; j < en && k < max_high_occ; ++j, ++k) bsel[k] = (uint64_t)(m_hi[j] - m_lo[j]) << 32 | (uint32_t)j;
                            for (; j < en; ++j) { /* (a max-heap in minimap2; here the largest entry is found by a scan) */
                                int top = 0;
                                for (int z = 1; z < k; ++z)
                                    if (bsel[z] > bsel[top]) top = z;
                                if ((uint64_t)(m_hi[j] - m_lo[j]) < bsel[top] >> 32) bsel[top] = (uint64_t)(m_hi[j] - m_lo[j]) << 32 | (uint32_t)j;
                            }
                            for (j = 0; j < k; ++j) m_flt[(uint32_t)bsel[j]] = 1;
                        }
                        for (int j = st; j < en; ++j) m_flt[j] ^= 1;
                        for (int j = st; j < en; ++j)
                            if (m_hi[j] - m_lo[j] > MM_MAX_MAX_OCC) m_flt[j] = 1;
                    }
                    last0 = i;
                }
            }
        } else if (n_m0 == 1 && n_high > 0) { /* (mm_seed_select returns early for a single seed: it stays) */
        }
    } else {
        for (int i = 0; i < n_m0; ++i)
            if (m_hi[i] - m_lo[i] > mi->mid_occ) m_flt[i] = 1;
    }
    {
        int rep_st = 0, rep_en = 0;
        for (int i = 0; i < n_m0; ++i) {
            if (!m_flt[i]) continue;
            const mm128 *qm = &mv.a[m_ix[i]];
            const int en = (int)((uint32_t)qm->y >> 1) + 1, st = en - (int)(qm->x & 0xff);
            if (st > rep_en) rep_len += rep_en - rep_st, rep_st = st, rep_en = en;
            else rep_en = en;
        }
        rep_len += rep_en - rep_st;
    }
    for (int mi_ = 0; mi_ < n_m0; ++mi_) { /* seeds -> anchors */
        if (m_flt[mi_]) continue;
        const int64_t m = m_ix[mi_], lo = m_lo[mi_], hi = m_hi[mi_];
        uint32_t q_pos = (uint32_t)mv.a[m].y, q_span = (uint32_t)(mv.a[m].x & 0xff);
        for (int64_t o = lo; o < hi; ++o) {
            uint64_t ry = mi->mz[o].y;
            int32_t rpos = (int32_t)((uint32_t)ry >> 1);
            mm128 p;
            if ((ry & 1) == (q_pos & 1)) {
                p.x = (ry & 0xffffffff00000000ULL) | (uint32_t)rpos;
                p.y = (uint64_t)q_span << 32 | q_pos >> 1;
            } else {
                p.x = 1ULL << 63 | (ry & 0xffffffff00000000ULL) | (uint32_t)rpos;
                p.y = (uint64_t)q_span << 32 | (uint32_t)(qlen - (int32_t)((q_pos >> 1) + 1 - q_span) - 1);
            }
            push(&av, p);
        }
    }
    int64_t n_hits = 0;
    if (av.n > 0) {
        qsort(av.a, (size_t)av.n, sizeof(mm128), cmp128);
        int n_ch;
        mm128 *b;
        chain_t *ch = chain_dp(av.n, av.a, &n_ch, &b);
        int64_t n_b = 0;
        for (int c = 0; c < n_ch; ++c) n_b += ch[c].cnt;
        int n_regs = n_ch, m_regs = n_ch * 2 + 4;
        reg_t *regs = (reg_t *)calloc((size_t)m_regs, sizeof(reg_t));
        for (int c = 0; c < n_ch; ++c) { /* mm_gen_regs */
            reg_t *g = &regs[c];
            const mm128 *f = &b[ch[c].as], *la = &b[ch[c].as + ch[c].cnt - 1];
            g->cnt = ch[c].cnt, g->as = ch[c].as, g->score = ch[c].score, g->order_key = c;
            g->rev = (uint8_t)(f->x >> 63), g->rid = (int32_t)(f->x << 1 >> 33);
            g->rs = (int32_t)f->x + 1 - (int32_t)(f->y >> 32 & 0xff), g->re = (int32_t)la->x + 1;
            int32_t q0 = (int32_t)f->y + 1 - (int32_t)(f->y >> 32 & 0xff), q1 = (int32_t)la->y + 1;
            if (g->rev) g->qs = qlen - q1, g->qe = qlen - q0;
            else g->qs = q0, g->qe = q1;
            cal_fuzzy_len(g, b);
        }
        qsort(regs, (size_t)n_regs, sizeof(reg_t), cmp_reg_chain);
        set_parent(n_regs, regs, MM_A * 2 + MM_B); /* chain_post; mm_select_sub is a no-op at pri_ratio 0 */
        ez_t ez = {0, 0, 0, 0, 0, 0, 0, 0};
        for (int i = 0; i < n_regs; ++i) { /* mm_align_skeleton */
            reg_t r2;
            align1(mi, qlen, qc, &regs[i], &r2, n_b, b, &ez);
            if (r2.cnt > 0) { /* mm_insert_reg after i */
                if (n_regs + 1 > m_regs) {
                    m_regs *= 2;
                    regs = (reg_t *)realloc(regs, sizeof(reg_t) * (size_t)m_regs);
                }
                memmove(&regs[i + 2], &regs[i + 1], sizeof(reg_t) * (size_t)(n_regs - i - 1));
                regs[i + 1] = r2;
                ++n_regs;
            }
        }
        free(ez.cigar);
        int k = 0;
        for (int i = 0; i < n_regs; ++i) { /* mm_filter_regs */
            reg_t *g = &regs[i];
            if (g->cnt < MM_MIN_CNT || !g->has_p) continue;
            if (g->mlen < MM_MIN_CHAIN_SC || g->dp_max < MM_MIN_DP_MAX) continue;
            regs[k] = *g, regs[k].order_key = k;
            ++k;
        }
        n_regs = k;
        qsort(regs, (size_t)n_regs, sizeof(reg_t), cmp_reg_dp); /* mm_hit_sort */
        set_parent(n_regs, regs, MM_A * 2 + MM_B);
        set_mapq(n_regs, regs, rep_len);
        for (int i = 0; i < n_regs; ++i) {
            const reg_t *g = &regs[i];
            if (n_out + n_hits < cap) {
                mm2_hit *h = &out[n_out + n_hits];
                h->gene = gene, h->contig = g->rid, h->q_start = g->qs, h->q_end = g->qe, h->t_start = g->rs, h->t_end = g->re;
                h->score = score_kind == 1 ? g->dp_max : g->dp_score;
                h->matches = g->mlen, h->block_len = g->blen, h->strand = g->rev ? -1 : 1, h->mapq = g->mapq;
                h->n_seeds = (uint8_t)(g->cnt > 255 ? 255 : g->cnt), h->pad_ = 0;
            }
            ++n_hits;
        }
        free(regs), free(ch), free(b);
    }
    free(m_lo), free(m_hi), free(m_ix), free(m_flt);
    free(mv.a), free(av.a), free(qc[0]), free(qc[1]);
    return n_hits;
}

/* Maps every gene (ASCII, back to back) against the index, genes in order, each gene's hits in minimap2's output order
 * (dp_max descending).  score_kind 0: the hit's score is the DP alignment score (PAF AS:i); 1: the best segment's
 * score (PAF ms:i).  Returns the total number of hits; only the first `cap` are written. */
MM_API int64_t mm2_map(void *idx, const uint8_t *genes, const int64_t *gene_off, const int32_t *gene_len, int n_genes,
                       mm2_hit *out, int64_t cap, int score_kind) {
    const mm2_index *mi = (const mm2_index *)idx;
    int64_t n = 0;
    for (int g = 0; g < n_genes; ++g) n += map_one(mi, g, genes + gene_off[g], gene_len[g], out, cap, n, score_kind);
    return n;
}

/* test hooks -------------------------------------------------------------------------------------------------------------- */
/* end_flt: mm_fix_bad_ends + mm_filter_bad_seeds(_alt) + mm_adjust_minier (minimap2's default; 0 = the round-5 model);
 * seed_select: mm_seed_select (0 = every seed above mid_occ is dropped) */
MM_API void mm2_set_options(int end_flt, int seed_select) { g_end_flt = end_flt, g_seed_select = seed_select; }

MM_API int64_t mm2_sketch(const uint8_t *seq, int len, uint64_t *x, uint64_t *y, int64_t cap) {
    const uint8_t *t = nt4();
    uint8_t *code = (uint8_t *)malloc((size_t)len + 1);
    for (int i = 0; i < len; ++i) code[i] = t[seq[i]];
    vec128 v = {0, 0, 0};
    sketch(code, len, 0, &v);
    for (int64_t i = 0; i < v.n && i < cap; ++i) x[i] = v.a[i].x, y[i] = v.a[i].y;
    int64_t n = v.n;
    free(v.a), free(code);
    return n;
}

/* global or extension alignment of two code strings; returns n_cigar, fills score/max/max_t/max_q/zdropped in res[5] */
MM_API int mm2_ksw(const uint8_t *q, int qlen, const uint8_t *t, int tlen, int w, int zdrop, int ext_only, int right,
                   int32_t *res, uint32_t *cigar, int cap) {
    ez_t ez = {0, 0, 0, 0, 0, 0, 0, 0};
    ksw_extd2(qlen, q, tlen, t, w, zdrop, ext_only, right, 0, &ez);
    res[0] = ez.score, res[1] = ez.max, res[2] = ez.max_t, res[3] = ez.max_q, res[4] = ez.zdropped;
    for (int i = 0; i < ez.n_cigar && i < cap; ++i) cigar[i] = ez.cigar[i];
    int n = ez.n_cigar;
    free(ez.cigar);
    return n;
}
