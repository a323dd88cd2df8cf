// kp_internal.h -- shared declarations of the HIP implementation (not part of the public ABI).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <string>
#include <vector>

#include "../../include/kaptive_amd.h"

// ---- device-side views -------------------------------------------------------------------------------------------
// Seed index over the genes' minimizers (kp_spec.h, v3): open-addressing table keyed by x = kp_hash30(canonical 15-mer);
// slot = {x, first posting}; postings[first] holds the count n, followed by 2 n posting words: n for a contig seed whose
// strand bit is 0, then the same n gene seeds as a contig seed with strand bit 1 meets them (the other strand of the gene).
// A posting word is the anchor key of the seed for target position 0:
// (gs << 46) | ((KP_DIAG_BIAS - qpos) << 16) | qpos ; adding (tpos << 16) yields the anchor key.
// Presence filter: a blocked Bloom filter of 2^KP_FILTER_LOG2 bits in 64-bit blocks, keyed by the seed value x -- itself
// the output of an invertible mixing function (kp_hash30), so its own bits address the filter: block = the LOW
// KP_FILTER_LOG2 - 6 bits of x (seeds are window minima: their high bits crowd towards zero, the low ones stay uniform),
// one bit in each 32-bit half of the block from bits 18-22 and 23-27.  A probe is one 8-byte gather and six vector
// instructions; 2 MB stays resident in every XCD's L2.  With the ~0.68 M distinct seeds of the KpSC K database a block
// holds 2.6 keys: 1.2 % of foreign seeds pass (measured on the seeds of random sequence).
#define KP_FILTER_LOG2 24
__host__ __device__ inline uint32_t kp_filter_block(uint32_t x) { return x & ((1u << (KP_FILTER_LOG2 - 6)) - 1u); }
__host__ __device__ inline uint2 kp_filter_mask2(uint32_t x) {
    uint2 m;
    m.x = 1u << ((x >> 18) & 31u);
    m.y = 1u << ((x >> 23) & 31u);
    return m;
}
__host__ __device__ inline bool kp_filter_test(uint2 got, uint32_t x) {
    return ((got.x >> ((x >> 18) & 31u)) & (got.y >> ((x >> 23) & 31u)) & 1u) != 0u;
}
__host__ __device__ inline uint64_t kp_filter_mask(uint32_t kmer) {
    const uint2 m = kp_filter_mask2(kmer);
    return ((uint64_t)m.y << 32) | m.x;
}
// Second filter, consulted only for what passed the first (a few per cent of the selected k-mers): 2^KP_FILTER2_LOG2 bits
// in 64-bit blocks, own hashes.  It runs when a wave flushes its staged candidates -- dense, every lane busy, many reads
// in flight -- and removes most of the first filter's false positives before they are written: the candidate list
// shrinks to roughly the k-mers that really are in the index, which is what the scan kernel writes to HBM and what
// kp_expand_kernel has to probe the table for.
#define KP_FILTER2_LOG2 23
__host__ __device__ inline uint32_t kp_filter2_block(uint32_t kmer) { return (kmer * 0xC2B2AE35u) >> (32 - (KP_FILTER2_LOG2 - 6)); }
__host__ __device__ inline uint2 kp_filter2_mask2(uint32_t kmer) {
    const uint32_t h = kmer * 0x27D4EB2Fu;
    uint2 m;
    m.x = (1u << (h >> 27)) | (1u << ((h >> 22) & 31u));
    m.y = (1u << ((h >> 17) & 31u)) | (1u << ((h >> 12) & 31u));
    return m;
}
// A candidate is one word: batch-wide base position of the seed's first base (33 bits: a batch holds less than 2^33 bases),
// the seed's strand bit z and its value x (30 bits).
#define KP_CAND_POS_BITS 33
__host__ __device__ inline uint64_t kp_cand_pack(uint64_t pos, uint32_t z, uint32_t x) { return (pos << 31) | ((uint64_t)z << 30) | x; }
// Which seeds the streaming kernel decides and which the edge kernel (kp_spec.h): inside a clean stretch [S, E) of a contig
// (no ambiguous base) a 15-mer that starts at t is the streaming kernel's iff t >= S + KP_W and t + KP_K + KP_W <= E.
__host__ __device__ inline bool kp_seed_is_interior(int64_t t, int64_t S, int64_t E) { return t >= S + KP_W && t + KP_K + KP_W <= E; }

struct KpSeedIndex {
    const uint64_t *filter;    // [2^KP_FILTER_LOG2 / 64] presence filter over the indexed k-mers
    const uint64_t *filter2;   // [2^KP_FILTER2_LOG2 / 64] recheck filter (see above)
    const uint2 *slots;        // [n_slots], key == 0xFFFFFFFF marks an empty slot
    const uint64_t *postings;  // count word + postings, per distinct k-mer
    uint32_t slot_mask;        // n_slots - 1 (power of two)
    uint32_t slot_shift;       // 32 - log2(n_slots): slot = (kmer * 2654435769u) >> slot_shift
};

struct KpGenes {
    const uint32_t *nib;      // 4-bit codes (0..3 ACGT, 4 = N), 8 per word, each gene starts on a word boundary;
                              // forward sequences first, then the reverse complements (same layout)
    const int32_t *word_off;  // [2 * n_genes] first word of gene g (forward) / n_genes + g (reverse complement)
    const int32_t *len;       // [n_genes]
    int32_t n_genes;
    // the same sequences as the fill kernel wants them: the score profile of every row (kp_row_profile, 16 bits), eight
    // rows per uint4 -- entry i belongs to word i of `nib`; rows past a gene's end hold 0 (a row outside the gene)
    const uint4 *prof;
    const uint8_t *has_n;     // [n_genes] the gene holds an N
};

// Score profile of a query row for the fill kernel (kp_sw.hip): five 3-bit fields, field t = score against target code t
// (0..3 ACGT, 4 = N) + 4 under kp_spec.h's scores (match 2, mismatch -4, N -1); 0 = a row outside the gene (-4 everywhere).
__host__ __device__ inline uint32_t kp_row_profile(uint32_t qcode) {
    return qcode < 4u ? ((6u << (3u * qcode)) | (3u << 12)) : (3u | (3u << 3) | (3u << 6) | (3u << 9) | (3u << 12));
}

// Anchors are sorted (and chained) on a compact form of the spec's key -- same fields in the same order, but only as
// many bits per field as this database / batch can set -- so that the radix sort has fewer digits to go through:
//   [gene*2+strand][diagonal : db bits][query position : qb bits]
struct KpKeyBits {
    uint32_t qb, db;  // qb: bits of the longest gene's positions, db: bits of longest assembly + KP_DIAG_BIAS
};
__host__ __device__ inline uint64_t kp_key_pack(uint64_t spec_key, KpKeyBits kb) {
    return ((spec_key >> 46) << (kb.qb + kb.db)) | (((spec_key >> 16) & 0x3FFFFFFFull) << kb.qb) | (spec_key & 0xFFFFull);
}
__host__ __device__ inline uint32_t kp_ckey_qpos(uint64_t k, KpKeyBits kb) { return (uint32_t)(k & ((1ull << kb.qb) - 1ull)); }
__host__ __device__ inline uint32_t kp_ckey_diag(uint64_t k, KpKeyBits kb) {
    return (uint32_t)((k >> kb.qb) & ((1ull << kb.db) - 1ull));
}
__host__ __device__ inline uint32_t kp_ckey_gs(uint64_t k, KpKeyBits kb) { return (uint32_t)(k >> (kb.qb + kb.db)); }
__host__ __device__ inline uint64_t kp_key_unpack(uint64_t k, KpKeyBits kb) {
    return ((uint64_t)kp_ckey_gs(k, kb) << 46) | ((uint64_t)kp_ckey_diag(k, kb) << 16) | kp_ckey_qpos(k, kb);
}

struct KpBatchView {
    const uint32_t *words;          // packed bases of the whole batch
    const int64_t *asm_word_off;    // [n_asm + 1]
    const int32_t *ctg_start;       // per contig, in its assembly's padded space
    const int32_t *ctg_len;
    const int32_t *asm_first_ctg;   // [n_asm + 1]
    const int32_t *n_runs;          // pairs
    const int32_t *asm_first_nrun;  // [n_asm + 1]
    int32_t n_asm;
    int64_t total_words;
};

#define KP_N_CLASSES 4  // band-width classes: 16, 32, 64, 128 diagonals
// A band task (one banded alignment).  `asm_id` and the derived fields are filled by the chaining kernel.
struct KpTask {
    int32_t asm_id;
    int32_t gs;         // gene * 2 + (strand < 0)
    int32_t contig;     // contig index within the assembly
    int32_t lo;         // lowest diagonal of the band (tpos - qpos, assembly coordinates)
    int32_t width;      // 16 / 32 / 64 / 128
    int32_t n_anchors;  // anchors of the cluster's chain (kp_spec.h)
    uint32_t qspan;     // qmin | qmax << 16: query extent of the cluster's anchors (positions fit 16 bits, kp_spec.h)
    int32_t chain_score;
};  // 32 bytes: the fill kernel takes the first half with one 16-byte load
static_assert(sizeof(KpTask) == 32, "KpTask is read as two 16-byte halves");

// Raw result of one task (before the score filter), same meaning as the oracle's kpo_sw rows.
struct KpSwResult {
    int32_t score, q_start, q_end, t_start, t_end, matches, block_len;
};

// Rows of a band task that can touch its contig.  Cell (row r, band index bi) sits on column lo + r + bi, bi in [0, width):
// above r_lo every cell of a row lies before the contig, from r_hi on every cell lies behind it (or the gene has ended).
// Such cells hold H = 0 and gap states of -(open + ext) whatever happened before them (kp_spec.h: cells outside the contig
// read as H = 0, E = F = -inf), so a fill that starts at r_lo from the all-zero state and stops at r_hi computes the same
// values, the same best cell and the same direction bits for every cell a path can visit.  r_lo is a multiple of 8 (a
// word of the packed gene holds eight rows).
// (a piece of a join: the same, within its rows [r0, r1))
__host__ __device__ inline void kp_task_rows(int lo, int width, int cstart, int cend, int qlen, int *r_lo, int *r_hi);
__host__ __device__ inline void kp_piece_rows(int lo, int width, int cstart, int cend, int qlen, int r0, int r1, int *r_lo, int *r_hi) {
    kp_task_rows(lo, width, cstart, cend, qlen, r_lo, r_hi);
    if (*r_lo < r0) *r_lo = r0;
    if (*r_hi > r1) *r_hi = r1;
    if (*r_hi < *r_lo) *r_hi = *r_lo;
}
__host__ __device__ inline void kp_task_rows(int lo, int width, int cstart, int cend, int qlen, int *r_lo, int *r_hi) {
    int a = cstart - lo - (width - 1);
    if (a < 0) a = 0;
    a &= ~7;
    int z = cend - lo;
    if (z > qlen) z = qlen;
    if (z < a) z = a;
    *r_lo = a;
    *r_hi = z;
}

// What the fill kernel leaves per task: the best cell and where the task's direction bits are (16-byte units).
struct KpSwEnd {
    int32_t score;  // 0 when the trace buffer had no room for the task (the host grows it and reruns the pass)
    int32_t er, eb; // row and band index of the best cell; eb also carries KP_SWEND_HAS_N
    uint32_t trace_off;
};
#define KP_SWEND_HAS_N 0x100  // the gene or the task's target window holds an N (matches are then counted base by base)

// ---- kp-align v5: joins (kp_spec.h) ------------------------------------------------------------------------------------------------
// A task reference: band class in the top bits, slot in the class's list below; KP_REF_NONE = a cluster without a band task.
#define KP_TASK_REF(cls, slot) (((uint32_t)(cls) << 28) | (uint32_t)(slot))
#define KP_REF_CLS(ref) ((ref) >> 28)
#define KP_REF_SLOT(ref) ((ref) & 0x0FFFFFFFu)
#define KP_REF_NONE 0xFFFFFFFFu
// A group of clusters of one gene/strand and contig (kp_chain.hip appends them as it meets them): their tasks and where
// their anchors lie in the assembly's sorted list.  Weak clusters (too few anchors / query bases for a band task) are members
// too: the chaining DP of kp_join.hip runs over all the group's anchors.
struct KpGroup {
    int32_t asm_id, n;
    int32_t gs, contig;
    uint32_t total;                     // anchors of all its clusters (the chaining kernels pick their groups by it)
    uint32_t task[KP_JOIN_GROUP_MAX];   // KP_TASK_REF or KP_REF_NONE
    uint32_t first[KP_JOIN_GROUP_MAX];  // first anchor
    uint32_t cnt[KP_JOIN_GROUP_MAX];    // anchors of the cluster (all of them, not the chain's)
};
// A join: a chain of anchors that falls into two or more pieces.  The chaining kernel fills the head, the joined fill the
// per-piece bookkeeping, the walk-back the results (same meaning as the oracle's kpo_join).
struct KpJoin {
    int32_t asm_id, gs, contig, n_pieces, n_anchors, chain_score, width;
    int32_t lo[KP_JOIN_MAX_PIECES];     // lowest diagonal of every piece's band, query order
    int32_t cmask[KP_JOIN_MAX_PIECES];  // bit c: the piece holds an anchor of the group's cluster c
    int32_t r0[KP_JOIN_MAX_PIECES], r1[KP_JOIN_MAX_PIECES];  // rows [r0, r1) of the piece (kp_spec.h; r0 a multiple of 8, r1 clipped to the gene by the fill)
    int32_t n_members;
    int32_t weak_mask;                  // bit k: piece k belongs to a weak end of the chain (kp_weak_ends)
    uint32_t member_task[KP_JOIN_GROUP_MAX];  // the group's clusters: KP_TASK_REF of their band tasks or KP_REF_NONE
    // joined fill: where each piece's direction bytes and its exports towards the next piece are (16-byte units of the
    // trace buffer; 0xFFFFFFFF = the buffer had no room: the host grows it and reruns the pass), its best cell
    uint32_t trace_off[KP_JOIN_MAX_PIECES], export_off[KP_JOIN_MAX_PIECES];
    int32_t end_s[KP_JOIN_MAX_PIECES], end_r[KP_JOIN_MAX_PIECES], end_b[KP_JOIN_MAX_PIECES];
    // walk-back
    int32_t state[KP_JOIN_MAX_PIECES], visited[KP_JOIN_MAX_PIECES];
    int32_t res[KP_JOIN_MAX_PIECES][9];
    int32_t drop_mask;
};

#define KP_HIP_CHECK(ctx, expr)                                                                     \
    do {                                                                                            \
        hipError_t e_ = (expr);                                                                     \
        if (e_ != hipSuccess) return kp_fail((ctx), KP_EHIP, std::string(#expr) + ": " + hipGetErrorString(e_)); \
    } while (0)

struct kp_ctx;
int kp_fail(kp_ctx *ctx, int code, const std::string &msg);

// ---- kernel launchers (one per .hip file) ------------------------------------------------------------------------
// kp_scan.hip: pass 1 streams the packed contigs and records candidate positions (selected k-mers that pass the presence
//   filters) at the front of `cand` (cand_cap words, kp_cand_pack; n_cand[0] counts them), the edge kernel the seeds next
//   to contig ends and N runs at its back (n_cand[1]); n_cand[0] + n_cand[1] > cand_cap = overflow; pass 2 turns
//   candidates into anchor keys.  Each
//   assembly's anchor region of sub_cap * KP_ANCHOR_SUBS keys is cut into KP_ANCHOR_SUBS sub-slices with their own
//   counters (sub_count[a * KP_ANCHOR_SUBS + s] keeps counting past sub_cap = overflow); kp_launch_anchor_compact then
//   packs each assembly's slices into one run.  `after_scan` (optional) is recorded between the two passes.
#define KP_ANCHOR_SUBS 64
void kp_launch_scan(const KpBatchView &b, const KpSeedIndex &idx, uint64_t *cand, unsigned long long *n_cand,
                    uint64_t cand_cap, uint64_t *anchors, uint32_t *sub_count, uint32_t sub_cap, KpKeyBits key_bits,
                    int ablate_mode, int32_t n_ctg_total, hipStream_t stream, hipEvent_t after_scan);
void kp_launch_anchor_compact(const KpBatchView &b, const uint64_t *sliced, const uint32_t *sub_count, uint32_t sub_cap,
                              uint64_t *out, uint32_t *count, uint32_t *need, hipStream_t stream);
// kp_chain.hip: the occurrence cut on the sorted anchors (kp_spec.h, OCCURRENCE CUT): seeds with more anchors in an assembly than
// minimap2's mid_occ of that assembly lose them; the lists are compacted in place and anchor_count updated.  occ_keys / occ_cnts:
// `occ_slots` counting tables of 2^occ_log2_size entries for the assemblies that need their quantile worked out (a gene seed
// beyond the floor of ten); occ_state: 2 * n_asm + occ_slots words of per-assembly state; *occ_demand (zeroed by the caller) ends
// up as how many asked.
size_t kp_occ_state_words(size_t n_asm, uint32_t occ_slots);  // 32-bit words of the occurrence cut's `occ_state`
void kp_launch_occ_cut(const KpBatchView &b, const int32_t *gene_len, uint64_t *sorted_anchors, uint32_t *anchor_count, uint32_t cap,
                       KpKeyBits key_bits, uint32_t *occ_keys, uint32_t *occ_cnts, uint32_t *occ_state, unsigned long long *occ_demand,
                       uint32_t occ_slots, uint32_t occ_log2_size, hipStream_t stream);
// kp_chain.hip: sorted anchors -> band tasks, appended per width class (class c region = tasks[c * cap ..)).
void kp_launch_chain(const KpBatchView &b, const uint64_t *sorted_anchors, const uint32_t *anchor_count, uint32_t cap,
                     KpKeyBits key_bits, KpTask *tasks, uint32_t *task_count /*[KP_N_CLASSES]*/, uint32_t task_cap,
                     KpGroup *groups, uint32_t *group_count, uint32_t group_cap, hipStream_t stream);
// kp_sw.hip: banded Smith-Waterman of every ORDERED task; class c (16/32/64/128 diagonals) has its tasks, order, ends and
// results at c * task_cap and the length of its order at ordered_count[c]; one fill launch covers all four, one traceback launch follows.
// `trace` holds trace_cap_units 16-byte units; *trace_top (zeroed by the caller) ends up as the units the pass needs.
void kp_launch_sw(const KpBatchView &b, const KpGenes &genes, const KpTask *tasks, const uint32_t *ordered_count,
                  uint32_t task_cap, const uint32_t *order, KpSwEnd *ends, void *trace, unsigned long long *trace_top,
                  uint64_t trace_cap_units, KpSwResult *results, bool has_long_genes, hipStream_t stream,
                  hipEvent_t after_fill);
// kp_chain.hip: settles the provisional tasks (chain score and anchor count of every cluster, or rejection: kp_spec.h), then
// builds, per width class, a permutation of the surviving tasks ordered by query length (longest first).  `hist` is
// KP_ORDER_HEAD zeroed words: histogram, cursors and, at KP_ORDER_COUNTS, how many tasks each class's order holds.
// (65 buckets per class: 64 by length for the packed fill kernel, the 65th holds the tasks of genes longer than
// KP_FILL16_MAX_GENE_LEN, which come last in the order and are filled by kp_sw_long_kernel; counts: [KP_N_CLASSES] ordinary,
// then [KP_N_CLASSES] long)
#define KP_ORDER_BUCKETS 65
#define KP_ORDER_COUNTS (2 * KP_N_CLASSES * KP_ORDER_BUCKETS)
#define KP_ORDER_HEAD (KP_ORDER_COUNTS + 2 * KP_N_CLASSES)
void kp_launch_task_order(const KpBatchView &b, const KpGenes &genes, const uint64_t *sorted_anchors, uint32_t cap, KpKeyBits key_bits,
                          KpTask *tasks, const uint32_t *task_count, uint32_t task_cap, KpSwResult *results, uint32_t *hist,
                          uint32_t *order, hipStream_t stream);
// kp_join.hip (kp-align v5): groups -> joins (one list per band class: joins[c * join_cap ..), join_count[c]); the joined fill and
// walk-back of every join (direction bytes and exports come out of the same trace buffer as the band tasks', in multiples of
// 128 bytes); band tasks whose clusters a chain consumes get their flag in `task_drop` (one byte per task slot and class).
void kp_launch_join_chain(const KpBatchView &b, const KpGenes &genes, const uint64_t *sorted_anchors, uint32_t anchor_cap, KpKeyBits kb,
                          const KpTask *tasks, uint32_t task_cap, const KpGroup *groups, const uint32_t *group_count, uint32_t group_cap,
                          KpJoin *joins, uint32_t *join_count, uint32_t join_cap, uint8_t *scratch, hipStream_t stream);
size_t kp_join_chain_scratch_bytes();  // `scratch`: working arrays of the chaining instance for groups beyond 1024 anchors
void kp_launch_join_fill(const KpBatchView &b, const KpGenes &genes, KpJoin *joins, const uint32_t *join_count, uint32_t join_cap,
                         void *trace, unsigned long long *trace_top, uint64_t trace_cap_units, hipStream_t stream);
void kp_launch_join_trace(const KpBatchView &b, const KpGenes &genes, KpJoin *joins, const uint32_t *join_count, uint32_t join_cap,
                          uint32_t task_cap, const void *trace, uint8_t *task_drop, hipStream_t stream);
// kp_prot.hip
// kp_reduce.hip: assembly a's hits with gene in [gene_lo, gene_hi) (one run: hits are sorted by gene) -> out rows, gene
// indices relative to gene_lo; out_n[a] = how many
void kp_launch_hit_split(const kp_hit *hits, const uint32_t *n_hits, uint32_t hit_cap, int32_t gene_lo, int32_t gene_hi,
                         kp_hit *out, uint32_t *out_n, int32_t n_asm, hipStream_t stream);
// kp_reduce.hip: `bytes` (a multiple of 4) of device memory into page-locked host memory, written by a kernel
void kp_launch_read_back(const void *src, void *dst_pinned, size_t bytes, hipStream_t stream);
void kp_launch_pack_rows(const uint32_t *src, size_t src_pitch, uint32_t *dst, size_t dst_pitch, size_t width, int rows,
                         hipStream_t stream);  // kp_reduce.hip: row-wise copy between pitched word matrices
#define KP_PROT_ROWBUF_FIELDS 8  // ints per column of the strip kernel's row buffer (scratch: fields x (longest target + 1))
void kp_launch_protein(const uint8_t *q, const int32_t *q_off, const int32_t *q_len, const uint8_t *t,
                       const int32_t *t_off, const int32_t *t_len, int32_t n, const int32_t *n_dev, const int8_t *blosum,
                       int32_t *out8, int32_t *scratch, size_t scratch_ints_per_block, int n_blocks, hipStream_t stream,
                       hipStream_t aux, hipEvent_t fork, hipEvent_t join,  // aux != null: wide-band kernel runs beside the other
                       const int32_t *seed_off = nullptr, int seed_k = 0);  // seeded mode: one diagonal offset per pair, band k
// kp_bsort.hip: sub-slices -> sorted run per assembly by buckets of the key's gene/strand field (one block per assembly);
// count[a] / need[a] as kp_launch_anchor_compact leaves them.  n_bins = 2 * genes; kp_bsort_fits says whether it can run.
bool kp_bsort_fits(uint32_t n_bins);
void kp_launch_anchor_bsort(const KpBatchView &b, const uint64_t *sliced, const uint32_t *sub_count, uint32_t sub_cap,
                            uint64_t *grouped, uint64_t *out, uint32_t *count, uint32_t *need, uint32_t n_bins,
                            KpKeyBits kb, hipStream_t stream);
// kp_sort.hip: segmented sort of the anchor regions (wraps rocPRIM)
int kp_sort_anchors(kp_ctx *ctx, uint64_t *keys_in, uint64_t *keys_out, const uint32_t *d_count, uint32_t cap,
                    int32_t n_asm, void **temp, size_t *temp_bytes, uint32_t *d_seg_begin, uint32_t *d_seg_end,
                    int end_bit, hipStream_t stream);
