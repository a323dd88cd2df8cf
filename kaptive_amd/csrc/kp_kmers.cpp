// kp_kmers.cpp -- syncmer-linked randstrobes over proteins and the top-hit seed intersection (host only).
//
// Restates the reference's numba kernels behind kaptive.core.kmers.RandstrobeIndex (src/kaptive/core/kmers.py:997-1155:
// _count_randstrobes_kernel / _populate_randstrobes_kernel; 779-819: _radix_sort_records; 1158-1282:
// _compute_query_offsets / _tally_single_query / _intersect_top_hit_kernel), which feed the seeded protein alignments of
// compare.LocusComparator (src/kaptive/compare.py:343-366): SURVEY.md section 8 row f4.  A locus is a few dozen proteins,
// so this is small host work; the alignments it seeds run on the GPU (kp_protein_align_seeded).
//   * residues map to the reference's 12-letter MMseqs alphabet through a 256-byte table the caller passes;
//   * open syncmers: a k-mer whose minimum s-mer hash (splitmix64 of its base-12 value, first minimum wins) sits at its
//     first or last position; every syncmer but the last w_min is linked to the later syncmer, w_min .. w_max places
//     on, that minimises splitmix64(h1 ^ splitmix64(h2)) (first minimum wins);
//   * records are {hash u64, seq_idx u32, pos1 u32, pos2 u32} packed in 20 bytes, in sequence and position order, or
//     stably sorted by hash (the reference's LSD radix sort is stable);
//   * top hit of a query = the target sequence most of its records' hashes occur in (first maximum), diagonal offset =
//     pos1(query) - pos1(target) of the first such occurrence met.
#include <algorithm>
#include <cstring>
#include <vector>

#include "../../include/kaptive_amd.h"

namespace {

inline uint64_t splitmix64(uint64_t x) {
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}

#pragma pack(push, 1)
struct Record {
    uint64_t hash;
    uint32_t seq_idx, pos1, pos2;
};
#pragma pack(pop)
static_assert(sizeof(Record) == 20, "numpy's packed RANDSTROBE_DTYPE");

void syncmers_of(const uint8_t *seq, int len, const uint8_t *lut, int k, int s, std::vector<uint32_t> &pos,
                 std::vector<uint64_t> &hash) {
    pos.clear();
    hash.clear();
    for (int i = 0; i + k <= len; ++i) {
        uint64_t min_hash = ~0ull;
        int min_idx = -1;
        for (int j = 0; j + s <= k; ++j) {
            uint64_t v = 0;
            for (int c = 0; c < s; ++c) v = v * 12u + lut[seq[i + j + c]];
            const uint64_t h = splitmix64(v);
            if (h < min_hash) { min_hash = h; min_idx = j; }
        }
        if (min_idx == 0 || min_idx == k - s) {
            uint64_t v = 0;
            for (int c = 0; c < k; ++c) v = v * 12u + lut[seq[i + c]];
            pos.push_back((uint32_t)i);
            hash.push_back(splitmix64(v));
        }
    }
}

}  // namespace

extern "C" {

int64_t kp_randstrobes(const uint8_t *seqs, const int32_t *offsets, const int32_t *lengths, int32_t n_seqs,
                       const uint8_t *lut256, int32_t k, int32_t s, int32_t w_min, int32_t w_max, int32_t sort_by_hash,
                       void *out_records, int64_t cap) {
    if (n_seqs < 0 || (n_seqs > 0 && (!seqs || !offsets || !lengths)) || !lut256 || s < 1 || s >= k || k > 27 || w_min < 0 ||
        w_max < w_min || cap < 0 || (cap > 0 && !out_records))
        return KP_EINVAL;
    std::vector<Record> recs;
    std::vector<uint32_t> pos;
    std::vector<uint64_t> hash;
    for (int32_t idx = 0; idx < n_seqs; ++idx) {
        if (lengths[idx] < k) continue;
        syncmers_of(seqs + offsets[idx], lengths[idx], lut256, k, s, pos, hash);
        const int n = (int)pos.size();
        for (int i = 0; i + w_min < n; ++i) {
            uint64_t best = ~0ull;
            int best_j = -1;
            const int end_j = std::min(n, i + w_max + 1);
            for (int j = i + w_min; j < end_j; ++j) {
                const uint64_t combined = splitmix64(hash[(size_t)i] ^ splitmix64(hash[(size_t)j]));
                if (combined < best) { best = combined; best_j = j; }
            }
            if (best_j != -1) recs.push_back(Record{best, (uint32_t)idx, pos[(size_t)i], pos[(size_t)best_j]});
        }
    }
    if (sort_by_hash)
        std::stable_sort(recs.begin(), recs.end(), [](const Record &a, const Record &b) { return a.hash < b.hash; });
    const int64_t n = (int64_t)recs.size();
    if (n <= cap && n > 0) std::memcpy(out_records, recs.data(), (size_t)n * sizeof(Record));
    return n;
}

int kp_randstrobe_top_hits(const void *query_records, int64_t n_query_records, int32_t n_queries,
                           const void *target_records_sorted, int64_t n_target_records, int32_t n_targets,
                           uint32_t *best_target, uint32_t *best_score, int32_t *diagonal_offset) {
    if (n_queries < 0 || n_targets < 0 || n_query_records < 0 || n_target_records < 0 ||
        (n_queries > 0 && (!best_target || !best_score || !diagonal_offset)))
        return KP_EINVAL;
    const Record *q = static_cast<const Record *>(query_records), *t = static_cast<const Record *>(target_records_sorted);
    for (int32_t i = 0; i < n_queries; ++i) { best_target[i] = 0; best_score[i] = 0; diagonal_offset[i] = 0; }
    std::vector<uint32_t> tally((size_t)n_targets);
    std::vector<int32_t> anchor((size_t)n_targets);
    int64_t at = 0;
    for (int32_t qi = 0; qi < n_queries; ++qi) {  // query records come grouped by sequence, in sequence order
        while (at < n_query_records && (int32_t)q[at].seq_idx < qi) ++at;
        int64_t end = at;
        while (end < n_query_records && (int32_t)q[end].seq_idx == qi) ++end;
        if (at == end) continue;
        std::fill(tally.begin(), tally.end(), 0u);
        std::fill(anchor.begin(), anchor.end(), 0);
        for (int64_t r = at; r < end; ++r) {
            const uint64_t h = q[r].hash;
            int64_t lo = 0, hi = n_target_records;
            while (lo < hi) {
                const int64_t mid = (lo + hi) / 2;
                if (t[mid].hash < h) lo = mid + 1; else hi = mid;
            }
            for (int64_t c = lo; c < n_target_records && t[c].hash == h; ++c) {
                const uint32_t id = t[c].seq_idx;
                if (id >= (uint32_t)n_targets) return KP_EINVAL;
                if (++tally[id] == 1) anchor[id] = (int32_t)q[r].pos1 - (int32_t)t[c].pos1;
            }
        }
        uint32_t score = 0, target = 0;
        for (int32_t x = 0; x < n_targets; ++x)
            if (tally[(size_t)x] > score) { score = tally[(size_t)x]; target = (uint32_t)x; }
        best_target[qi] = target; best_score[qi] = score; diagonal_offset[qi] = n_targets ? anchor[target] : 0;
        at = end;
    }
    return KP_OK;
}

}  // extern "C"
