// kp_scan.hip -- seed scan: stream the 2-bit packed contigs once, emit anchors against the resident seed index.
//
// Stands in for the seeding half of rammappy's map_batch (reference call site src/kaptive/serotyping/core.py:154); the
// rule and key layout are those of include/kp_spec.h.  This is the only kernel that touches every base of every
// assembly, so it is the one priced against the HBM-read roofline: algorithmic bytes = total_words * 4 per launch.
//
// Mapping: one lane owns 64 consecutive bases (one 16-byte load, lanes of a wave are contiguous -> 1 KiB per wave
// instruction).  The 65th..80th base needed by k-mers that start near the end of the lane's span come from the next
// lane's first word (DPP/shuffle), so every word is fetched from HBM exactly once.  The context-free seed rule
// (c[p]^c[p+1]^c[p+3]==1) is evaluated for 16 positions at a time with word-wide bit operations; only the selected
// quarter of positions goes on to the blocked-Bloom presence filter (2 MB, stays in each XCD's L2; kp_internal.h).  What passes
// the filter is only recorded (kp_scan_kernel); a second, perfectly balanced kernel (kp_expand_kernel) probes the
// k-mer table (tens of MB, Infinity Cache), validates and expands the postings into anchors.
#include "kp_internal.h"

namespace {

// LDS written by some lanes of a wave, read by others of the same wave (no block barrier: waves loop independently)
__device__ __forceinline__ void wave_lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

__device__ __forceinline__ int upper_bound_i64(const int64_t *a, int n, int64_t v) {  // first i with a[i] > v
    int lo = 0, hi = n;
    while (lo < hi) {
        int mid = (lo + hi) >> 1;
        if (a[mid] <= v) lo = mid + 1; else hi = mid;
    }
    return lo;
}

__device__ __forceinline__ int upper_bound_i32(const int32_t *a, int n, int32_t v) {
    int lo = 0, hi = n;
    while (lo < hi) {
        int mid = (lo + hi) >> 1;
        if (a[mid] <= v) lo = mid + 1; else hi = mid;
    }
    return lo;
}

// ---- pass 1: stream + select + presence filter -> candidate positions --------------------------------------------------
// Every base is read once.  A selected k-mer that passes the filter becomes a candidate: its batch-wide base position
// is staged in the wave's own LDS slice (ballot + popcount, no atomics) and flushed to the global candidate list with
// one atomic per flush, so the streaming kernel has no data-dependent slow path: what a candidate costs later
// (table probe, contig / N validation, posting expansion) is done by kp_expand_kernel with one thread per candidate.
// MODE 0 = product; 1 = no filter reads (stream + select + hash only); 2 = stream only.  Modes 1 and 2 exist for the
// ablation in tools/scan_ablate.py and write a checksum so that the work is not optimised away.
//
// A lane's 64 positions are handled as two halves of 32 (two packed words each): the selected positions of a half are
// one 32-bit mask (word 0's at the even bits, word 1's at the odd bits), and the lane walks exactly its own set bits,
// PROBES at a time, so that a round's filter reads are all in flight before the first is looked at.  (The round-1 kernel
// ran 8 probe slots per word and round whatever the word had selected: three quarters of its instructions were idle.)
//
// Two filter tiers.  A database with few k-mers (O loci: tens of thousands) gets a filter small enough for LDS
// (idx.lds_filter, <= KP_LDS_FILTER_BLOCKS 64-bit blocks): LDSF = true copies it into the block's LDS once and probes it
// there, so the kernel no longer pays one L2 request per selected k-mer (the L2 tier runs at the L2's request rate);
// blocks are 16 waves wide (one per CU: the filter takes most of its LDS) with a small candidate stage per wave.
// Otherwise the 2 MB filter is probed in L2 (LDSF = false, 4-wave blocks, several per CU).
template <bool LDSF> struct ScanShape {
    static constexpr int WAVES = LDSF ? 16 : 4;
    // a round of PROBES positions per lane adds at most 64 * PROBES entries (flush after the round)
    static constexpr int STAGE = LDSF ? 320 : 1024;
    static constexpr int FILTER_BLOCKS = LDSF ? KP_LDS_FILTER_BLOCKS : 1;
};

#ifndef KP_SCAN_PROBES
#define KP_SCAN_PROBES 4
#endif
constexpr int PROBES = KP_SCAN_PROBES;  // filter reads a lane has in flight

template <int MODE, bool LDSF>
__global__ __launch_bounds__(64 * ScanShape<LDSF>::WAVES) void kp_scan_kernel(KpBatchView b, KpSeedIndex idx,
                                                                              uint64_t *__restrict__ cand,
                                                                              unsigned long long *__restrict__ n_cand,
                                                                              uint64_t cand_cap) {
    constexpr int WAVES = ScanShape<LDSF>::WAVES, STAGE_PER_WAVE = ScanShape<LDSF>::STAGE;
    __shared__ uint64_t s_stage[WAVES][STAGE_PER_WAVE];
    __shared__ uint2 s_filter[ScanShape<LDSF>::FILTER_BLOCKS];
    const uint2 *g_filter = reinterpret_cast<const uint2 *>(idx.filter);
    if (LDSF) {
        const uint2 *src = reinterpret_cast<const uint2 *>(idx.lds_filter);
        for (uint32_t i = threadIdx.x; i < idx.lds_filter_blocks; i += blockDim.x) s_filter[i] = src[i];
        __syncthreads();
    }
    uint32_t checksum = 0;
    const int64_t n_units = b.total_words >> 2;  // 16-byte units; every assembly is a whole number of them
    const int64_t n_iter_units = (n_units + 63) & ~(int64_t)63;  // whole waves iterate together (shuffle below)
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    const uint4 *vec = reinterpret_cast<const uint4 *>(b.words);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint64_t *stage = s_stage[wave];
    uint32_t staged = 0;  // wave-uniform
    const unsigned long long below = (1ull << lane) - 1ull;
    const uint2 *g_filter2 = reinterpret_cast<const uint2 *>(idx.filter2);

    // The staged candidates go through the second filter (one read each, all lanes busy), the survivors are packed to the
    // front of the stage and leave with one atomic for the whole flush.
    auto flush = [&]() {
        uint32_t kept = 0;  // wave-uniform
        for (uint32_t i0 = 0; i0 < staged; i0 += 64) {
            const uint32_t i = i0 + lane;
            uint64_t c = 0;
            bool ok = false;
            if (i < staged) {
                c = stage[i];
                const uint32_t kmer = (uint32_t)c & KP_KMER_MASK;
                const uint2 got = g_filter2[kp_filter2_block(kmer)], need = kp_filter2_mask2(kmer);
                ok = (got.x & need.x) == need.x && (got.y & need.y) == need.y;
            }
            const unsigned long long pass = __ballot(ok);
            if (ok) stage[kept + (uint32_t)__builtin_popcountll(pass & below)] = c;  // kept <= i0: never ahead of the reads
            kept += (uint32_t)__builtin_popcountll(pass);
        }
        unsigned long long base = 0;
        if (lane == 0 && kept) base = atomicAdd(n_cand, (unsigned long long)kept);
        base = __shfl(base, 0);
        for (uint32_t i = lane; i < kept; i += 64)
            if (base + i < cand_cap) cand[base + i] = stage[i];
        staged = 0;
    };

    for (int64_t u = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; u < n_iter_units; u += stride) {
        uint4 v = make_uint4(0, 0, 0, 0);
        if (u < n_units) v = vec[u];
        uint32_t next = __shfl_down(v.x, 1);
        if (lane == 63) next = (u + 1 < n_units) ? b.words[(u + 1) << 2] : 0u;
        const uint32_t w[5] = {v.x, v.y, v.z, v.w, next};
        if (MODE == 2) { checksum += v.x ^ v.y ^ v.z ^ v.w ^ next; continue; }
        uint32_t sel[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const uint32_t lo = w[k], hi = w[k + 1];
            // 2-bit lanes: x = c[p] ^ c[p+1] ^ c[p+3] for the 16 positions of this word
            const uint32_t x = lo ^ __builtin_amdgcn_alignbit(hi, lo, 2) ^ __builtin_amdgcn_alignbit(hi, lo, 6);
            sel[k] = x & ~(x >> 1) & 0x55555555u;  // value 01: low bit set, high bit clear
        }
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            const uint32_t w0 = w[2 * half], w1 = w[2 * half + 1], w2 = w[2 * half + 2];
            uint32_t m = sel[2 * half] | (sel[2 * half + 1] << 1);  // bit 2i: word 0 position i; bit 2i + 1: word 1 position i
            const uint64_t half_base = (uint64_t)((u << 2) + 2 * half) << 4;  // batch-wide position of the half's first base
            while (__any(m != 0)) {  // wave-uniform: a lane that ran out of selected positions idles along
                uint32_t kmers[PROBES], pos[PROBES], pass[PROBES];
                uint2 got[PROBES];
#pragma unroll
                for (int j = 0; j < PROBES; ++j) {
                    const bool have = m != 0;
                    const int bit = have ? __builtin_ctz(m) : 0;
                    m &= m - 1;  // no-op once m is 0
                    const bool odd = bit & 1;
                    const uint32_t lo = odd ? w1 : w0, hi = odd ? w2 : w1;
                    const uint32_t kmer = kmers[j] = __builtin_amdgcn_alignbit(hi, lo, bit & 30) & KP_KMER_MASK;
                    pos[j] = (uint32_t)(bit >> 1) + (odd ? 16u : 0u);
                    pass[j] = have ? 1u : 0u;
                    if (MODE == 1) { checksum += have ? kmer * 2654435769u : 0u; continue; }
                    const uint32_t blk = LDSF ? kp_lds_filter_block(kmer, idx.lds_filter_blocks) : kp_filter_block(kmer);
                    got[j] = LDSF ? s_filter[blk] : (have ? g_filter[blk] : make_uint2(0u, 0u));
                }
                if (MODE != 0) continue;
#pragma unroll
                for (int j = 0; j < PROBES; ++j) {
                    const uint2 need = kp_filter_mask2(kmers[j]);
                    const bool hit = pass[j] && (got[j].x & need.x) == need.x && (got[j].y & need.y) == need.y;
                    const unsigned long long ballot = __ballot(hit);
                    if (!ballot) continue;  // ~99 % of selected positions stop at the filter (KpSC K database)
                    if (hit)  // position and k-mer in one word: the expansion pass does not touch the bases again
                        stage[staged + (uint32_t)__builtin_popcountll(ballot & below)] = kp_cand_pack(half_base + pos[j], kmers[j]);
                    staged += (uint32_t)__builtin_popcountll(ballot);
                }
                if (staged > STAGE_PER_WAVE - 64 * PROBES) flush();
            }
        }
    }
    if (MODE == 0 && staged) flush();
    if (MODE != 0 && checksum == 0x9E3779B1u) n_cand[0] = checksum;  // practically never; keeps the work alive
}

// ---- pass 1, dense form (the L2 tier of the filter: every database but the smallest) ---------------------------------------
// Same reads, same rule, same filters, same candidates as kp_scan_kernel<0, false>; what changes is who probes what.  There
// a lane walks the selected positions of its OWN 64 bases, and a wave runs at the pace of its fullest lane: a quarter of
// the positions is selected on average (8 per half of 32), the fullest of 64 lanes holds about 14, so half of the probe
// slots were idle lanes.  Here the wave first COMPACTS its selected positions into a list in LDS (a cheap loop: find the
// bit, store 16 bits), keeps its 257 packed words in LDS as well, and then walks the list 64 entries at a time with every
// lane busy: entry -> position within the wave's 4096 bases -> two words from LDS -> k-mer -> filter block (one L2 read,
// PROBES rounds in flight).  About 30 vector instructions per selected position instead of 60.
// MODE as above (0 product, 1 no filter reads, 2 stream only).
constexpr int DENSE_WAVES = 4;
constexpr int DENSE_LIST = 2048;   // selected positions of a wave's iteration the list holds (mean 1024; see `groups`)
constexpr int DENSE_STAGE = 512;   // candidates staged per wave before a flush

template <int MODE>
__global__ __launch_bounds__(64 * DENSE_WAVES) void kp_scan_dense_kernel(KpBatchView b, KpSeedIndex idx,
                                                                         uint64_t *__restrict__ cand,
                                                                         unsigned long long *__restrict__ n_cand,
                                                                         uint64_t cand_cap) {
    __shared__ uint64_t s_stage[DENSE_WAVES][DENSE_STAGE];
    __shared__ __attribute__((aligned(16))) uint32_t s_words[DENSE_WAVES][260];
    __shared__ uint16_t s_list[DENSE_WAVES][DENSE_LIST];
    const uint2 *g_filter = reinterpret_cast<const uint2 *>(idx.filter);
    const uint2 *g_filter2 = reinterpret_cast<const uint2 *>(idx.filter2);
    uint32_t checksum = 0;
    const int64_t n_units = b.total_words >> 2;
    const int64_t n_iter_units = (n_units + 63) & ~(int64_t)63;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    const uint4 *vec = reinterpret_cast<const uint4 *>(b.words);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint64_t *stage = s_stage[wave];
    uint32_t *words = s_words[wave];
    uint16_t *list = s_list[wave];
    uint32_t staged = 0;  // wave-uniform
    const unsigned long long below = (1ull << lane) - 1ull;

    auto flush = [&]() {  // second filter on what is staged, survivors out with one atomic (as in kp_scan_kernel)
        uint32_t kept = 0;
        for (uint32_t i0 = 0; i0 < staged; i0 += 64) {
            const uint32_t i = i0 + lane;
            uint64_t c = 0;
            bool ok = false;
            if (i < staged) {
                c = stage[i];
                const uint32_t kmer = (uint32_t)c & KP_KMER_MASK;
                const uint2 got = g_filter2[kp_filter2_block(kmer)], need = kp_filter2_mask2(kmer);
                ok = (got.x & need.x) == need.x && (got.y & need.y) == need.y;
            }
            const unsigned long long pass = __ballot(ok);
            if (ok) stage[kept + (uint32_t)__builtin_popcountll(pass & below)] = c;
            kept += (uint32_t)__builtin_popcountll(pass);
        }
        unsigned long long base = 0;
        if (lane == 0 && kept) base = atomicAdd(n_cand, (unsigned long long)kept);
        base = __shfl(base, 0);
        for (uint32_t i = lane; i < kept; i += 64)
            if (base + i < cand_cap) cand[base + i] = stage[i];
        staged = 0;
    };

    for (int64_t u = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; u < n_iter_units; u += stride) {
        uint4 v = make_uint4(0, 0, 0, 0);
        if (u < n_units) v = vec[u];
        uint32_t next = __shfl_down(v.x, 1);
        if (lane == 63) next = (u + 1 < n_units) ? b.words[(u + 1) << 2] : 0u;
        if (MODE == 2) { checksum += v.x ^ v.y ^ v.z ^ v.w ^ next; continue; }
        const uint32_t w[5] = {v.x, v.y, v.z, v.w, next};
        // selected positions of the lane's 64 bases: two masks of 32 (bit 2i: position i of the even word, bit 2i + 1:
        // position i of the odd word of the pair)
        uint32_t m2[2];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            uint32_t sel[2];
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const uint32_t lo = w[2 * h + k], hi = w[2 * h + k + 1];
                const uint32_t x = lo ^ __builtin_amdgcn_alignbit(hi, lo, 2) ^ __builtin_amdgcn_alignbit(hi, lo, 6);
                sel[k] = x & ~(x >> 1) & 0x55555555u;
            }
            m2[h] = sel[0] | (sel[1] << 1);
        }
        *reinterpret_cast<uint4 *>(&words[4 * lane]) = v;
        if (lane == 63) words[256] = next;
        const uint32_t mine = (uint32_t)__builtin_popcount(m2[0]) + (uint32_t)__builtin_popcount(m2[1]);
        uint32_t incl = mine;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t t = __shfl_up(incl, o);
            if (lane >= o) incl += t;
        }
        const uint32_t total = __shfl(incl, 63);
        // the list holds DENSE_LIST entries: an iteration with more selected positions (a low-complexity stretch: the rule
        // can select every position) goes through in two groups of 32 lanes, each at most 32 x 64 = 2048 positions
        const int groups = total > (uint32_t)DENSE_LIST ? 2 : 1;
        const uint32_t before_half = __shfl(incl, 31);  // selected positions of lanes 0..31
        const uint64_t wave_base = (uint64_t)((u - lane) << 2) << 4;  // batch-wide position of the wave's first base
        for (int g = 0; g < groups; ++g) {
            const bool active = groups == 1 || (lane >> 5) == g;
            const uint32_t n_list = groups == 1 ? total : (g == 0 ? before_half : total - before_half);
            uint32_t off = incl - mine - ((groups == 2 && g == 1) ? before_half : 0u);
            wave_lds_sync();  // (the previous walk is done with the list; the words are visible)
            if (active) {
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    uint32_t m = m2[h];
                    while (m) {
                        const int bit = __builtin_ctz(m);
                        m &= m - 1;
                        list[off++] = (uint16_t)(64 * lane + 32 * h + 16 * (bit & 1) + (bit >> 1));
                    }
                }
            }
            wave_lds_sync();
            for (uint32_t e0 = 0; e0 < n_list; e0 += 64 * PROBES) {
                uint32_t kmers[PROBES], pos[PROBES];
                bool have[PROBES];
                uint2 got[PROBES];
#pragma unroll
                for (int j = 0; j < PROBES; ++j) {
                    const uint32_t e = e0 + 64 * j + lane;
                    have[j] = e < n_list;
                    const uint32_t p = have[j] ? list[e] : 0u;
                    const uint32_t lo = words[p >> 4], hi = words[(p >> 4) + 1];
                    kmers[j] = __builtin_amdgcn_alignbit(hi, lo, 2 * (p & 15u)) & KP_KMER_MASK;
                    pos[j] = p;
                    if (MODE == 1) { checksum += have[j] ? kmers[j] * 2654435769u : 0u; continue; }
                    got[j] = have[j] ? g_filter[kp_filter_block(kmers[j])] : make_uint2(0u, 0u);
                }
                if (MODE != 0) continue;
#pragma unroll
                for (int j = 0; j < PROBES; ++j) {
                    const uint2 need = kp_filter_mask2(kmers[j]);
                    const bool hit = have[j] && (got[j].x & need.x) == need.x && (got[j].y & need.y) == need.y;
                    const unsigned long long ballot = __ballot(hit);
                    if (!ballot) continue;
                    if (hit) stage[staged + (uint32_t)__builtin_popcountll(ballot & below)] = kp_cand_pack(wave_base + pos[j], kmers[j]);
                    staged += (uint32_t)__builtin_popcountll(ballot);
                }
                if (staged > DENSE_STAGE - 64 * PROBES) flush();
            }
        }
    }
    if (MODE == 0 && staged) flush();
    if (MODE != 0 && checksum == 0x9E3779B1u) n_cand[0] = checksum;
}

// ---- pass 2: candidates -> anchors -------------------------------------------------------------------------------------
// One thread per candidate: re-read its k-mer, probe the table, validate against contig bounds and N runs, reserve room
// for its postings in its assembly's anchor region.  The postings themselves are then copied by the whole wave as one
// flat list (prefix sums of the counts, every lane finds the owner of its output item by binary search): a shared gene
// family has ~160 postings per seed, a per-thread copy loop would run the wave at the pace of its longest list.  Appends of one assembly are spread over
// KP_ANCHOR_SUBS counters because all hits of an assembly come in one burst (the typed locus) and would otherwise
// serialise on a single atomic word.
__global__ __launch_bounds__(256) void kp_expand_kernel(KpBatchView b, KpSeedIndex idx, const uint64_t *__restrict__ cand,
                                                         const unsigned long long *__restrict__ n_cand, uint64_t cand_cap,
                                                         uint64_t *__restrict__ anchors, uint32_t *__restrict__ sub_count,
                                                         uint32_t sub_cap, KpKeyBits kb) {
    unsigned long long n = *n_cand;
    if (n > cand_cap) n = cand_cap;
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    struct WaveStage {  // survivors of the wave's current 64 candidates, in lane order
        uint32_t start[64], room[64];
        const uint64_t *src[64];
        uint64_t *dst[64];
        uint64_t shift[64];
    };
    __shared__ WaveStage s_stage[4];
    WaveStage &sw = s_stage[threadIdx.x >> 6];
    const int lane = threadIdx.x & 63;
    // whole waves iterate together: the copy phase below needs every lane of the wave
    for (uint64_t i0 = (uint64_t)blockIdx.x * blockDim.x + (threadIdx.x & ~63u); i0 < n; i0 += stride) {
        const uint64_t i = i0 + lane;
        uint32_t first_posting = 0xFFFFFFFFu, cnt = 0, base = 0, t = 0;
        size_t slice = 0;
        if (i < n) {
            const uint64_t pos = cand[i] >> 30;
            const int64_t word = (int64_t)(pos >> 4);
            const uint32_t kmer = (uint32_t)cand[i] & KP_KMER_MASK;
            uint32_t slot = (kmer * 2654435769u) >> idx.slot_shift;
            for (;;) {
                const uint2 e = idx.slots[slot];
                if (e.x == kmer) { first_posting = e.y; break; }
                if (e.x == 0xFFFFFFFFu) break;
                slot = (slot + 1) & idx.slot_mask;
            }
            if (first_posting != 0xFFFFFFFFu) {  // else: filter false positive
                bool ok = false;
                const int a = upper_bound_i64(b.asm_word_off, b.n_asm + 1, word) - 1;
                if (a >= 0 && a < b.n_asm) {
                    t = (uint32_t)(pos - ((uint64_t)b.asm_word_off[a] << 4));
                    const int c0 = b.asm_first_ctg[a], nc = b.asm_first_ctg[a + 1] - c0;
                    const int c = upper_bound_i32(b.ctg_start + c0, nc, (int32_t)t) - 1;
                    // the k-mer must not run past its contig (or sit in padding) ...
                    ok = c >= 0 && (int32_t)t + KP_K <= b.ctg_start[c0 + c] + b.ctg_len[c0 + c];
                    const int r0 = b.asm_first_nrun[a], nr = b.asm_first_nrun[a + 1] - r0;
                    if (ok && nr > 0) {  // ... nor overlap an N run: first run whose end is > t starts before t + K
                        int lo = 0, hi = nr;
                        while (lo < hi) {
                            int mid = (lo + hi) >> 1;
                            if (b.n_runs[2 * (r0 + mid) + 1] <= (int32_t)t) lo = mid + 1; else hi = mid;
                        }
                        if (lo < nr && b.n_runs[2 * (r0 + lo)] < (int32_t)t + KP_K) ok = false;
                    }
                    if (ok) {
                        cnt = (uint32_t)idx.postings[first_posting];
                        slice = (size_t)a * KP_ANCHOR_SUBS + (threadIdx.x & (KP_ANCHOR_SUBS - 1));
                        base = atomicAdd(&sub_count[slice], cnt);
                    }
                }
                if (!ok) cnt = 0;
            }
        }
        // ---- load-balanced copy: output item k of the wave belongs to the survivor whose prefix range holds k ----------
        const unsigned long long live = __ballot(cnt != 0);
        if (!live) continue;
        uint32_t incl = cnt;  // inclusive prefix sum of cnt over the wave
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t v = __shfl_up(incl, o);
            if (lane >= o) incl += v;
        }
        const uint32_t total = __shfl(incl, 63);
        const int rank = __builtin_popcountll(live & ((1ull << lane) - 1ull));
        const int n_live = __builtin_popcountll(live);
        if (cnt) {
            sw.start[rank] = incl - cnt;
            sw.src[rank] = idx.postings + first_posting + 1;
            sw.dst[rank] = anchors + slice * sub_cap + base;
            sw.room[rank] = base < sub_cap ? sub_cap - base : 0u;
            sw.shift[rank] = (uint64_t)t << 16;
        }
        wave_lds_sync();
        for (uint32_t k = lane; k < total; k += 64) {
            int lo = 0, hi = n_live;  // last survivor with start <= k
            while (hi - lo > 1) {
                const int mid = (lo + hi) >> 1;
                if (sw.start[mid] <= k) lo = mid; else hi = mid;
            }
            const uint32_t j = k - sw.start[lo];
            if (j < sw.room[lo]) sw.dst[lo][j] = kp_key_pack(sw.src[lo][j] + sw.shift[lo], kb);  // compact sort key
        }
        wave_lds_sync();
    }
}

// sub-slices of each assembly -> one contiguous run per assembly (input of the sort); count[a] = anchors stored,
// need[a] = the largest demand of any of its sub-slices (overflow if > sub_cap)
__global__ __launch_bounds__(256) void kp_anchor_compact_kernel(const uint64_t *__restrict__ sliced,
                                                                const uint32_t *__restrict__ sub_count, uint32_t sub_cap,
                                                                uint64_t *__restrict__ out, uint32_t *__restrict__ count,
                                                                uint32_t *__restrict__ need) {
    __shared__ uint32_t s_off[KP_ANCHOR_SUBS + 1];
    const int a = blockIdx.x;
    const uint32_t *sc = sub_count + (size_t)a * KP_ANCHOR_SUBS;
    if (threadIdx.x == 0) {
        uint32_t acc = 0, mx = 0;
        for (int k = 0; k < KP_ANCHOR_SUBS; ++k) {
            s_off[k] = acc;
            acc += sc[k] < sub_cap ? sc[k] : sub_cap;
            mx = sc[k] > mx ? sc[k] : mx;
        }
        s_off[KP_ANCHOR_SUBS] = acc;
        count[a] = acc;
        need[a] = mx;
    }
    __syncthreads();
    const size_t cap = (size_t)sub_cap * KP_ANCHOR_SUBS;
    for (int k = 0; k < KP_ANCHOR_SUBS; ++k) {
        const uint32_t n = s_off[k + 1] - s_off[k];
        const uint64_t *src = sliced + ((size_t)a * KP_ANCHOR_SUBS + k) * sub_cap;
        uint64_t *dst = out + (size_t)a * cap + s_off[k];
        for (uint32_t i = threadIdx.x; i < n; i += blockDim.x) dst[i] = src[i];
    }
}

}  // namespace

void kp_launch_anchor_compact(const KpBatchView &b, const uint64_t *sliced, const uint32_t *sub_count, uint32_t sub_cap,
                              uint64_t *out, uint32_t *count, uint32_t *need, hipStream_t stream) {
    if (b.n_asm == 0) return;
    hipLaunchKernelGGL(kp_anchor_compact_kernel, dim3(b.n_asm), dim3(256), 0, stream, sliced, sub_count, sub_cap, out,
                       count, need);
}

void kp_launch_scan(const KpBatchView &b, const KpSeedIndex &idx, uint64_t *cand, unsigned long long *n_cand,
                    uint64_t cand_cap, uint64_t *anchors, uint32_t *sub_count, uint32_t sub_cap, KpKeyBits key_bits,
                    int mode, bool no_lds, hipStream_t stream, hipEvent_t after_scan) {
    if (b.total_words == 0) return;
    const int64_t n_units = b.total_words >> 2;
    if (idx.lds_filter_blocks && !no_lds && mode == 0) {
        // one 16-wave block per CU (the filter fills most of its LDS); a few blocks per CU in the grid even out the tail
        int64_t blocks = (n_units + 1023) / 1024;
        if (blocks > 256 * 4) blocks = 256 * 4;
        hipLaunchKernelGGL((kp_scan_kernel<0, true>), dim3((unsigned)blocks), dim3(1024), 0, stream, b, idx, cand, n_cand, cand_cap);
    } else {
        int64_t blocks = (n_units + 255) / 256;
        if (blocks > 256 * 8) blocks = 256 * 8;  // 256 CUs x 8 resident blocks, grid-stride beyond that
        const dim3 grid((unsigned)blocks), block(256);
#ifdef KP_SCAN_LANE_OWNED  // (A/B builds: the round-2 kernel, every lane walking its own selected positions)
        if (mode == 1) hipLaunchKernelGGL((kp_scan_kernel<1, false>), grid, block, 0, stream, b, idx, cand, n_cand, cand_cap);
        else if (mode == 2) hipLaunchKernelGGL((kp_scan_kernel<2, false>), grid, block, 0, stream, b, idx, cand, n_cand, cand_cap);
        else hipLaunchKernelGGL((kp_scan_kernel<0, false>), grid, block, 0, stream, b, idx, cand, n_cand, cand_cap);
#else
        if (mode == 1) hipLaunchKernelGGL((kp_scan_dense_kernel<1>), grid, block, 0, stream, b, idx, cand, n_cand, cand_cap);
        else if (mode == 2) hipLaunchKernelGGL((kp_scan_dense_kernel<2>), grid, block, 0, stream, b, idx, cand, n_cand, cand_cap);
        else hipLaunchKernelGGL((kp_scan_dense_kernel<0>), grid, block, 0, stream, b, idx, cand, n_cand, cand_cap);
#endif
    }
    if (after_scan) (void)hipEventRecord(after_scan, stream);
    hipLaunchKernelGGL(kp_expand_kernel, dim3(256 * 8), dim3(256), 0, stream, b, idx, cand, n_cand, cand_cap,
                       anchors, sub_count, sub_cap, key_bits);
}
