// kp_bsort.hip -- per-assembly sort of the anchor keys by buckets: one block per assembly.
//
// Input: the sub-slices the expansion kernel appended an assembly's compact keys to (kp_scan.hip); output: the keys of
// every assembly in ascending order in one contiguous run -- what kp_anchor_compact_kernel + the library's segmented
// radix sort produce (kp_sort.hip: six 8-bit passes over 58 M keys per 1000 assemblies, 2.24 + 0.28 ms; this kernel:
// 1.78 ms, profiles/).  Keys are unique, so any correct sort gives the same bytes: kp_batch_anchors and the chaining kernel
// cannot tell the two paths apart (tests/test_gpu_parity.py compares the sorted anchors of both with the oracle's).
//
// The key's top field is gene * 2 + strand, a few thousand values per database, and an assembly's ~6 x 10^4 anchors fall
// into a few thousand of those buckets with a handful to a few hundred keys each.  So:
//   1. histogram of the top field in LDS (one counter per bucket; 0.09 ms), exclusive scan, scatter into bucket order in
//      global memory (0.7 ms: 58 M scattered 8-byte stores, the rate at which the memory behind the L2s takes random
//      requests -- tools/microbench/l2_gather.hip measured 56 G/s for reads; this is the part a radix pass coalesces);
//   2. every bucket is sorted on its own by the cheapest means for its size (1.0 ms): one key is copied; 2..8 keys go
//      through a sorting network in the registers of ONE lane (64 buckets per wave at a time); up to BS_RANK_MAX keys are
//      ranked by a wave against lane broadcasts (ranking = counting the smaller keys: keys are unique, the count is the
//      position), four buckets' loads in flight; up to BS_STAGE keys go through a bitonic network in a wave's registers
//      (shuffles for partners in other lanes, static register pairs within a lane); anything larger -- one gene with
//      thousands of anchors: a 15 kb gene, or a gene present hundreds of times -- is ranked by the whole block at the end.
// No memory access sits inside a sorting loop: an LDS read per comparison ran at the LDS's latency (4.6 ms of a 6 ms first
// version), and a guard per unrolled comparison cost a scalar branch each.
#include <atomic>

#include "kp_internal.h"

namespace {

constexpr int BS_THREADS = 512, BS_WAVES = BS_THREADS / 64;
constexpr int BS_STAGE = 512;     // keys one wave ranks in registers (8 per lane)
constexpr int BS_RANK_MAX = 24;   // buckets up to this size are ranked against lane broadcasts, larger ones go through a bitonic network
constexpr int BS_TILE = 1024;     // keys per LDS tile of the block-wide pass
#ifndef KP_BS_BIG_LIST  // (both list lengths can be cut down at build time so that a test run meets their overflow paths:
#define KP_BS_BIG_LIST 1024  //  tools/gpu_bsort_overflow.sh)
#endif
#ifndef KP_BS_HUGE_LIST
#define KP_BS_HUGE_LIST 64
#endif
constexpr int BS_BIG_LIST = KP_BS_BIG_LIST; // buckets of BS_RANK_MAX + 1 .. BS_STAGE keys remembered for the shared pass (more: sorted where they are met)
constexpr int BS_HUGE_LIST = KP_BS_HUGE_LIST;  // buckets larger than BS_STAGE remembered for the block-wide pass (more: ranked in place, slowly)

__device__ __forceinline__ void wave_lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

__device__ __forceinline__ void cmp_swap(uint64_t &a, uint64_t &b) {
    const bool s = b < a;
    const uint64_t lo = s ? b : a, hi = s ? a : b;
    a = lo; b = hi;
}

// One wave sorts a bucket of up to 64 * R keys with a bitonic network: element e = 64 r + lane lives in register r of
// lane `lane`, padded with the largest value.  Partners less than 64 apart are other lanes' copies of the same register
// (one shuffle each way), partners further apart are other registers of the same lane (static indices: everything is
// unrolled).  About 7 instructions per key and stage against 2 + 2R per broadcast key for the ranking: cheaper from a
// few dozen keys upwards.
template <int R>
__device__ __forceinline__ void bitonic_bucket(const uint64_t *__restrict__ grp, uint64_t *__restrict__ dst, uint32_t bs,
                                               uint32_t bm, int lane) {
    uint64_t k[R];
#pragma unroll
    for (int r = 0; r < R; ++r) k[r] = 64u * r + lane < bm ? grp[bs + 64u * r + lane] : ~0ull;
    constexpr int N = 64 * R;
#pragma unroll
    for (int kk = 2; kk <= N; kk <<= 1) {
#pragma unroll
        for (int j = kk >> 1; j >= 1; j >>= 1) {
#pragma unroll
            for (int r = 0; r < R; ++r) {
                if (j >= 64) {  // partner: register r ^ (j / 64) of this lane; each pair once, from its lower register
                    const int pr = r ^ (j >> 6);
                    if (pr > r) {
                        const bool asc = ((64 * r) & kk) == 0;  // (kk >= 128 here: a property of the register)
                        const bool sw = asc ? (k[pr] < k[r]) : (k[r] < k[pr]);
                        const uint64_t a = sw ? k[pr] : k[r], b = sw ? k[r] : k[pr];
                        k[r] = a; k[pr] = b;
                    }
                } else {  // partner: lane ^ j, same register
                    const uint64_t other = __shfl_xor(k[r], j);
                    const bool asc = kk >= 64 ? (((64 * r) & kk) == 0) : ((lane & kk) == 0);
                    const bool keep_min = ((lane & j) == 0) == asc;
                    const bool other_less = other < k[r];
                    k[r] = (other_less == keep_min) ? other : k[r];
                }
            }
        }
    }
#pragma unroll
    for (int r = 0; r < R; ++r)
        if (64u * r + lane < bm) dst[bs + 64u * r + lane] = k[r];
}

// the same network for one key per lane that is already in a register (64 keys, padded with the largest value)
__device__ __forceinline__ uint64_t bitonic64(uint64_t k, int lane) {
#pragma unroll
    for (int kk = 2; kk <= 64; kk <<= 1) {
#pragma unroll
        for (int j = kk >> 1; j >= 1; j >>= 1) {
            const uint64_t other = __shfl_xor(k, j);
            const bool asc = kk >= 64 ? true : ((lane & kk) == 0);
            const bool keep_min = ((lane & j) == 0) == asc;
            const bool other_less = other < k;
            k = (other_less == keep_min) ? other : k;
        }
    }
    return k;
}

// `out` may be the buffer `sliced` points into: a block reads only its own assembly's region of `sliced`, all of it before
// the first store to that region of `out` (phase 2 starts behind a barrier).
__global__ __launch_bounds__(BS_THREADS) void kp_anchor_bsort_kernel(const uint64_t *sliced,
                                                                     const uint32_t *__restrict__ sub_count, uint32_t sub_cap,
                                                                     uint64_t *__restrict__ grouped, uint64_t *out,
                                                                     uint32_t *__restrict__ count, uint32_t *__restrict__ need,
                                                                     uint32_t n_bins, uint32_t key_shift) {
    extern __shared__ uint32_t s_bin[];  // [n_bins]: counts, then running offsets, then bucket ends
    __shared__ uint32_t s_off[KP_ANCHOR_SUBS + 1];
    __shared__ uint32_t s_part[BS_THREADS];
    __shared__ uint64_t s_tile[BS_TILE];  // the block-wide pass streams a huge bucket through it
    __shared__ uint32_t s_huge[BS_HUGE_LIST][2];
    __shared__ uint32_t s_n_huge;
    __shared__ uint32_t s_big[BS_BIG_LIST][2];
    __shared__ uint32_t s_n_big, s_big_next;
    const int a = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const size_t cap = (size_t)sub_cap * KP_ANCHOR_SUBS;
    const uint32_t *sc = sub_count + (size_t)a * KP_ANCHOR_SUBS;
    static_assert(KP_ANCHOR_SUBS == 64, "one sub-slice per lane below");
    if (wave == 0) {  // offsets of the sub-slices: one per lane, a wave scan (a single thread walking them is 64 trips to memory)
        const uint32_t raw = sc[lane], mine = raw < sub_cap ? raw : sub_cap;
        uint32_t incl = mine, mx = raw;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t u = __shfl_up(incl, o);
            if (lane >= o) incl += u;
            mx = max(mx, (uint32_t)__shfl_xor(mx, o));
        }
        s_off[lane] = incl - mine;
        if (lane == 63) {
            s_off[KP_ANCHOR_SUBS] = incl;
            count[a] = incl;
            need[a] = mx;
            s_n_huge = 0; s_n_big = 0; s_big_next = 0;
        }
    }
    for (uint32_t i = tid; i < n_bins; i += BS_THREADS) s_bin[i] = 0;
    __syncthreads();
    const uint32_t n = s_off[KP_ANCHOR_SUBS];
    if (n == 0) return;
    const uint64_t *src_a = sliced + (size_t)a * cap;
    uint64_t *grp = grouped + (size_t)a * cap, *dst = out + (size_t)a * cap;

    // ---- 1. histogram, scan, scatter ---------------------------------------------------------------------------------------
    // A wave takes whole sub-slices and keeps UNR independent loads in flight per lane (a block is latency-bound otherwise).
    constexpr int UNR = 8;
    for (int k = wave; k < KP_ANCHOR_SUBS; k += BS_WAVES) {
        const uint32_t nk = s_off[k + 1] - s_off[k];
        const uint64_t *src = src_a + (size_t)k * sub_cap;
        for (uint32_t i0 = lane; i0 < nk; i0 += 64 * UNR) {
            uint64_t key[UNR];
#pragma unroll
            for (int u = 0; u < UNR; ++u) key[u] = i0 + 64 * u < nk ? src[i0 + 64 * u] : 0ull;
#pragma unroll
            for (int u = 0; u < UNR; ++u)
                if (i0 + 64 * u < nk) {
                    const uint32_t bin = (uint32_t)(key[u] >> key_shift);
                    atomicAdd(&s_bin[bin < n_bins ? bin : n_bins - 1], 1u);
                }
        }
    }
    __syncthreads();
    {   // exclusive scan of s_bin: a contiguous piece per thread, then the pieces' sums
        const uint32_t per = (n_bins + BS_THREADS - 1) / BS_THREADS, lo = tid * per, hi = min(n_bins, lo + per);
        uint32_t sum = 0;
        for (uint32_t i = lo; i < hi; ++i) sum += s_bin[i];
        s_part[tid] = sum;
        __syncthreads();
        if (wave == 0) {  // 512 partial sums: eight per lane, then a wave scan
            uint32_t v[BS_WAVES], t = 0;
#pragma unroll
            for (int j = 0; j < BS_WAVES; ++j) { v[j] = s_part[lane * BS_WAVES + j]; t += v[j]; }
            uint32_t incl = t;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const uint32_t u = __shfl_up(incl, o);
                if (lane >= o) incl += u;
            }
            uint32_t run = incl - t;
#pragma unroll
            for (int j = 0; j < BS_WAVES; ++j) { s_part[lane * BS_WAVES + j] = run; run += v[j]; }
        }
        __syncthreads();
        uint32_t run = s_part[tid];
        for (uint32_t i = lo; i < hi; ++i) {
            const uint32_t c = s_bin[i];
            s_bin[i] = run;
            run += c;
        }
    }
    __syncthreads();
    for (int k = wave; k < KP_ANCHOR_SUBS; k += BS_WAVES) {
        const uint32_t nk = s_off[k + 1] - s_off[k];
        const uint64_t *src = src_a + (size_t)k * sub_cap;
        for (uint32_t i0 = lane; i0 < nk; i0 += 64 * UNR) {
            uint64_t key[UNR];
#pragma unroll
            for (int u = 0; u < UNR; ++u) key[u] = i0 + 64 * u < nk ? src[i0 + 64 * u] : 0ull;
#pragma unroll
            for (int u = 0; u < UNR; ++u)
                if (i0 + 64 * u < nk) {
                    const uint32_t bin = (uint32_t)(key[u] >> key_shift);
                    const uint32_t pos = atomicAdd(&s_bin[bin < n_bins ? bin : n_bins - 1], 1u);
                    grp[pos] = key[u];
                }
        }
    }
    __threadfence_block();
    __syncthreads();  // s_bin[b] is now the end of bucket b; its start is the end of bucket b - 1

    // ---- 2. buckets, 64 per wave at a time ---------------------------------------------------------------------------------------
    for (uint32_t c0 = 64u * wave; c0 < n_bins; c0 += 64u * BS_WAVES) {
        const uint32_t bin = c0 + lane;
        uint32_t start = 0, m = 0;
        if (bin < n_bins) {
            start = bin ? s_bin[bin - 1] : 0u;
            m = s_bin[bin] - start;
        }
        if (m == 1) dst[start] = grp[start];
        if (__any(m >= 2 && m <= 8)) {  // one lane, one bucket: Batcher's network for 8 keys, padded with the largest value
            uint64_t k[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) k[j] = (m >= 2 && m <= 8 && (uint32_t)j < m) ? grp[start + j] : ~0ull;
            cmp_swap(k[0], k[1]); cmp_swap(k[2], k[3]); cmp_swap(k[4], k[5]); cmp_swap(k[6], k[7]);
            cmp_swap(k[0], k[2]); cmp_swap(k[1], k[3]); cmp_swap(k[4], k[6]); cmp_swap(k[5], k[7]);
            cmp_swap(k[1], k[2]); cmp_swap(k[5], k[6]);
            cmp_swap(k[0], k[4]); cmp_swap(k[1], k[5]); cmp_swap(k[2], k[6]); cmp_swap(k[3], k[7]);
            cmp_swap(k[2], k[4]); cmp_swap(k[3], k[5]);
            cmp_swap(k[1], k[2]); cmp_swap(k[3], k[4]); cmp_swap(k[5], k[6]);
#pragma unroll
            for (int j = 0; j < 8; ++j)
                if (m >= 2 && m <= 8 && (uint32_t)j < m) dst[start + j] = k[j];
        }
        unsigned long long mid = __ballot(m > 8 && m <= BS_RANK_MAX);
        while (mid) {  // the whole wave on one bucket of up to 64 keys; four buckets' loads are in flight together
            constexpr int NB = 4;
            uint32_t bs[NB], bm[NB];
            uint64_t key[NB];
#pragma unroll
            for (int u = 0; u < NB; ++u) {
                const int owner = mid ? __builtin_ctzll(mid) : 0;
                bs[u] = (uint32_t)__shfl((int)start, owner);
                bm[u] = mid ? (uint32_t)__shfl((int)m, owner) : 0u;
                mid &= mid - 1;  // (no-op once empty)
                key[u] = (uint32_t)lane < bm[u] ? grp[bs[u] + lane] : ~0ull;
            }
#pragma unroll
            for (int u = 0; u < NB; ++u) {
                const uint32_t klo = (uint32_t)key[u], khi = (uint32_t)(key[u] >> 32);
                uint32_t rank = 0;
                for (uint32_t j = 0; j < bm[u]; ++j) {
                    const uint32_t olo = (uint32_t)__builtin_amdgcn_readlane((int)klo, (int)j);
                    const uint32_t ohi = (uint32_t)__builtin_amdgcn_readlane((int)khi, (int)j);
                    // (keys are unique -- one posting per gene seed, one candidate per contig position, the streaming and the
                    // edge kernel own disjoint positions -- but equal keys would still get slots of their own: source order)
                    const uint64_t other = ((uint64_t)ohi << 32) | olo;
                    rank += (other < key[u] || (other == key[u] && j < (uint32_t)lane)) ? 1u : 0u;
                }
                if ((uint32_t)lane < bm[u]) dst[bs[u] + rank] = key[u];
            }
        }
        // Buckets that take a wave's network (BS_RANK_MAX + 1 .. BS_STAGE keys) are only listed here and shared out below: they
        // are the genes of the typed locus and their relatives, neighbours in the gene order, so they all fall into one or
        // two of these 64-bin pieces -- one wave sorted them one after the other while the block's other seven waited
        // (1.80 against 1.64 ms; A/B: -DKP_BS_NO_LIST).
        const bool listed_size = m > BS_RANK_MAX && m <= BS_STAGE;
        const unsigned long long to_list = __ballot(listed_size);
        uint32_t list_base = 0;
        if (to_list) {
            if (lane == 0) list_base = atomicAdd(&s_n_big, (uint32_t)__builtin_popcountll(to_list));
            list_base = (uint32_t)__shfl((int)list_base, 0);
        }
        const uint32_t list_at = list_base + (uint32_t)__builtin_popcountll(to_list & ((1ull << lane) - 1ull));
#ifdef KP_BS_NO_LIST
        const bool listed = false;
#else
        const bool listed = listed_size && list_at < BS_BIG_LIST;
#endif
        if (listed) { s_big[list_at][0] = start; s_big[list_at][1] = m; }
        unsigned long long big = __ballot(m > BS_RANK_MAX && !listed);
        while (big) {  // the whole wave on one bucket (the list was full, or the bucket is larger than BS_STAGE)
            const int owner = __builtin_ctzll(big);
            big &= big - 1;
            const uint32_t bs = (uint32_t)__shfl((int)start, owner), bm = (uint32_t)__shfl((int)m, owner);
            if (bm <= BS_STAGE) {
                if (bm <= 64) bitonic_bucket<1>(grp, dst, bs, bm, lane);  // (uniform)
                else if (bm <= 128) bitonic_bucket<2>(grp, dst, bs, bm, lane);
                else if (bm <= 256) bitonic_bucket<4>(grp, dst, bs, bm, lane);
                else bitonic_bucket<BS_STAGE / 64>(grp, dst, bs, bm, lane);
            } else if (lane == 0) {
                const uint32_t slot = atomicAdd(&s_n_huge, 1u);
                if (slot < BS_HUGE_LIST) { s_huge[slot][0] = bs; s_huge[slot][1] = bm; }
                else {  // (practically never: more than BS_HUGE_LIST buckets beyond BS_STAGE keys in one assembly)
                    for (uint32_t i = 0; i < bm; ++i) {
                        const uint64_t key = grp[bs + i];
                        uint32_t rank = 0;
                        for (uint32_t j = 0; j < bm; ++j) rank += (grp[bs + j] < key || (grp[bs + j] == key && j < i)) ? 1u : 0u;
                        dst[bs + rank] = key;
                    }
                }
            }
        }
    }
    __syncthreads();
    // ---- 2b. the listed buckets: every wave takes the next one until none is left ----------------------------------------------
    {
        const uint32_t n_big = min(s_n_big, (uint32_t)BS_BIG_LIST);
#ifndef KP_BS_NB
#define KP_BS_NB 1
#endif
        constexpr int NB = KP_BS_NB;  // buckets taken at a time (more, with their loads in flight together: 1.72 / 1.74 ms for 4 / 8 against 1.64)
        for (;;) {
            uint32_t at = 0;
            if (lane == 0) at = atomicAdd(&s_big_next, (uint32_t)NB);
            at = (uint32_t)__shfl((int)at, 0);
            if (at >= n_big) break;
            uint32_t bs[NB], bm[NB];
            uint64_t key[NB];
#pragma unroll
            for (int u = 0; u < NB; ++u) {
                const bool have = at + u < n_big;
                bs[u] = have ? s_big[at + u][0] : 0u;
                bm[u] = have ? s_big[at + u][1] : 0u;
                key[u] = (bm[u] <= 64 && (uint32_t)lane < bm[u]) ? grp[bs[u] + lane] : ~0ull;
            }
#pragma unroll
            for (int u = 0; u < NB; ++u) {
                if (bm[u] == 0) continue;  // (uniform)
                if (bm[u] <= 64) {
                    const uint64_t k = bitonic64(key[u], lane);
                    if ((uint32_t)lane < bm[u]) dst[bs[u] + lane] = k;
                } else if (bm[u] <= 128) bitonic_bucket<2>(grp, dst, bs[u], bm[u], lane);
                else if (bm[u] <= 256) bitonic_bucket<4>(grp, dst, bs[u], bm[u], lane);
                else bitonic_bucket<BS_STAGE / 64>(grp, dst, bs[u], bm[u], lane);
            }
        }
    }
    // ---- 3. the few buckets beyond BS_STAGE keys: ranked by the whole block, tile by tile through LDS ---------------------
    const uint32_t n_huge = min(s_n_huge, (uint32_t)BS_HUGE_LIST);
    uint64_t *tile = s_tile;
    constexpr uint32_t TILE = BS_TILE;
    for (uint32_t h = 0; h < n_huge; ++h) {
        const uint32_t bs = s_huge[h][0], bm = s_huge[h][1];
        for (uint32_t i0 = 0; i0 < bm; i0 += BS_THREADS) {  // BS_THREADS keys are ranked per round
            const uint32_t i = i0 + tid;
            const uint64_t key = i < bm ? grp[bs + i] : 0ull;
            uint32_t rank = 0;
            for (uint32_t t0 = 0; t0 < bm; t0 += TILE) {
                const uint32_t tn = min(TILE, bm - t0);
                __syncthreads();
                for (uint32_t j = tid; j < tn; j += BS_THREADS) tile[j] = grp[bs + t0 + j];
                __syncthreads();
                if (i < bm)
                    for (uint32_t j = 0; j < tn; ++j) rank += (tile[j] < key || (tile[j] == key && t0 + j < i)) ? 1u : 0u;
            }
            if (i < bm) dst[bs + rank] = key;
        }
    }
}

}  // namespace

size_t kp_bsort_lds_bytes(uint32_t n_bins) { return (size_t)n_bins * sizeof(uint32_t); }

// The bucket counters are dynamic LDS on top of ~19 KB of static LDS (s_tile, s_big, s_part, s_huge, s_off); gfx950 has
// 160 KB per CU.  BS_DYN_MAX = 128 KB of counters = 32768 buckets.  The Kaptive-shaped databases (a few thousand genes:
// 15-30 KB, three or more blocks per CU) have one bucket per value of the top key field (gene * 2 + strand); a context
// with more than 16384 genes (several large databases at once) gets buckets of 2, 4 or 8 neighbouring values -- a bucket
// is sorted on the whole key, so what it spans does not matter to the result -- and one block per CU.
constexpr size_t BS_DYN_MAX = 128u * 1024u;

// how many low bits of the top key field a bucket spans for a database with n_bins = 2 * genes values of it
static uint32_t bucket_span_bits(uint32_t n_bins) {
    uint32_t s = 0;
    while (kp_bsort_lds_bytes((n_bins + (1u << s) - 1) >> s) > BS_DYN_MAX) ++s;
    return s;
}

// true when the bucket path can take a database with n_bins = 2 * genes values of the top key field (always, up to
// KP_MAX_GENES; the library's radix sort stays behind the `library_sort` option)
bool kp_bsort_fits(uint32_t n_bins) { return n_bins > 0; }

void kp_launch_anchor_bsort(const KpBatchView &b, const uint64_t *sliced, const uint32_t *sub_count, uint32_t sub_cap,
                            uint64_t *grouped, uint64_t *out, uint32_t *count, uint32_t *need, uint32_t n_bins,
                            KpKeyBits kb, hipStream_t stream) {
    if (b.n_asm == 0) return;
    // more than 64 KB of LDS per block has to be asked for, per device (the attribute belongs to the device's copy of the
    // kernel): once per device and process, whichever thread comes first
    static std::atomic<uint64_t> raised{0};
    int dev = 0;
    (void)hipGetDevice(&dev);
    const uint64_t bit = 1ull << (dev & 63);
    if (!(raised.load(std::memory_order_acquire) & bit)) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kp_anchor_bsort_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  (int)BS_DYN_MAX);
        raised.fetch_or(bit, std::memory_order_release);
    }
    const uint32_t span = bucket_span_bits(n_bins), n_buckets = (n_bins + (1u << span) - 1) >> span;
    hipLaunchKernelGGL(kp_anchor_bsort_kernel, dim3(b.n_asm), dim3(BS_THREADS), kp_bsort_lds_bytes(n_buckets), stream, sliced,
                       sub_count, sub_cap, grouped, out, count, need, n_buckets, kb.qb + kb.db + span);
}
