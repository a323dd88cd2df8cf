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
// quarter of positions goes on: first a 2^KP_FILTER_LOG2-bit presence filter (2 MB, stays in each XCD's L2) that
// rejects most of them, then the k-mer table (tens of MB, Infinity Cache) for the rest.  Hits are rare outside the
// typed locus, so the per-hit work (assembly / contig / N-run lookup, posting expansion, atomics) is off the
// streaming path.
#include "kp_internal.h"

namespace {

constexpr int PROBES = 8;

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

// A seed at batch word `word`, base `i` matched the table: validate it against contig bounds and N runs, then append
// one anchor per posting to the assembly's region.
__device__ __noinline__ void emit_seed(const KpBatchView &b, const KpSeedIndex &idx, uint32_t first_posting,
                                       int64_t word, int i, uint64_t *anchors, uint32_t *anchor_count, uint32_t cap) {
    const int a = upper_bound_i64(b.asm_word_off, b.n_asm + 1, word) - 1;
    if (a < 0 || a >= b.n_asm) return;
    const int32_t t = (int32_t)((word - b.asm_word_off[a]) * 16 + i);
    const int c0 = b.asm_first_ctg[a], nc = b.asm_first_ctg[a + 1] - c0;
    const int c = upper_bound_i32(b.ctg_start + c0, nc, t) - 1;
    if (c < 0) return;
    if (t + KP_K > b.ctg_start[c0 + c] + b.ctg_len[c0 + c]) return;  // k-mer runs past the contig (or sits in padding)
    const int r0 = b.asm_first_nrun[a], nr = b.asm_first_nrun[a + 1] - r0;
    if (nr > 0) {  // first run whose end is > t; it overlaps the k-mer iff it starts before t + K
        int lo = 0, hi = nr;
        while (lo < hi) {
            int mid = (lo + hi) >> 1;
            if (b.n_runs[2 * (r0 + mid) + 1] <= t) lo = mid + 1; else hi = mid;
        }
        if (lo < nr && b.n_runs[2 * (r0 + lo)] < t + KP_K) return;
    }
    const uint32_t cnt = (uint32_t)idx.postings[first_posting];
    const uint32_t base = atomicAdd(&anchor_count[a], cnt);
    uint64_t *dst = anchors + (size_t)a * cap;
    const uint64_t shift = (uint64_t)(uint32_t)t << 16;
    for (uint32_t j = 0; j < cnt; ++j)
        if (base + j < cap) dst[base + j] = idx.postings[first_posting + 1 + j] + shift;
}

__global__ __launch_bounds__(256) void kp_scan_kernel(KpBatchView b, KpSeedIndex idx, uint64_t *__restrict__ anchors,
                                                       uint32_t *__restrict__ anchor_count, uint32_t cap) {
    const int64_t n_units = b.total_words >> 2;  // 16-byte units; every assembly is a whole number of them
    const int64_t n_iter_units = (n_units + 63) & ~(int64_t)63;  // whole waves iterate together (shuffle below)
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    const uint4 *vec = reinterpret_cast<const uint4 *>(b.words);
    const int lane = threadIdx.x & 63;
    for (int64_t u = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; u < n_iter_units; u += stride) {
        uint4 v = make_uint4(0, 0, 0, 0);
        if (u < n_units) v = vec[u];
        uint32_t next = __shfl_down(v.x, 1);
        if (lane == 63) next = (u + 1 < n_units) ? b.words[(u + 1) << 2] : 0u;
        const uint32_t w[5] = {v.x, v.y, v.z, v.w, next};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const uint32_t lo = w[k], hi = w[k + 1];
            // 2-bit lanes: x = c[p] ^ c[p+1] ^ c[p+3] for the 16 positions of this word
            const uint32_t x = lo ^ __builtin_amdgcn_alignbit(hi, lo, 2) ^ __builtin_amdgcn_alignbit(hi, lo, 6);
            uint32_t sel = x & ~(x >> 1) & 0x55555555u;  // value 01: low bit set, high bit clear
            const uint64_t both = ((uint64_t)hi << 32) | lo;
            while (sel) {
                // up to PROBES selected positions at a time: all their filter words are requested before any is
                // looked at, so a lane keeps several independent L2 reads in flight
                uint32_t kmer[PROBES], filt[PROBES];
                int bit[PROBES];
#pragma unroll
                for (int j = 0; j < PROBES; ++j) {
                    const bool have = sel != 0;
                    bit[j] = have ? __builtin_ctz(sel) : 0;
                    sel &= sel - 1;  // no-op once sel is 0
                    kmer[j] = (uint32_t)(both >> bit[j]) & KP_KMER_MASK;
                    const uint32_t h = (kmer[j] * 2654435769u) >> (32 - KP_FILTER_LOG2);
                    filt[j] = have ? ((idx.filter[h >> 5] >> (h & 31)) & 1u) : 0u;
                }
#pragma unroll
                for (int j = 0; j < PROBES; ++j) {
                    if (!filt[j]) continue;  // ~93 % of selected positions stop here (KpSC K database)
                    uint32_t slot = (kmer[j] * 2654435769u) >> idx.slot_shift;
                    for (;;) {
                        const uint2 e = idx.slots[slot];
                        if (e.x == kmer[j]) {
                            emit_seed(b, idx, e.y, (u << 2) + k, bit[j] >> 1, anchors, anchor_count, cap);
                            break;
                        }
                        if (e.x == 0xFFFFFFFFu) break;
                        slot = (slot + 1) & idx.slot_mask;
                    }
                }
            }
        }
    }
}

}  // namespace

void kp_launch_scan(const KpBatchView &b, const KpSeedIndex &idx, uint64_t *anchors, uint32_t *anchor_count,
                    uint32_t cap, hipStream_t stream) {
    if (b.total_words == 0) return;
    const int64_t n_units = b.total_words >> 2;
    int64_t blocks = (n_units + 255) / 256;
    if (blocks > 256 * 8) blocks = 256 * 8;  // 256 CUs x 8 resident blocks, grid-stride beyond that
    hipLaunchKernelGGL(kp_scan_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, b, idx, anchors, anchor_count, cap);
}
