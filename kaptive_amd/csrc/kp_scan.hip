// kp_scan.hip -- seed scan: stream the 2-bit packed contigs once, emit anchors against the resident seed index.
//
// Stands in for the seeding half of rammappy's map_batch (reference call site src/kaptive/serotyping/core.py:154); the
// rule and key layout are those of include/kp_spec.h.  This is the only kernel that touches every base of every
// assembly, so it is the one priced against the HBM-read roofline: algorithmic bytes = total_words * 4 per launch.
//
// Seeds are minimap2's (10, 15) minimizers (kp_spec.h, v3).  Mapping: one lane owns 64 consecutive bases (one 16-byte
// load, lanes of a wave are contiguous), computes the value of the 64 15-mers that start there (canonical 15-mer ->
// kp_hash30, all 32-bit) and decides which of them are window minima; the nine values it needs from either neighbour
// cross lanes by DPP.  Only the selected positions (2 / 11 of all) go on to the blocked-Bloom presence filter (2 MB, stays
// in each XCD's L2; kp_internal.h).  What passes the filters is only recorded (kp_scan_dense_kernel); contig ends and
// the flanks of N runs, where minimap2's state machine does not reduce to the window rule, are handled by
// kp_edge_kernel; a balanced kernel (kp_expand_kernel) probes the seed table, validates and expands the postings into
// anchors.
#include "kp_internal.h"
#include "kp_sketch.h"

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

// ---- pass 1: stream + minimizers + presence filter -> candidate positions ------------------------------------------------
// Every lane computes x(p) (kp_spec.h) for the 64 positions of its unit; the stream is taken as one clean sequence (N runs
// and the padding between contigs are code 0): positions whose answer that falsifies are not this kernel's
// (kp_seed_is_interior) and are rejected by the expansion.  p is a seed iff x(p) is a smallest value of one of the ten
// windows of ten consecutive 15-mers that contain it, i.e. iff
//      max over s in [p - 9, p] of ( min over [s, s + 9] of x )  ==  x(p)
// (every window that contains p has a minimum <= x(p); ties -- equal canonical 15-mers within a window -- select all
// their positions, as mm_sketch does).  A wave's lanes 0 and 63 only supply their neighbours' flanks: successive wave
// iterations overlap by two units (62 of 64 lanes emit).
//
// Both strands' 15-mers come straight off the packed words with one v_alignbit each: the stream's own bit order is the
// reverse strand's (complemented words), the words with their sixteen bases reversed give the forward strand's; the two
// values are taken top-aligned (a stray neighbouring base in bits 0-1 cannot decide the comparison: k is odd, the 30 bits
// above differ) and the smaller one, shifted down, is hashed.  The strand bit is not needed to select: it is worked out
// for the one seed in two hundred that passes the filter.
//
// Selected positions go into a per-wave list in LDS as (position, x), eight positions of every lane at a time (a ballot
// ranks the lanes), and the list is walked 64 entries at a time with every lane busy: entry -> filter block (one L2 read,
// PROBES rounds in flight) -> bit test; what passes is staged per wave, checked against the second filter at flush time
// and written as one word each.  MODE 0 = product; 1 = no filter reads; 2 = stream only (tools/scan_ablate.py).
#ifndef KP_SCAN_PROBES
#define KP_SCAN_PROBES 4
#endif
constexpr int PROBES = KP_SCAN_PROBES;  // filter reads a lane has in flight
constexpr int DENSE_WAVES = 4;
// entries of a wave's list (an iteration selects ~720; eight positions add <= 512).  1024, not 1280 (round 6): the block's LDS
// drops from 45 to 39 KB and a CU holds four blocks instead of three -- 5.74 ms alone against 6.0, +1 % on the overlapped step
// (walks start a little earlier: at 512 listed entries instead of 768)
#ifndef KP_SCAN_DENSE_LIST
#define KP_SCAN_DENSE_LIST 1024
#endif
constexpr int DENSE_LIST = KP_SCAN_DENSE_LIST;
constexpr int DENSE_STAGE = 320;   // candidates staged per wave (a round of the walk adds <= 64 * PROBES)
constexpr int SCAN_OWN = 62;       // lanes of a wave iteration that emit (lanes 1..62)
#ifndef KP_SCAN_WAVES_PER_SIMD
#define KP_SCAN_WAVES_PER_SIMD 3
#endif

__device__ __forceinline__ uint32_t lane_from_below(uint32_t v) {  // lane L receives lane L - 1's value (lane 0: 0)
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x138, 0xf, 0xf, false);  // wave_shr:1
}
__device__ __forceinline__ uint32_t lane_from_above(uint32_t v) {  // lane L receives lane L + 1's value (lane 63: 0)
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x130, 0xf, 0xf, false);  // wave_shl:1
}
// the sixteen 2-bit groups of a word in reverse order
__device__ __forceinline__ uint32_t reverse_groups(uint32_t w) {
    const uint32_t r = __builtin_bitreverse32(w);
    return ((r >> 1) & 0x55555555u) | ((r & 0x55555555u) << 1);
}
__device__ __forceinline__ uint32_t min3u(uint32_t a, uint32_t b, uint32_t c) { return min(min(a, b), c); }
__device__ __forceinline__ uint32_t max3u(uint32_t a, uint32_t b, uint32_t c) { return max(max(a, b), c); }

// strand bit of the 15-mer whose bases sit in the low 30 bits of `e` (first base in bits 0-1): 0 = the forward 15-mer is
// the canonical one
__device__ __forceinline__ uint32_t strand_bit_from_low(uint32_t e) {
    const uint32_t rev = e ^ KP_KMER_MASK;  // complement; the stream's layout is the reverse complement's
    const uint32_t r = __builtin_bitreverse32(e) >> 2;
    const uint32_t fwd = ((r >> 1) & 0x15555555u) | ((r & 0x15555555u) << 1);
    return fwd < rev ? 0u : 1u;
}

template <int MODE>
__global__ __launch_bounds__(64 * DENSE_WAVES, KP_SCAN_WAVES_PER_SIMD) void kp_scan_dense_kernel(
    KpBatchView b, KpSeedIndex idx, uint64_t *__restrict__ cand, unsigned long long *__restrict__ n_cand, uint64_t cand_cap) {
    __shared__ uint64_t s_stage[DENSE_WAVES][DENSE_STAGE];
    __shared__ __attribute__((aligned(16))) uint32_t s_words[DENSE_WAVES][260];
    __shared__ uint32_t s_lx[DENSE_WAVES][DENSE_LIST];
    __shared__ uint16_t s_lp[DENSE_WAVES][DENSE_LIST];
    const uint2 *g_filter = reinterpret_cast<const uint2 *>(idx.filter);
    const uint2 *g_filter2 = reinterpret_cast<const uint2 *>(idx.filter2);
    uint32_t checksum = 0;
    const int64_t n_units = b.total_words >> 2;
    const int64_t n_iters = (n_units + SCAN_OWN - 1) / SCAN_OWN;
    const uint4 *vec = reinterpret_cast<const uint4 *>(b.words);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint64_t *stage = s_stage[wave];
    uint32_t *words = s_words[wave];
    uint32_t *lx = s_lx[wave];
    uint16_t *lp = s_lp[wave];
    uint32_t staged = 0;  // wave-uniform
    uint32_t listed = 0;  // wave-uniform: entries in the list
    int64_t wave_base = 0;  // batch-wide position of lane 0's first base in the current iteration (may be -64)
    const unsigned long long below = (1ull << lane) - 1ull;

    auto flush = [&]() {  // second filter on what is staged, survivors out with one atomic
        uint32_t kept = 0;
        for (uint32_t i0 = 0; i0 < staged; i0 += 64) {
            const uint32_t i = i0 + lane;
            uint64_t c = 0;
            bool ok = false;
            if (i < staged) {
                c = stage[i];
                const uint32_t x = (uint32_t)c & KP_KMER_MASK;
                const uint2 got = g_filter2[kp_filter2_block(x)], need = kp_filter2_mask2(x);
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

    auto walk = [&]() {  // the list -> filter probes -> stage
        wave_lds_sync();  // the list and the words are visible
        for (uint32_t e0 = 0; e0 < listed; e0 += 64 * PROBES) {
            uint32_t xs[PROBES], pos[PROBES];
            bool have[PROBES];
            uint2 got[PROBES];
#pragma unroll
            for (int j = 0; j < PROBES; ++j) {
                const uint32_t e = e0 + 64 * j + lane;
                have[j] = e < listed;
                xs[j] = have[j] ? lx[e] : 0u;
                pos[j] = have[j] ? lp[e] : 0u;
                if (MODE == 1) { checksum += xs[j] * 2654435769u; continue; }
                got[j] = have[j] ? g_filter[kp_filter_block(xs[j])] : make_uint2(0u, 0u);
            }
            if (MODE != 0) continue;
#pragma unroll
            for (int j = 0; j < PROBES; ++j) {
                const bool hit = have[j] && kp_filter_test(got[j], xs[j]);
                const unsigned long long ballot = __ballot(hit);
                if (!ballot) continue;
                if (hit) {
                    const uint32_t p = pos[j];
                    const uint32_t lo = words[p >> 4], hi = words[(p >> 4) + 1];
                    const uint32_t z = strand_bit_from_low(__builtin_amdgcn_alignbit(hi, lo, 2 * (p & 15u)) & KP_KMER_MASK);
                    stage[staged + (uint32_t)__builtin_popcountll(ballot & below)] = kp_cand_pack((uint64_t)(wave_base + p), z, xs[j]);
                }
                staged += (uint32_t)__builtin_popcountll(ballot);
            }
            if (staged > DENSE_STAGE - 64 * PROBES) flush();
        }
        listed = 0;
        wave_lds_sync();  // the walk is done with the list before the next entries arrive
    };

    const int64_t wave_stride = (int64_t)gridDim.x * DENSE_WAVES;
    for (int64_t it = (int64_t)blockIdx.x * DENSE_WAVES + wave; it < n_iters; it += wave_stride) {
        const int64_t u = it * SCAN_OWN - 1 + lane;  // lane 0 repeats the previous iteration's last unit, lane 63 the next's first
        uint4 v = make_uint4(0, 0, 0, 0);
        if (u >= 0 && u < n_units) v = vec[u];
        const uint32_t next = lane_from_above(v.x);
        if (MODE == 2) { checksum += v.x ^ v.y ^ v.z ^ v.w ^ next; continue; }
        wave_base = (u - lane) * 64;
        *reinterpret_cast<uint4 *>(&words[4 * lane]) = v;  // (lane 63 emits nothing: nobody reads past its unit)
        const uint32_t w[5] = {v.x, v.y, v.z, v.w, next};
        uint32_t cw[5], rw[5];  // complemented words: the reverse strand's layout; bases reversed: the forward strand's
#pragma unroll
        for (int k = 0; k < 5; ++k) { cw[k] = ~w[k]; rw[k] = reverse_groups(w[k]); }
        // x of the lane's 64 positions with nine of either neighbour's on both sides: X[9 + p], p in [-9, 72]
        uint32_t X[82];
#pragma unroll
        for (int p = 0; p < 64; ++p) {
            const int a = p >> 4, o = p & 15;
            const uint32_t fwd = o == 0 ? rw[a] : __builtin_amdgcn_alignbit(rw[a], rw[a + 1], 32 - 2 * o);
            const uint32_t rev = o == 0 ? cw[a] << 2 : __builtin_amdgcn_alignbit(cw[a + 1], cw[a], 2 * o - 2);
            X[9 + p] = kp_hash30(min(fwd, rev) >> 2);
        }
#pragma unroll
        for (int j = 0; j < 9; ++j) {
            X[j] = lane_from_below(X[9 + 55 + j]);
            X[9 + 64 + j] = lane_from_above(X[9 + j]);
        }
        // window minima (index s + 9 holds the minimum over positions [s, s + 9]), then the maxima of ten of those
        uint32_t c1[80], c2[76], wm[73], d1[71], d2[67];
#pragma unroll
        for (int i = 0; i < 80; ++i) c1[i] = min3u(X[i], X[i + 1], X[i + 2]);
#pragma unroll
        for (int i = 0; i < 76; ++i) c2[i] = min3u(c1[i], c1[i + 2], c1[i + 4]);
#pragma unroll
        for (int i = 0; i < 73; ++i) wm[i] = min(c2[i], c2[i + 3]);
#pragma unroll
        for (int i = 0; i < 71; ++i) d1[i] = max3u(wm[i], wm[i + 1], wm[i + 2]);
#pragma unroll
        for (int i = 0; i < 67; ++i) d2[i] = max3u(d1[i], d1[i + 2], d1[i + 4]);
        const bool owner = lane >= 1 && lane <= SCAN_OWN && u < n_units;
        const uint32_t lane64 = 64u * (uint32_t)lane;
        const unsigned long long owner_mask = __builtin_amdgcn_ballot_w64(owner);
#pragma unroll
        for (int p0 = 0; p0 < 64; p0 += 8) {
#pragma unroll
            for (int p = p0; p < p0 + 8; ++p) {
                // selected: the maximum over the windows that start in [p - 9, p] is x(p) itself; the comparison lands in a scalar
                // register pair, the owner mask is applied there, and the same pair predicates the two stores
                const unsigned long long ballot = __builtin_amdgcn_ballot_w64(max(d2[p], d2[p + 3]) == X[9 + p]) & owner_mask;
                const uint32_t at = __builtin_amdgcn_mbcnt_hi((uint32_t)(ballot >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)ballot, listed));
                if (__builtin_amdgcn_inverse_ballot_w64(ballot)) {
                    lx[at] = X[9 + p];
                    lp[at] = (uint16_t)(lane64 + p);
                }
                listed += (uint32_t)__builtin_popcountll(ballot);
            }
            if (p0 == 56 || listed > DENSE_LIST - 512) walk();
        }
    }
    if (MODE == 0 && staged) flush();
    if (MODE != 0 && checksum == 0x9E3779B1u) n_cand[0] = checksum;
}

// ---- pass 1b: the seeds next to contig ends and N runs ---------------------------------------------------------------------
// Two threads per contig walk its clean stretches [S, E) (no ambiguous base) and run kp_spec.h's state machine
// (kp_sketch.h) where kp_seed_is_interior says the streaming kernel must not decide: over the first bases of a stretch
// from a fresh state (at a contig start that is mm_sketch's own start; after an N run every older entry has left the
// window and the tracked minimum has been dropped by the time the stretch's first 15-mer is complete), and over its last
// bases after a warm-up of 48 steps from a fresh state (window, minimum and run length depend on the last KP_W + KP_K
// steps only; what the warm-up settles wrongly lies before the positions kept).  Past the stretch's end it keeps stepping
// through ambiguous bases until the tracked minimum has left the window -- dropped, as mm_sketch drops it -- or the
// contig ends, where the minimum is a seed.  Seeds that pass both presence filters are appended at the BACK of the
// candidate list (n_cand[1]); they need no validation.
__global__ __launch_bounds__(256) void kp_edge_kernel(KpBatchView b, KpSeedIndex idx, uint64_t *__restrict__ cand,
                                                       unsigned long long *__restrict__ n_cand, uint64_t cand_cap,
                                                       int32_t n_ctg_total) {
    const uint2 *g_filter = reinterpret_cast<const uint2 *>(idx.filter);
    const uint2 *g_filter2 = reinterpret_cast<const uint2 *>(idx.filter2);
    // two threads per contig: the left flanks of its stretches (and stretches too short to have two) and the right flanks --
    // the kernel is a chain of dependent steps per thread (2.0 ms per 1000 assemblies of 1500 contigs with one thread per contig)
    const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int32_t c = (int32_t)(tid >> 1);
    const int side = (int)(tid & 1);
#ifdef KP_EDGE_ONE_THREAD
    if (side == 1) return;
#endif
    if (c >= n_ctg_total) return;
    const int a = upper_bound_i32(b.asm_first_ctg, b.n_asm + 1, c) - 1;  // the contig's assembly
    const int64_t asm_base = (int64_t)b.asm_word_off[a] << 4;
    const uint32_t *aw = b.words + b.asm_word_off[a];
    const int64_t cs = b.ctg_start[c], ce = cs + b.ctg_len[c];
    const int r0 = b.asm_first_nrun[a], nr = b.asm_first_nrun[a + 1] - r0;
    int r = 0;  // first N run that ends after the contig's start
    {
        int lo = 0, hi = nr;
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (b.n_runs[2 * (r0 + mid) + 1] <= cs) lo = mid + 1; else hi = mid;
        }
        r = lo;
    }
    int64_t S = cs;
    while (S < ce) {
        // the stretch [S, E): up to the next N run inside the contig
        int64_t E = ce, resume = ce;
        if (r < nr && b.n_runs[2 * (r0 + r)] < ce) {
            E = max((int64_t)b.n_runs[2 * (r0 + r)], S);
            resume = min((int64_t)b.n_runs[2 * (r0 + r) + 1], ce);
            ++r;
        }
        if (E - S >= KP_K) {
            bool keep_left = true, keep_right = true;  // which flank the current run settles
            auto emit = [&](int64_t t, uint32_t z, uint32_t x) {
                if (t < S || kp_seed_is_interior(t, S, E)) return;  // a warm-up artefact, or the streaming kernel's
                if (t < S + KP_W ? !keep_left : !keep_right) return;  // (a position in both flanks goes with the left one)
                const uint2 g1 = g_filter[kp_filter_block(x)], n1 = kp_filter_mask2(x);
                if ((g1.x & n1.x) != n1.x || (g1.y & n1.y) != n1.y) return;
                const uint2 g2 = g_filter2[kp_filter2_block(x)], n2 = kp_filter2_mask2(x);
                if ((g2.x & n2.x) != n2.x || (g2.y & n2.y) != n2.y) return;
                const unsigned long long k = atomicAdd(n_cand + 1, 1ull);
                if (k < cand_cap) cand[cand_cap - 1 - k] = kp_cand_pack((uint64_t)(asm_base + t), z, x);
            };
            auto run = [&](int64_t from, int64_t to) {  // steps [from, to) from a fresh state; bases at or past E are ambiguous
                KpSketchState st;
                kp_sketch_reset(st);
                for (int64_t i = from; i < to; ++i) {
                    const uint32_t code = i < E ? ((aw[i >> 4] >> (2 * (i & 15))) & 3u) : 4u;
                    kp_sketch_step(st, i, code, emit);
                }
                if (to == ce) kp_sketch_final(st, ce - 1, emit);
            };
            const int64_t stop = min(E + KP_W, ce);  // by then the tracked minimum has left the window
            const int64_t warm = E - (KP_K + KP_W) - 48;
            if (warm <= S) {
                if (side == 0) run(S, stop);
            } else if (side == 0) {
                keep_right = false;
                run(S, S + 2 * KP_W + KP_K);  // settles every position before S + KP_W (retired by step S + KP_W - 1 + KP_K - 1 + KP_W)
#ifdef KP_EDGE_ONE_THREAD
                keep_right = true, keep_left = false;
                run(warm, stop);
#endif
            } else {
                keep_left = false;
                run(warm, stop);
            }
        }
        S = resume;
    }
}

// ---- pass 2: candidates -> anchors -------------------------------------------------------------------------------------
// One thread per candidate: probe the table, find its assembly, contig and clean stretch, keep it if it is the streaming
// kernel's to decide (kp_seed_is_interior; the edge kernel's candidates, at the back of the list, were decided by the
// state machine and are taken as they are), reserve room for its postings in its assembly's anchor region.  The postings
// themselves are then copied by the whole wave as one
// flat list (prefix sums of the counts, every lane finds the owner of its output item by binary search): a shared gene
// family has ~160 postings per seed, a per-thread copy loop would run the wave at the pace of its longest list.  Appends of one assembly are spread over
// KP_ANCHOR_SUBS counters because all hits of an assembly come in one burst (the typed locus) and would otherwise
// serialise on a single atomic word.
__global__ __launch_bounds__(256) void kp_expand_kernel(KpBatchView b, KpSeedIndex idx, const uint64_t *__restrict__ cand,
                                                         const unsigned long long *__restrict__ n_cand, uint64_t cand_cap,
                                                         uint64_t *__restrict__ anchors, uint32_t *__restrict__ sub_count,
                                                         uint32_t sub_cap, KpKeyBits kb) {
    unsigned long long n_front = n_cand[0], n_back = n_cand[1];
    if (n_front + n_back > cand_cap) {  // overflow: the host grows the list and reruns the pass
        if (n_front > cand_cap) n_front = cand_cap;
        n_back = cand_cap - n_front;
    }
    const unsigned long long n = n_front + n_back;
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
        uint32_t first_posting = 0xFFFFFFFFu, cnt = 0, base = 0, t = 0, zt = 0;
        size_t slice = 0;
        if (i < n) {
            const bool edge = i >= n_front;
            const uint64_t cw = edge ? cand[cand_cap - 1 - (i - n_front)] : cand[i];
            const uint64_t pos = cw >> 31;
            const int64_t word = (int64_t)(pos >> 4);
            const uint32_t x = (uint32_t)cw & KP_KMER_MASK;
            zt = (uint32_t)(cw >> 30) & 1u;
            uint32_t slot = (x * 2654435769u) >> idx.slot_shift;
            for (;;) {
                const uint2 e = idx.slots[slot];
                if (e.x == x) { first_posting = e.y; break; }
                if (e.x == 0xFFFFFFFFu) break;
                slot = (slot + 1) & idx.slot_mask;
            }
            if (first_posting != 0xFFFFFFFFu) {  // else: filter false positive
                bool ok = false;
                const int a = upper_bound_i64(b.asm_word_off, b.n_asm + 1, word) - 1;
                if (a >= 0 && a < b.n_asm) {
                    t = (uint32_t)(pos - ((uint64_t)b.asm_word_off[a] << 4));
                    ok = edge;
                    if (!edge) {
                        const int c0 = b.asm_first_ctg[a], nc = b.asm_first_ctg[a + 1] - c0;
                        const int c = upper_bound_i32(b.ctg_start + c0, nc, (int32_t)t) - 1;
                        if (c >= 0) {
                            // the clean stretch [S, E) around t: the contig, cut at the nearest N runs on either side
                            int64_t S = b.ctg_start[c0 + c], E = S + b.ctg_len[c0 + c];
                            const int r0 = b.asm_first_nrun[a], nr = b.asm_first_nrun[a + 1] - r0;
                            if (nr > 0) {
                                int lo = 0, hi = nr;  // first run whose end is > t
                                while (lo < hi) {
                                    const int mid = (lo + hi) >> 1;
                                    if (b.n_runs[2 * (r0 + mid) + 1] <= (int32_t)t) lo = mid + 1; else hi = mid;
                                }
                                if (lo < nr) E = min(E, (int64_t)b.n_runs[2 * (r0 + lo)]);  // (a run that holds t makes E <= t)
                                if (lo > 0) S = max(S, (int64_t)b.n_runs[2 * (r0 + lo - 1) + 1]);
                            }
                            ok = kp_seed_is_interior((int64_t)t, S, E);
                        }
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
            sw.src[rank] = idx.postings + first_posting + 1 + (zt ? cnt : 0u);  // the list for this contig seed's strand bit
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
                    int mode, int32_t n_ctg_total, hipStream_t stream, hipEvent_t after_scan) {
    if (b.total_words == 0) return;
    const int64_t n_units = b.total_words >> 2;
    int64_t blocks = ((n_units + SCAN_OWN - 1) / SCAN_OWN + DENSE_WAVES - 1) / DENSE_WAVES;
    if (blocks > 256 * 8) blocks = 256 * 8;  // 256 CUs x 8 blocks, wave-iteration stride beyond that
    const dim3 grid((unsigned)blocks), block(64 * DENSE_WAVES);
    if (mode == 1) hipLaunchKernelGGL((kp_scan_dense_kernel<1>), grid, block, 0, stream, b, idx, cand, n_cand, cand_cap);
    else if (mode == 2) hipLaunchKernelGGL((kp_scan_dense_kernel<2>), grid, block, 0, stream, b, idx, cand, n_cand, cand_cap);
    else hipLaunchKernelGGL((kp_scan_dense_kernel<0>), grid, block, 0, stream, b, idx, cand, n_cand, cand_cap);
    if (mode == 0 && n_ctg_total > 0)
        hipLaunchKernelGGL(kp_edge_kernel, dim3((unsigned)((2 * (int64_t)n_ctg_total + 255) / 256)), dim3(256), 0, stream, b, idx, cand, n_cand,
                           cand_cap, n_ctg_total);
    if (after_scan) (void)hipEventRecord(after_scan, stream);
    hipLaunchKernelGGL(kp_expand_kernel, dim3(256 * 8), dim3(256), 0, stream, b, idx, cand, n_cand, cand_cap,
                       anchors, sub_count, sub_cap, key_bits);
}
