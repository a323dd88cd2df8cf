// kp_prot.hip -- batched banded Smith-Waterman-Gotoh on proteins with full traceback statistics.
//
// Restates the reference's numba kernel _batched_banded_gotoh in its unseeded mode
// (src/kaptive/core/pairwise.py:395-584): BLOSUM62 via a 256x256 byte lookup (pairwise.py:343-391), gap open 11 +
// extend 1, band |i-j| <= k with k = max(20, |len1-len2|+1), cells outside the band read as M=0 / D=I=-1e9, ties:
// opening a gap beats extending it, diagonal beats D (vertical) beats I (horizontal), best<=0 restarts, the reported
// cell is the first maximum in row-major order, and matches / mismatches / gaps / start are what the reference's
// traceback loop would count.
//
// One wave per pair.  The band is indexed by b = j - i + k; cell (i, b) is computed at time t = 2i + b, so at each
// time step the cells of one parity are independent (anti-diagonal order) and only need their neighbours of the
// other parity (left: b-1, up: b+1) or their own previous value (diagonal).  Instead of storing traceback matrices,
// every state carries the statistics of the path it came from (same tie rules).
//
// Two kernels, same arithmetic:
//   * kp_protein_kernel (band <= 64 diagonals and both proteins <= 2048 residues -- every full-length gene): lane b
//     owns diagonal b; state lives in registers; the two neighbour exchanges per step are DPP wave shifts; both
//     sequences and a 32x32 BLOSUM62 are staged in LDS; no barrier inside the time loop.
//   * kp_protein_wide_kernel (wider bands: truncated or partial genes): band state in LDS (up to LDS_CELLS entries, else
//     global scratch), one barrier per time step, and only the band entries whose cell exists at that step are visited.
#include "kp_internal.h"

namespace {

constexpr int LDS_CELLS = 1024;  // band cells held in LDS by the wide-band kernel (48 KB); wider bands use global scratch
constexpr int NF = 12;          // ints per band cell: M,D,I + 3 payload words each
constexpr int NEGP = KP_PROT_NEG_INF;
constexpr int GO = KP_PROT_GAP_OPEN + KP_PROT_GAP_EXT;
constexpr int GE = KP_PROT_GAP_EXT;
constexpr int REG_MAX_LEN = 2048;

struct Pay {  // path statistics: a = matches << 16 | mismatches, g = gaps, s = start_i << 16 | start_j
    unsigned a, g, s;
};

// one-lane wave shifts; the edge lane reads 0 (bound_ctrl) and is overridden by the caller where that matters
__device__ __forceinline__ int from_lower(int v) {  // lane b <- lane b-1
    return __builtin_amdgcn_mov_dpp(v, 0x138 /*wave_shr:1*/, 0xf, 0xf, true);
}
__device__ __forceinline__ int from_upper(int v) {  // lane b <- lane b+1
    return __builtin_amdgcn_mov_dpp(v, 0x130 /*wave_shl:1*/, 0xf, 0xf, true);
}

struct Result {
    int best, bi, bj;
    Pay bp;
};

// ---- register path ------------------------------------------------------------------------------------------------------
// s_seq1/s_seq2: residues as (raw byte << 8 | BLOSUM index 0..24, 32 = outside the alphabet); s_mat: 32x32 scores.
__device__ __forceinline__ Result protein_pair_registers(const uint16_t *s_seq1, const uint16_t *s_seq2,
                                                         const int8_t *s_mat, int len1, int len2, int k, int lane) {
    const int nb = 2 * k + 1, b = lane;
    int m = 0, dv = NEGP, iv = NEGP;
    Pay pm{0, 0, 0}, pd{0, 0, 0}, pi{0, 0, 0};
    Result r{0, 0, 0, Pay{0, 0, 0}};
    const int t_last = 2 * len1 + 2 * k;
    for (int tm = 2; tm <= t_last; ++tm) {
        // neighbours of the other parity: up = lane b+1 (row i-1), left = lane b-1 (row i)
        int um = from_upper(m), ud = from_upper(dv);
        Pay upm{(unsigned)from_upper((int)pm.a), (unsigned)from_upper((int)pm.g), (unsigned)from_upper((int)pm.s)};
        Pay upd{(unsigned)from_upper((int)pd.a), (unsigned)from_upper((int)pd.g), (unsigned)from_upper((int)pd.s)};
        int lm = from_lower(m), li = from_lower(iv);
        Pay lpm{(unsigned)from_lower((int)pm.a), (unsigned)from_lower((int)pm.g), (unsigned)from_lower((int)pm.s)};
        Pay lpi{(unsigned)from_lower((int)pi.a), (unsigned)from_lower((int)pi.g), (unsigned)from_lower((int)pi.s)};
        if (b + 1 >= nb) { um = 0; ud = NEGP; }
        if (b == 0) { lm = 0; li = NEGP; }
        const int i2 = tm - b, i = i2 >> 1, j = i + b - k;
        const bool mine = ((i2 & 1) == 0) && b < nb;  // this lane's parity
        const bool in = mine && i2 >= 2 && i <= len1 && j >= 1 && j <= len2;
        int nm = 0, ndv = NEGP, niv = NEGP;
        Pay npm{0, 0, 0}, npd{0, 0, 0}, npi{0, 0, 0};
        if (in) {
            if (um == 0) upm = Pay{0, 0, ((unsigned)(i - 1) << 16) | (unsigned)j};
            const int d_open = um - GO, d_ext = ud - GE;
            if (d_open >= d_ext) { ndv = d_open; npd = upm; } else { ndv = d_ext; npd = upd; }
            npd.g += 1;
            if (lm == 0) lpm = Pay{0, 0, ((unsigned)i << 16) | (unsigned)(j - 1)};
            const int i_open = lm - GO, i_ext = li - GE;
            if (i_open >= i_ext) { niv = i_open; npi = lpm; } else { niv = i_ext; npi = lpi; }
            npi.g += 1;
            Pay dp = pm;
            if (m == 0) dp = Pay{0, 0, ((unsigned)(i - 1) << 16) | (unsigned)(j - 1)};
            const unsigned c1 = s_seq1[i - 1], c2 = s_seq2[j - 1];
            const unsigned x1 = c1 & 255u, x2 = c2 & 255u;
            const int sc = (x1 | x2) >= 32u ? KP_PROT_FILL : (int)s_mat[x1 * 32 + x2];  // 32 marks a byte outside the alphabet
            int bv = m + sc;
            npm = dp;
            npm.a += ((c1 >> 8) == (c2 >> 8)) ? 0x10000u : 1u;
            if (ndv > bv) { bv = ndv; npm = npd; }
            if (niv > bv) { bv = niv; npm = npi; }
            if (bv > 0) {
                nm = bv;
                if (nm > r.best) { r.best = nm; r.bi = i; r.bj = j; r.bp = npm; }  // rows only grow within a lane
            }
        }
        if (mine) {  // cells outside the matrix take boundary values, exactly as the band array of the general path
            m = nm; dv = ndv; iv = niv; pm = npm; pd = npd; pi = npi;
        }
    }
    return r;
}

// ---- general path ---------------------------------------------------------------------------------------------------------
template <class BandPtr>
__device__ __forceinline__ Result protein_pair_general(BandPtr st, int cap, const uint8_t *__restrict__ s1,
                                                       const uint8_t *__restrict__ s2, int len1, int len2, int k,
                                                       const int8_t *__restrict__ blosum, int lane) {
    const int nb = 2 * k + 1;
#define F(field, b) st[(field) * cap + (b)]
    __syncthreads();
    for (int b = lane; b < nb; b += 64) {
        F(0, b) = 0; F(1, b) = NEGP; F(2, b) = NEGP;
        for (int f = 3; f < NF; ++f) F(f, b) = 0;
    }
    Result r{0, 0, 0, Pay{0, 0, 0}};
    const int t_last = 2 * len1 + 2 * k;
    for (int tm = 2; tm <= t_last; ++tm) {
        __syncthreads();
        // band entries whose cell (i, j) lies inside the matrix at this time step; entries outside keep what they hold
        // (the initial boundary, or a cell no later cell reads), so nothing else needs to be touched
        int b_lo = max(0, max(tm - 2 * len1, 2 * k + 2 - tm));
        const int b_hi = min(nb - 1, min(tm - 2, 2 * k + 2 * len2 - tm));
        b_lo += (b_lo ^ tm) & 1;
        for (int b = b_lo + 2 * lane; b <= b_hi; b += 128) {
            const int i2 = tm - b;  // = 2i
            const int i = i2 >> 1, j = i + b - k;
            int m = 0, dv = NEGP, iv = NEGP;
            Pay pm{0, 0, 0}, pd{0, 0, 0}, pi{0, 0, 0};
            {
                int um = 0, ud = NEGP; Pay upm{0, 0, 0}, upd{0, 0, 0};
                if (b + 1 < nb) {
                    um = F(0, b + 1); ud = F(1, b + 1);
                    upm = Pay{(unsigned)F(3, b + 1), (unsigned)F(4, b + 1), (unsigned)F(5, b + 1)};
                    upd = Pay{(unsigned)F(6, b + 1), (unsigned)F(7, b + 1), (unsigned)F(8, b + 1)};
                }
                if (um == 0) upm = Pay{0, 0, ((unsigned)(i - 1) << 16) | (unsigned)j};  // path would start there
                const int d_open = um - GO, d_ext = ud - GE;
                if (d_open >= d_ext) { dv = d_open; pd = upm; } else { dv = d_ext; pd = upd; }
                pd.g += 1;
                int lm = 0, li = NEGP; Pay lpm{0, 0, 0}, lpi{0, 0, 0};
                if (b >= 1) {
                    lm = F(0, b - 1); li = F(2, b - 1);
                    lpm = Pay{(unsigned)F(3, b - 1), (unsigned)F(4, b - 1), (unsigned)F(5, b - 1)};
                    lpi = Pay{(unsigned)F(9, b - 1), (unsigned)F(10, b - 1), (unsigned)F(11, b - 1)};
                }
                if (lm == 0) lpm = Pay{0, 0, ((unsigned)i << 16) | (unsigned)(j - 1)};
                const int i_open = lm - GO, i_ext = li - GE;
                if (i_open >= i_ext) { iv = i_open; pi = lpm; } else { iv = i_ext; pi = lpi; }
                pi.g += 1;
                const int dm = F(0, b);
                Pay dp{(unsigned)F(3, b), (unsigned)F(4, b), (unsigned)F(5, b)};
                if (dm == 0) dp = Pay{0, 0, ((unsigned)(i - 1) << 16) | (unsigned)(j - 1)};
                const uint8_t c1 = s1[i - 1], c2 = s2[j - 1];
                int bv = dm + (int)blosum[(int)c1 * 256 + c2];
                pm = dp;
                pm.a += (c1 == c2) ? 0x10000u : 1u;
                if (dv > bv) { bv = dv; pm = pd; }
                if (iv > bv) { bv = iv; pm = pi; }
                if (bv > 0) {
                    m = bv;
                    if (m > r.best || (m == r.best && (i < r.bi || (i == r.bi && j < r.bj)))) {
                        r.best = m; r.bi = i; r.bj = j; r.bp = pm;
                    }
                }
            }
            F(0, b) = m; F(1, b) = dv; F(2, b) = iv;
            F(3, b) = (int)pm.a; F(4, b) = (int)pm.g; F(5, b) = (int)pm.s;
            F(6, b) = (int)pd.a; F(7, b) = (int)pd.g; F(8, b) = (int)pd.s;
            F(9, b) = (int)pi.a; F(10, b) = (int)pi.g; F(11, b) = (int)pi.s;
        }
    }
#undef F
    return r;
}

__device__ __forceinline__ bool fits_registers(int len1, int len2, int nb) {
    return nb <= 64 && len1 <= REG_MAX_LEN && len2 <= REG_MAX_LEN;
}

__device__ __forceinline__ void store_result(Result r, int lane, int32_t *__restrict__ o) {
    // wave reduction: max score, then smallest i, then smallest j
    int best = r.best, bi = r.bi, bj = r.bj;
    Pay bp = r.bp;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const int b2 = __shfl_xor(best, off), i2 = __shfl_xor(bi, off), j2 = __shfl_xor(bj, off);
        const unsigned a2 = __shfl_xor(bp.a, off), g2 = __shfl_xor(bp.g, off), s2v = __shfl_xor(bp.s, off);
        // lanes that never saw a positive cell carry best = 0 and must lose against any positive score
        const bool take = b2 > best || (b2 == best && b2 > 0 && (i2 < bi || (i2 == bi && j2 < bj)));
        if (take) { best = b2; bi = i2; bj = j2; bp = Pay{a2, g2, s2v}; }
    }
    if (lane == 0) {
        if (best > 0) {
            o[0] = best; o[1] = (int)(bp.a >> 16); o[2] = (int)(bp.a & 0xFFFFu); o[3] = (int)bp.g;
            o[4] = (int)(bp.s >> 16); o[5] = bi; o[6] = (int)(bp.s & 0xFFFFu); o[7] = bj;
        } else {
            for (int x = 0; x < 8; ++x) o[x] = 0;
        }
    }
}

// pairs whose band fits one diagonal per lane (and the empty ones)
__global__ __launch_bounds__(64) void kp_protein_kernel(const uint8_t *__restrict__ q, const int32_t *__restrict__ q_off,
                                                        const int32_t *__restrict__ q_len,
                                                        const uint8_t *__restrict__ t, const int32_t *__restrict__ t_off,
                                                        const int32_t *__restrict__ t_len, int32_t n_host,
                                                        const int32_t *__restrict__ n_dev,
                                                        const int8_t *__restrict__ blosum, int32_t *__restrict__ out8) {
    __shared__ uint16_t s_seq1[REG_MAX_LEN], s_seq2[REG_MAX_LEN];
    __shared__ int8_t s_mat[32 * 32];
    __shared__ uint8_t s_idx[256];
    const int lane = threadIdx.x;
    // compact substitution table: index of each byte in ARNDCQEGHILKMFPSTWYVBJZX*, 32 for everything else
    for (int c = lane; c < 256; c += 64) s_idx[c] = 32;
    __syncthreads();
    if (lane < 25) s_idx[(uint8_t)"ARNDCQEGHILKMFPSTWYVBJZX*"[lane]] = (uint8_t)lane;
    __syncthreads();
    for (int x = lane; x < 32 * 32; x += 64) {
        const int a = x >> 5, c = x & 31;
        s_mat[x] = (a < 25 && c < 25) ? blosum[(int)(uint8_t)"ARNDCQEGHILKMFPSTWYVBJZX*"[a] * 256 +
                                               (int)(uint8_t)"ARNDCQEGHILKMFPSTWYVBJZX*"[c]]
                                      : (int8_t)KP_PROT_FILL;
    }
    const int n = n_dev ? *n_dev : n_host;
    for (int p = blockIdx.x; p < n; p += gridDim.x) {
        const int len1 = q_len[p], len2 = t_len[p];
        if (len1 == 0 || len2 == 0) {  // nothing to align
            if (lane < 8) out8[8 * (size_t)p + lane] = 0;
            continue;
        }
        int d = len1 - len2;
        if (d < 0) d = -d;
        const int k = max(KP_PROT_K, d + 1);
        if (!fits_registers(len1, len2, 2 * k + 1)) continue;  // kp_protein_wide_kernel's
        const uint8_t *s1 = q + q_off[p], *s2 = t + t_off[p];
        __syncthreads();
        for (int x = lane; x < len1; x += 64) s_seq1[x] = (uint16_t)(((unsigned)s1[x] << 8) | s_idx[s1[x]]);
        for (int x = lane; x < len2; x += 64) s_seq2[x] = (uint16_t)(((unsigned)s2[x] << 8) | s_idx[s2[x]]);
        __syncthreads();
        store_result(protein_pair_registers(s_seq1, s_seq2, s_mat, len1, len2, k, lane), lane, out8 + 8 * (size_t)p);
    }
}

// the rest: wide bands (truncated / partial genes) and very long proteins
__global__ __launch_bounds__(64) void kp_protein_wide_kernel(const uint8_t *__restrict__ q, const int32_t *__restrict__ q_off,
                                                             const int32_t *__restrict__ q_len,
                                                             const uint8_t *__restrict__ t, const int32_t *__restrict__ t_off,
                                                             const int32_t *__restrict__ t_len, int32_t n_host,
                                                             const int32_t *__restrict__ n_dev,
                                                             const int8_t *__restrict__ blosum, int32_t *__restrict__ out8,
                                                             int32_t *__restrict__ scratch, size_t scratch_ints_per_block) {
    __shared__ int s_band[LDS_CELLS * NF];
    const int lane = threadIdx.x;
    const int n = n_dev ? *n_dev : n_host;
    for (int p = blockIdx.x; p < n; p += gridDim.x) {
        const int len1 = q_len[p], len2 = t_len[p];
        if (len1 == 0 || len2 == 0) continue;
        int d = len1 - len2;
        if (d < 0) d = -d;
        const int k = max(KP_PROT_K, d + 1);
        const int nb = 2 * k + 1;
        if (fits_registers(len1, len2, nb)) continue;
        const uint8_t *s1 = q + q_off[p], *s2 = t + t_off[p];
        Result r;
        if (nb <= LDS_CELLS) r = protein_pair_general(s_band, LDS_CELLS, s1, s2, len1, len2, k, blosum, lane);
        else r = protein_pair_general(scratch + (size_t)blockIdx.x * scratch_ints_per_block, nb, s1, s2, len1, len2, k,
                                      blosum, lane);
        store_result(r, lane, out8 + 8 * (size_t)p);
    }
}

}  // namespace

// n pairs; when n_dev is not null the pair count is read from device memory (n is then only an upper bound)
void kp_launch_protein(const uint8_t *q, const int32_t *q_off, const int32_t *q_len, const uint8_t *t,
                       const int32_t *t_off, const int32_t *t_len, int32_t n, const int32_t *n_dev, const int8_t *blosum,
                       int32_t *out8, int32_t *scratch, size_t scratch_ints_per_block, int n_blocks, hipStream_t stream,
                       hipStream_t aux, hipEvent_t fork, hipEvent_t join) {
    if (n == 0) return;
    // The wide-band kernel is a handful of long-running waves (its duration is one pair's time-step chain), the
    // register kernel fills the chip: with a second stream they run side by side, joined before `stream` goes on.
    hipStream_t wide = stream;
    if (aux) {
        (void)hipEventRecord(fork, stream);
        (void)hipStreamWaitEvent(aux, fork, 0);
        wide = aux;
    }
    hipLaunchKernelGGL(kp_protein_wide_kernel, dim3(n_blocks), dim3(64), 0, wide, q, q_off, q_len, t, t_off, t_len, n,
                       n_dev, blosum, out8, scratch, scratch_ints_per_block);
    hipLaunchKernelGGL(kp_protein_kernel, dim3(n_blocks), dim3(64), 0, stream, q, q_off, q_len, t, t_off, t_len, n, n_dev,
                       blosum, out8);
    if (aux) {
        (void)hipEventRecord(join, aux);
        (void)hipStreamWaitEvent(stream, join, 0);
    }
}
