// kp_prot.hip -- batched banded Smith-Waterman-Gotoh on proteins with full traceback statistics.
//
// Restates the reference's numba kernel _batched_banded_gotoh in its unseeded mode
// (src/kaptive/core/pairwise.py:395-584): BLOSUM62 via a 256x256 byte lookup (pairwise.py:343-391), gap open 11 +
// extend 1, band |i-j| <= k with k = max(20, |len1-len2|+1), cells outside the band read as M=0 / D=I=-1e9, ties:
// opening a gap beats extending it, diagonal beats D (vertical) beats I (horizontal), best<=0 restarts, the reported
// cell is the first maximum in row-major order, and matches / mismatches / gaps / start are what the reference's
// traceback loop would count.
//
// One wave per pair.  The band is held as one array indexed by b = j - i + k; cell (i, b) is computed at time
// t = 2i + b, so at each time step the cells of one parity are independent (anti-diagonal order) and only read
// entries of the other parity (left: b-1, up: b+1) or their own previous value (diagonal).  Instead of storing
// traceback matrices, every state carries the statistics of the path it came from (same tie rules), so the answer is
// available when the fill ends.  The band lives in LDS when it fits, else in a per-block global scratch area.
#include "kp_internal.h"

namespace {

constexpr int LDS_CELLS = 160;  // band cells held in LDS (covers k <= 79); wider bands use global scratch
constexpr int NF = 12;          // ints per band cell: M,D,I + 3 payload words each
constexpr int NEGP = KP_PROT_NEG_INF;
constexpr int GO = KP_PROT_GAP_OPEN + KP_PROT_GAP_EXT;
constexpr int GE = KP_PROT_GAP_EXT;

struct Pay {  // path statistics: a = matches << 16 | mismatches, g = gaps, s = start_i << 16 | start_j
    unsigned a, g, s;
};

__global__ __launch_bounds__(64) void kp_protein_kernel(const uint8_t *__restrict__ q, const int32_t *__restrict__ q_off,
                                                        const int32_t *__restrict__ q_len,
                                                        const uint8_t *__restrict__ t, const int32_t *__restrict__ t_off,
                                                        const int32_t *__restrict__ t_len, int32_t n,
                                                        const int8_t *__restrict__ blosum, int32_t *__restrict__ out8,
                                                        int32_t *__restrict__ scratch, size_t scratch_ints_per_block) {
    __shared__ int s_band[LDS_CELLS * NF];
    const int lane = threadIdx.x;
    for (int p = blockIdx.x; p < n; p += gridDim.x) {
        const int len1 = q_len[p], len2 = t_len[p];
        if (len1 == 0 || len2 == 0) {  // nothing to align (also the empty slots of the batched reduction)
            if (lane < 8) out8[8 * (size_t)p + lane] = 0;
            continue;
        }
        const uint8_t *s1 = q + q_off[p], *s2 = t + t_off[p];
        int d = len1 - len2;
        if (d < 0) d = -d;
        const int k = max(KP_PROT_K, d + 1);
        const int nb = 2 * k + 1;
        int *st = (nb <= LDS_CELLS) ? s_band : (scratch + (size_t)blockIdx.x * scratch_ints_per_block);
        const int cap = (nb <= LDS_CELLS) ? LDS_CELLS : nb;
#define F(field, b) st[(field) * cap + (b)]
        __syncthreads();
        for (int b = lane; b < nb; b += 64) {
            F(0, b) = 0; F(1, b) = NEGP; F(2, b) = NEGP;
            for (int f = 3; f < NF; ++f) F(f, b) = 0;
        }
        int best = 0, bi = 0, bj = 0;
        Pay bp{0, 0, 0};
        const int t_last = 2 * len1 + 2 * k;
        for (int tm = 2; tm <= t_last && len1 > 0 && len2 > 0; ++tm) {
            __syncthreads();
            for (int b = (tm & 1) + 2 * lane; b < nb; b += 128) {
                const int i2 = tm - b;  // = 2i
                const int i = i2 >> 1, j = i + b - k;
                int m = 0, dv = NEGP, iv = NEGP;
                Pay pm{0, 0, 0}, pd{0, 0, 0}, pi{0, 0, 0};
                if (i2 >= 2 && i <= len1 && j >= 1 && j <= len2) {
                    // vertical gap D from (i-1, j) = entry b+1
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
                    // horizontal gap I from (i, j-1) = entry b-1
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
                    // diagonal from (i-1, j-1) = this entry's previous value
                    const int dm = F(0, b);
                    Pay dp{(unsigned)F(3, b), (unsigned)F(4, b), (unsigned)F(5, b)};
                    if (dm == 0) dp = Pay{0, 0, ((unsigned)(i - 1) << 16) | (unsigned)(j - 1)};
                    const uint8_t c1 = s1[i - 1], c2 = s2[j - 1];
                    int bv = dm + (int)blosum[(int)c1 * 256 + c2];
                    pm = dp;
                    pm.a += (c1 == c2) ? 0x10000u : 1u;
                    if (dv > bv) { bv = dv; pm = pd; }
                    if (iv > bv) { bv = iv; pm = pi; }
                    if (bv <= 0) m = 0;
                    else {
                        m = bv;
                        if (m > best || (m == best && (i < bi || (i == bi && j < bj)))) { best = m; bi = i; bj = j; bp = pm; }
                    }
                }
                F(0, b) = m; F(1, b) = dv; F(2, b) = iv;
                F(3, b) = (int)pm.a; F(4, b) = (int)pm.g; F(5, b) = (int)pm.s;
                F(6, b) = (int)pd.a; F(7, b) = (int)pd.g; F(8, b) = (int)pd.s;
                F(9, b) = (int)pi.a; F(10, b) = (int)pi.g; F(11, b) = (int)pi.s;
            }
        }
#undef F
        // wave reduction: max score, then smallest i, then smallest j
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int b2 = __shfl_xor(best, o), i2 = __shfl_xor(bi, o), j2 = __shfl_xor(bj, o);
            const unsigned a2 = __shfl_xor(bp.a, o), g2 = __shfl_xor(bp.g, o), s2v = __shfl_xor(bp.s, o);
            // lanes that never saw a positive cell carry best = 0 and must lose against any positive score
            const bool take = b2 > best || (b2 == best && b2 > 0 && (i2 < bi || (i2 == bi && j2 < bj)));
            if (take) { best = b2; bi = i2; bj = j2; bp = Pay{a2, g2, s2v}; }
        }
        if (lane == 0) {
            int32_t *o = out8 + 8 * (size_t)p;
            if (best > 0) {
                o[0] = best; o[1] = (int)(bp.a >> 16); o[2] = (int)(bp.a & 0xFFFFu); o[3] = (int)bp.g;
                o[4] = (int)(bp.s >> 16); o[5] = bi; o[6] = (int)(bp.s & 0xFFFFu); o[7] = bj;
            } else {
                for (int x = 0; x < 8; ++x) o[x] = 0;
            }
        }
    }
}

}  // namespace

void kp_launch_protein(const uint8_t *q, const int32_t *q_off, const int32_t *q_len, const uint8_t *t,
                       const int32_t *t_off, const int32_t *t_len, int32_t n, const int8_t *blosum, int32_t *out8,
                       int32_t *scratch, size_t scratch_ints_per_block, int n_blocks, hipStream_t stream) {
    if (n == 0) return;
    hipLaunchKernelGGL(kp_protein_kernel, dim3(n_blocks), dim3(64), 0, stream, q, q_off, q_len, t, t_off, t_len, n,
                       blosum, out8, scratch, scratch_ints_per_block);
}
