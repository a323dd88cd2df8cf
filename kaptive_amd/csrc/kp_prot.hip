// kp_prot.hip -- batched banded Smith-Waterman-Gotoh on proteins with full traceback statistics.
//
// Restates the reference's numba kernel _batched_banded_gotoh (src/kaptive/core/pairwise.py:395-584), both modes:
// unseeded -- band |i-j| <= k with k = max(20, |len1-len2|+1) -- and seeded (pairwise.py:449-451; used by
// compare.LocusComparator) -- band |j - (i - offset)| <= k with the caller's k and one diagonal offset per pair ("shift"
// below; 0 in the unseeded mode).  BLOSUM62 via a 256x256 byte lookup (pairwise.py:343-391), gap open 11 +
// extend 1, cells outside the band read as M=0 / D=I=-1e9, ties:
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
//   * kp_protein_kernel (band <= 64 diagonals and both proteins <= 768 residues -- every full-length gene): 16 lanes
//     per pair, four adjacent diagonals per lane, four pairs per wave; state lives in registers; the two neighbour
//     exchanges per step are DPP row shifts; sequences and a 32x32 BLOSUM62 are staged in LDS; no barrier inside the
//     step loop.
//   * kp_protein_wide_kernel (wider bands: truncated or partial genes, or very long proteins): strips of 64 rows, lane =
//     row, cells of a row visited left to right one step behind the row above (see protein_pair_strips); pairs are
//     taken off a shared counter.
#include <algorithm>

#include "kp_internal.h"

namespace {

constexpr int NEGP = KP_PROT_NEG_INF;
constexpr int GO = KP_PROT_GAP_OPEN + KP_PROT_GAP_EXT;
constexpr int GE = KP_PROT_GAP_EXT;
constexpr int REG_MAX_LEN = 768;  // residues per sequence the register kernel stages (four pairs per block: 12 KB of LDS)

// Path statistics a DP state carries: a = matches << 16 | mismatches, g = gaps that consumed a query row << 16 | gaps that
// consumed a target column.  The start of the path is not carried (round 2 did: a third register through every select):
// a path that ends in (bi, bj) has consumed matches + mismatches + row gaps rows and matches + mismatches + column gaps
// columns, so start_i = bi - (m + mm + g_row) and start_j = bj - (m + mm + g_col) -- exactly what the reference's traceback
// would reach (it stops at the first cell whose M is 0, which is where the counts were reset here).
struct Pay {
    unsigned a, g;
};
constexpr unsigned GAP_ROW = 0x10000u, GAP_COL = 1u;

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
// Band of up to 64 diagonals on 16 lanes: lane l owns the four adjacent diagonals b = 4l .. 4l+3 (cells c = 0..3) and
// works on row i = m - l + 1 at step m, so a wave aligns four pairs side by side.  With that skew the first cell's left
// neighbour is lane l-1's last cell of the previous step, the last cell's upper neighbour is lane l+1's first cell of
// the same step, everything else is one of the lane's own registers (DPP row shifts; a group of 16 lanes is a DPP row).
// s_seq1/s_seq2: residues as (raw byte << 8 | BLOSUM index 0..24, 31 = outside the alphabet); s_mat: 32x32 scores.
constexpr int QP = 16;  // lanes per pair

__device__ __forceinline__ int row_lower(int v) {  // lane b <- lane b-1 within the group's DPP row
    return __builtin_amdgcn_mov_dpp(v, 0x111 /*row_shr:1*/, 0xf, 0xf, true);
}
__device__ __forceinline__ int row_upper(int v) {  // lane b <- lane b+1
    return __builtin_amdgcn_mov_dpp(v, 0x101 /*row_shl:1*/, 0xf, 0xf, true);
}
__device__ __forceinline__ Pay row_lower(Pay p) {
    return Pay{(unsigned)row_lower((int)p.a), (unsigned)row_lower((int)p.g)};
}
__device__ __forceinline__ Pay row_upper(Pay p) {
    return Pay{(unsigned)row_upper((int)p.a), (unsigned)row_upper((int)p.g)};
}

struct PCell {
    int m, dv, iv;
    Pay pm, pd, pi;
};

__device__ __forceinline__ Pay pick(bool c, const Pay &x, const Pay &y) {
    return Pay{c ? x.a : y.a, c ? x.g : y.g};
}

// one cell (i, j); `left` = (i, j-1): its M, I and their payloads; `up` = (i-1, j): its M, D and their payloads; the
// diagonal neighbour (i-1, j-1) comes in as c.m / c.pm.  Same statements as the reference's kernel (see top), written
// as selects: every lane executes the same instructions whatever its cell decides.
template <bool ROW_MAJOR = true>  // false: the lane does not visit its cells in row-major order and ties are settled by (i, j)
__device__ __forceinline__ void prot_cell(PCell &c, Result &r, bool in, int i, int j, unsigned c1, unsigned c2,
                                          const int8_t *s_mat, int lm, int li, Pay lpm, Pay lpi, int um, int ud, Pay upm,
                                          Pay upd) {
    // D: gap arriving from above.  (A path starts at a cell whose M is 0: such a cell keeps the payload of its M state at
    // zero -- the last statement below --, so none of its three readers has to ask.)
    const int d_open = um - GO, d_ext = ud - GE;
    int ndv = max(d_open, d_ext);
    Pay npd = pick(d_open >= d_ext, upm, upd);
    npd.g += GAP_ROW;
    // I: gap arriving from the left
    const int i_open = lm - GO, i_ext = li - GE;
    int niv = max(i_open, i_ext);
    Pay npi = pick(i_open >= i_ext, lpm, lpi);
    npi.g += GAP_COL;
    // M: diagonal first, then D, then I, each only if strictly better
    Pay npm = c.pm;
    const unsigned x1 = c1 & 255u, x2 = c2 & 255u;
    const int sc = (int)s_mat[x1 * 32 + x2];  // (index 31 = a byte outside the alphabet: its row and column hold KP_PROT_FILL)
    int bv = c.m + sc;
    npm.a += ((c1 >> 8) == (c2 >> 8)) ? 0x10000u : 1u;
    const bool take_d = ndv > bv;
    bv = max(bv, ndv);
    npm = pick(take_d, npd, npm);
    const bool take_i = niv > bv;
    bv = max(bv, niv);
    npm = pick(take_i, npi, npm);
    // cells outside the matrix or the band take boundary values, exactly as the reference's band array holds them
    const int nm = in ? max(bv, 0) : 0;
    // the first maximum in row-major order: a lane's cells come in that order (strict rise), or the tie is looked at
    if (nm > r.best || (!ROW_MAJOR && nm == r.best && nm > 0 && (i < r.bi || (i == r.bi && j < r.bj)))) {
        r.best = nm; r.bi = i; r.bj = j; r.bp = npm;
    }
    c.m = nm; c.dv = in ? ndv : NEGP; c.iv = in ? niv : NEGP;
    c.pm = pick(nm == 0, Pay{0, 0}, npm); c.pd = npd; c.pi = npi;
}

// NC adjacent diagonals per lane: 4 covers any band of up to 64 diagonals; 3 covers 48, which is enough for the default band
// (k = 20: 41 diagonals) of every pair whose proteins differ by at most 22 residues in length -- nearly all of them -- and
// costs a quarter less per step.
// W lanes per pair: 16 (four pairs per wave, neighbours by DPP row shifts) or 64 (one pair per wave, wave shifts: bands of
// up to 64 * NC diagonals -- the truncated genes of kp_protein_wide_kernel, whose strips cost several times as much per cell).
template <int W>
__device__ __forceinline__ int pair_lower(int v) { return W == 16 ? row_lower(v) : from_lower(v); }
template <int W>
__device__ __forceinline__ int pair_upper(int v) { return W == 16 ? row_upper(v) : from_upper(v); }
template <int W>
__device__ __forceinline__ Pay pair_lower(Pay p) { return Pay{(unsigned)pair_lower<W>((int)p.a), (unsigned)pair_lower<W>((int)p.g)}; }
template <int W>
__device__ __forceinline__ Pay pair_upper(Pay p) { return Pay{(unsigned)pair_upper<W>((int)p.a), (unsigned)pair_upper<W>((int)p.g)}; }

template <int NC, int W = QP>
__device__ __forceinline__ Result protein_quad_registers(const uint16_t *s_seq1, const uint16_t *s_seq2,
                                                         const int8_t *s_mat, int len1, int len2, int k, int shift,
                                                         int l, int max_steps) {
    const int nb = 2 * k + 1;
    PCell cell[NC];
    bool band_ok[NC];  // the lane's diagonals that exist in this pair's band
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        cell[c] = PCell{0, NEGP, NEGP, Pay{0, 0}, Pay{0, 0}, Pay{0, 0}};
        band_ok[c] = NC * l + c < nb;
    }
    Result r{0, 0, 0, Pay{0, 0}};
    for (int m = 0; m < max_steps; ++m) {  // max_steps is wave-uniform (the longest pair of the quad)
        const int i = m - l + 1, j0 = i - shift + NC * l - k;  // column of the lane's first cell
        const bool row_ok = i >= 1 && i <= len1;
        const unsigned c1 = row_ok ? s_seq1[i - 1] : 0u;
        unsigned c2[NC];
        bool in[NC];
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const bool col_ok = (unsigned)(j0 + c - 1) < (unsigned)len2;  // 1 <= j <= len2
            c2[c] = col_ok ? s_seq2[j0 + c - 1] : 0u;
            in[c] = row_ok && col_ok && band_ok[c];
        }
        // first cell: left neighbour = lane l-1's last cell of the previous step
        int lm = pair_lower<W>(cell[NC - 1].m), li = pair_lower<W>(cell[NC - 1].iv);
        Pay lpm = pair_lower<W>(cell[NC - 1].pm), lpi = pair_lower<W>(cell[NC - 1].pi);
        if (l == 0) { lm = 0; li = NEGP; }
        prot_cell(cell[0], r, in[0], i, j0, c1, c2[0], s_mat, lm, li, lpm, lpi, cell[1].m, cell[1].dv, cell[1].pm, cell[1].pd);
#pragma unroll
        for (int c = 1; c < NC - 1; ++c)  // inner cells: left = the cell just computed, up = the next cell's previous value
            prot_cell(cell[c], r, in[c], i, j0 + c, c1, c2[c], s_mat, cell[c - 1].m, cell[c - 1].iv, cell[c - 1].pm, cell[c - 1].pi,
                      cell[c + 1].m, cell[c + 1].dv, cell[c + 1].pm, cell[c + 1].pd);
        // last cell: upper neighbour = lane l+1's first cell of this step
        int um = pair_upper<W>(cell[0].m), ud = pair_upper<W>(cell[0].dv);
        Pay upm = pair_upper<W>(cell[0].pm), upd = pair_upper<W>(cell[0].pd);
        if (l == W - 1) { um = 0; ud = NEGP; }
        prot_cell(cell[NC - 1], r, in[NC - 1], i, j0 + NC - 1, c1, c2[NC - 1], s_mat, cell[NC - 2].m, cell[NC - 2].iv,
                  cell[NC - 2].pm, cell[NC - 2].pi, um, ud, upm, upd);
    }
    return r;
}

// ---- whole matrix in one wave, NR adjacent ROWS per lane --------------------------------------------------------------------
// For the pairs whose band is wider than the diagonal form above holds -- a fragment of at most 64 * NR residues against
// a much longer protein: the band then covers most of the matrix anyway.  Lane l owns rows NR * l + 1 .. NR * l + NR and
// walks the columns one step behind lane l - 1 (column j = t - l + 1 at step t): the upper neighbour of its first row is
// lane l - 1's last row of the previous step (one DPP shift), every other neighbour is one of its own registers.  One
// strip of 64 * NR rows, len2 + 63 steps -- the row strips take (len1 / 64) x (window + 63) steps of a costlier kind.
template <int NR>
__device__ __forceinline__ Result protein_wave_rows(const uint16_t *s_seq1, const uint16_t *s_seq2, const int8_t *s_mat,
                                                    int len1, int len2, int k, int shift, int lane) {
    PCell cell[NR];  // (i, j - 1) of the lane's rows
    unsigned c1[NR];
#pragma unroll
    for (int q = 0; q < NR; ++q) {
        cell[q] = PCell{0, NEGP, NEGP, Pay{0, 0}, Pay{0, 0}, Pay{0, 0}};
        const int i = NR * lane + q + 1;
        c1[q] = i <= len1 ? s_seq1[i - 1] : 0u;
    }
    int dm = 0;  // M of (first row - 1, j - 1) and its payload: what came from lane l - 1 a step earlier
    Pay dpm{0, 0};
    Result r{0, 0, 0, Pay{0, 0}};
    const int n_steps = len2 + 63;
    for (int t = 0; t < n_steps; ++t) {
        const int j = t - lane + 1;
        const bool col_ok = j >= 1 && j <= len2;
        const unsigned c2 = col_ok ? s_seq2[j - 1] : 0u;
        // upper neighbour of the lane's first row: lane l - 1's last row at this column (it was there in the previous step)
        int um = from_lower(cell[NR - 1].m), ud = from_lower(cell[NR - 1].dv);
        Pay upm{(unsigned)from_lower((int)cell[NR - 1].pm.a), (unsigned)from_lower((int)cell[NR - 1].pm.g)};
        Pay upd{(unsigned)from_lower((int)cell[NR - 1].pd.a), (unsigned)from_lower((int)cell[NR - 1].pd.g)};
        if (lane == 0) { um = 0; ud = NEGP; upm = Pay{0, 0}; upd = Pay{0, 0}; }
        int next_dm = um;
        Pay next_dpm = upm;
#pragma unroll
        for (int q = 0; q < NR; ++q) {
            const int i = NR * lane + q + 1;
            int db = i - shift - j;
            if (db < 0) db = -db;
            const bool in = col_ok && i <= len1 && db <= k;
            const PCell left = cell[q];  // (i, j - 1): the left neighbour, and the diagonal one of the row below
            PCell c;
            c.m = dm; c.pm = dpm;  // diagonal neighbour (i - 1, j - 1)
            prot_cell<false>(c, r, in, i, j, c1[q], c2, s_mat, left.m, left.iv, left.pm, left.pi, um, ud, upm, upd);
            cell[q] = c;
            dm = left.m; dpm = left.pm;                          // for the row below
            um = c.m; ud = c.dv; upm = c.pm; upd = c.pd;         // (i, j) is the upper neighbour of the row below
        }
        dm = next_dm; dpm = next_dpm;  // (first row - 1, j) is the diagonal neighbour at the next column
    }
    return r;
}

// ---- row-strip path -------------------------------------------------------------------------------------------------------
// Any band, any length.  The matrix is cut into strips of 64 rows; lane l owns row i0 + l of the strip and walks it left
// to right, one step behind lane l-1 (cell (i, j) at step t = j - j_lo + l).  With that skew the upper neighbour
// (i-1, j) is what lane l-1 produced in the previous step (one DPP shift of its M / D and their payloads), the diagonal
// one is the upper neighbour fetched a step earlier, and the left one is the lane's own previous cell.  Only columns that
// can be inside the band for some row of the strip are visited (j_lo .. j_hi); cells outside the band or the matrix
// produce the boundary values (M = 0, D = I = -inf), exactly what the reference's band array returns for them.
// The last row of a strip is handed to the next strip's lane 0 through a row buffer in global scratch (M, D and their
// payloads per column, one region per block), 64 columns at a time in both directions: lane 63 collects its cells in a
// small LDS array that the whole wave flushes one column per lane; the next strip's lanes fetch one column each into
// registers a chunk ahead and publish it to LDS when its turn comes, so the per-step accesses are LDS only, global
// latency is paid once per 64 steps, and the kernel needs 10 KB of LDS (it shares the CUs with kp_protein_kernel).
constexpr int RB_FIELDS = 6;  // M, D, payload of M (2), payload of D (2)
static_assert(RB_FIELDS <= KP_PROT_ROWBUF_FIELDS, "the callers size the scratch with KP_PROT_ROWBUF_FIELDS ints per column");
constexpr int RB_CHUNK = 64;   // columns published per refill (one per lane)
constexpr int S2_CAP = 2048;   // residues of the second sequence staged per strip window

struct RowBuf {
    int *base;   // global scratch of this block
    int stride;  // ints between fields (len2 + 1)
    __device__ __forceinline__ int &at(int field, int j) const { return base[field * stride + j]; }
};

// column j of the previous strip's last row as lane 0 must see it: stored values inside [pj_lo, pj_hi], boundary outside
__device__ __forceinline__ void row_buf_column(const RowBuf &rb, int j, int pj_lo, int pj_hi, int (&v)[RB_FIELDS]) {
    v[0] = 0; v[1] = NEGP;
#pragma unroll
    for (int f = 2; f < RB_FIELDS; ++f) v[f] = 0;
    if (j >= pj_lo && j <= pj_hi) {
#pragma unroll
        for (int f = 0; f < RB_FIELDS; ++f) v[f] = rb.at(f, j);
    }
}

__device__ __forceinline__ Result protein_pair_strips(RowBuf rb, int (*s_chunk)[RB_CHUNK], int (*s_out)[RB_CHUNK],
                                                      uint16_t *s_seq2,
                                                      const uint8_t *s_idx, const int8_t *s_mat,
                                                      const uint8_t *__restrict__ s1, const uint8_t *__restrict__ s2,
                                                      int len1, int len2, int k, int shift, int lane) {
    Result r{0, 0, 0, Pay{0, 0}};
    int pj_lo = 1, pj_hi = 0;  // columns the previous strip left in the row buffer (none yet)
    for (int i0 = 1; i0 <= len1; i0 += 64) {
        const int i = i0 + lane;
        const int j_lo = max(1, i0 - shift - k), j_hi = min(len2, i0 + 63 - shift + k);
        const int width = j_hi - j_lo + 1;
        const bool last_strip = i0 + 64 > len1;
        __threadfence_block();  // the previous strip's row-buffer stores (lane 63) are visible to every lane's loads (one block: a
                                // device-scope fence here is an L2 write-back and invalidate per strip, kp_join.hip)
        __syncthreads();
        for (int x = lane; x < min(width, S2_CAP); x += 64) {
            const uint8_t c = s2[j_lo - 1 + x];
            s_seq2[x] = (uint16_t)(((unsigned)c << 8) | s_idx[c]);
        }
        __syncthreads();
        unsigned c1 = 0;
        if (i <= len1) { const uint8_t c = s1[i - 1]; c1 = ((unsigned)c << 8) | s_idx[c]; }
        int m = 0, dv = NEGP, iv = NEGP;  // this lane's previous cell (i, j-1); D of it is what the lane below reads
        Pay pm{0, 0}, pd{0, 0}, pi{0, 0};
        int dm = 0;  // M of (i-1, j-1) and its payload: the upper neighbour of the previous step
        Pay dpm{0, 0};
        if (lane == 0 && j_lo - 1 >= pj_lo && j_lo - 1 <= pj_hi) {  // only row i0 can have an in-band cell left of the window
            dm = rb.at(0, j_lo - 1);
            dpm = Pay{(unsigned)rb.at(2, j_lo - 1), (unsigned)rb.at(3, j_lo - 1)};
        }
        int ahead[RB_FIELDS];  // this lane's column of the chunk that is published next
        row_buf_column(rb, j_lo + lane, pj_lo, pj_hi, ahead);
        const int n_steps = width + 63;
        for (int t = 0; t < n_steps; ++t) {
            if ((t & (RB_CHUNK - 1)) == 0) {  // wave-uniform: publish columns j_lo + t .. + 63, start fetching the next 64
#pragma unroll
                for (int f = 0; f < RB_FIELDS; ++f) s_chunk[f][lane] = ahead[f];
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                row_buf_column(rb, j_lo + t + RB_CHUNK + lane, pj_lo, pj_hi, ahead);
            }
            const int x = t - lane, j = j_lo + x;
            // upper neighbour (i-1, j): lane l-1's cell of the previous step; row i0 reads the previous strip's last row
            int um = from_lower(m), ud = from_lower(dv);
            Pay upm{(unsigned)from_lower((int)pm.a), (unsigned)from_lower((int)pm.g)};
            Pay upd{(unsigned)from_lower((int)pd.a), (unsigned)from_lower((int)pd.g)};
            if (lane == 0) {
                const int c = t & (RB_CHUNK - 1);
                um = s_chunk[0][c]; ud = s_chunk[1][c];
                upm = Pay{(unsigned)s_chunk[2][c], (unsigned)s_chunk[3][c]};
                upd = Pay{(unsigned)s_chunk[4][c], (unsigned)s_chunk[5][c]};
            }
            const int raw_um = um;
            const Pay raw_upm = upm;
            int db = i - shift - j;
            if (db < 0) db = -db;
            const bool in = x >= 0 && x < width && i <= len1 && db <= k;
            unsigned c2 = 0;
            if (x >= 0 && x < width) {
                if (x < S2_CAP) c2 = s_seq2[x];
                else { const uint8_t c = s2[j - 1]; c2 = ((unsigned)c << 8) | s_idx[c]; }
            }
            PCell cell;  // comes in holding the diagonal neighbour (i-1, j-1), goes out holding (i, j)
            cell.m = dm; cell.pm = dpm;
            prot_cell(cell, r, in, i, j, c1, c2, s_mat, m, iv, pm, pi, um, ud, upm, upd);
            dm = raw_um; dpm = raw_upm;  // (i-1, j) is the diagonal neighbour of the next column
            m = cell.m; dv = cell.dv; iv = cell.iv; pm = cell.pm; pd = cell.pd; pi = cell.pi;
            // hand the strip's last row to the next strip: lane 63 collects its cells in LDS, every 64 columns (and at the
            // end of the row) the whole wave writes them to the row buffer, one column per lane
            const int xo = t - 63;  // lane 63's column index: wave-uniform
            if (!last_strip && xo >= 0 && xo < width) {
                const int slot = xo & (RB_CHUNK - 1);
                if (lane == 63) {
                    s_out[0][slot] = m; s_out[1][slot] = dv;
                    s_out[2][slot] = (int)pm.a; s_out[3][slot] = (int)pm.g;
                    s_out[4][slot] = (int)pd.a; s_out[5][slot] = (int)pd.g;
                }
                if (slot == RB_CHUNK - 1 || xo == width - 1) {
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                    if (lane <= slot) {
#pragma unroll
                        for (int f = 0; f < RB_FIELDS; ++f) rb.at(f, j_lo + xo - slot + lane) = s_out[f][lane];
                    }
                }
            }
        }
        pj_lo = j_lo; pj_hi = j_hi;
    }
    return r;
}

__device__ __forceinline__ bool fits_registers(int len1, int len2, int nb) {
    return nb <= 64 && len1 <= REG_MAX_LEN && len2 <= REG_MAX_LEN;
}

// reduction over the `width` lanes that hold one pair (64 = the whole wave): max score, then smallest i, then smallest j
__device__ __forceinline__ void store_result(Result r, int lane_in_group, int32_t *__restrict__ o, int width = 64,
                                             bool active = true) {
    int best = r.best, bi = r.bi, bj = r.bj;
    Pay bp = r.bp;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        if (off >= width) break;
        const int b2 = __shfl_xor(best, off), i2 = __shfl_xor(bi, off), j2 = __shfl_xor(bj, off);
        const unsigned a2 = __shfl_xor(bp.a, off), g2 = __shfl_xor(bp.g, off);
        // lanes that never saw a positive cell carry best = 0 and must lose against any positive score
        const bool take = b2 > best || (b2 == best && b2 > 0 && (i2 < bi || (i2 == bi && j2 < bj)));
        if (take) { best = b2; bi = i2; bj = j2; bp = Pay{a2, g2}; }
    }
    if (lane_in_group == 0 && active) {
        if (best > 0) {
            const int m = (int)(bp.a >> 16), mm = (int)(bp.a & 0xFFFFu), g_row = (int)(bp.g >> 16), g_col = (int)(bp.g & 0xFFFFu);
            o[0] = best; o[1] = m; o[2] = mm; o[3] = g_row + g_col;
            o[4] = bi - (m + mm + g_row); o[5] = bi; o[6] = bj - (m + mm + g_col); o[7] = bj;
        } else {
            for (int x = 0; x < 8; ++x) o[x] = 0;
        }
    }
}

// compact substitution table in LDS: s_idx = index of each byte in ARNDCQEGHILKMFPSTWYVBJZX* (31 for everything else),
// s_mat = the 32 x 32 corner of the reference's 256 x 256 lookup those indices address
__device__ __forceinline__ void stage_blosum(const int8_t *__restrict__ blosum, int8_t *s_mat, uint8_t *s_idx, int lane) {
    for (int c = lane; c < 256; c += 64) s_idx[c] = 31;
    __syncthreads();
    if (lane < 25) s_idx[(uint8_t)"ARNDCQEGHILKMFPSTWYVBJZX*"[lane]] = (uint8_t)lane;
    __syncthreads();
    for (int x = lane; x < 32 * 32; x += 64) {
        const int a = x >> 5, c = x & 31;
        s_mat[x] = (a < 25 && c < 25) ? blosum[(int)(uint8_t)"ARNDCQEGHILKMFPSTWYVBJZX*"[a] * 256 +
                                               (int)(uint8_t)"ARNDCQEGHILKMFPSTWYVBJZX*"[c]]
                                      : (int8_t)KP_PROT_FILL;
    }
    __syncthreads();
}

// ---- exact prefix: no DP needed ------------------------------------------------------------------------------------------------
// When the query (the protein translated from the assembly, stop excluded) consists of the 20 standard residues only and
// equals the first len1 residues of the target (the database protein, stop included), the reference's result is known
// without filling a matrix: score = sum of the residues' self-scores, matches = len1, no mismatch, no gap, both spans
// [0, len1).  Why: every aligned pair scores at most the query residue's self-score, which only the identical residue
// reaches (true of BLOSUM62 for the 20 standard residues; not for B / Z / X), and every gap costs; so an alignment ending
// at (i, i) other than the diagonal from (0, 0) scores strictly less than that diagonal, which also never drops to <= 0
// (self-scores are >= 4): the diagonal wins every cell of the traceback, the path starts at (0, 0), and the global
// maximum sum(self) is reached at (len1, len1) -- another placement of the whole query further right would end in the
// same row at a larger column, i.e. later in the reference's row-major scan.  The band always holds the main diagonal
// (k >= 20).  Most genes of a typed isolate in real data are identical to their reference; in the synthetic benchmark
// (0-3 % substitutions per locus) practically none is, there the check costs a pass over the residues.
// `idx` = index in ARNDCQEGHILKMFPSTWYVBJZX*: the standard residues are 0..19.
__device__ __forceinline__ bool standard_residue(unsigned idx) { return idx < 20u; }

// pairs whose band fits 64 diagonals (and the empty ones): four pairs per wave
__global__ __launch_bounds__(64) void kp_protein_kernel(const uint8_t *__restrict__ q, const int32_t *__restrict__ q_off,
                                                        const int32_t *__restrict__ q_len,
                                                        const uint8_t *__restrict__ t, const int32_t *__restrict__ t_off,
                                                        const int32_t *__restrict__ t_len, int32_t n_host,
                                                        const int32_t *__restrict__ n_dev,
                                                        const int8_t *__restrict__ blosum, int32_t *__restrict__ out8,
                                                        const int32_t *__restrict__ seed_off, int seed_k) {
    __shared__ uint16_t s_seq1[4][REG_MAX_LEN], s_seq2[4][REG_MAX_LEN];
    __shared__ int8_t s_mat[32 * 32];
    __shared__ uint8_t s_idx[256];
    const int lane = threadIdx.x, g = lane / QP, l = lane % QP;
    stage_blosum(blosum, s_mat, s_idx, lane);
    const int n = n_dev ? *n_dev : n_host;
    for (int p0 = 4 * blockIdx.x; p0 < n; p0 += 4 * gridDim.x) {
        const int p = p0 + g;
        int len1 = 0, len2 = 0;
        if (p < n) { len1 = q_len[p]; len2 = t_len[p]; }
        const bool empty = p < n && (len1 == 0 || len2 == 0);  // nothing to align: all zeros
        int d = len1 - len2;
        if (d < 0) d = -d;
        const int k = seed_off ? seed_k : max(KP_PROT_K, d + 1);
        const int shift = (seed_off && p < n) ? seed_off[p] : 0;
        const bool mine = p < n && !empty && fits_registers(len1, len2, 2 * k + 1);  // else kp_protein_wide_kernel's
        __syncthreads();
        if (mine) {
            const uint8_t *s1 = q + q_off[p], *s2 = t + t_off[p];
            for (int x = l; x < len1; x += QP) s_seq1[g][x] = (uint16_t)(((unsigned)s1[x] << 8) | s_idx[s1[x]]);
            for (int x = l; x < len2; x += QP) s_seq2[g][x] = (uint16_t)(((unsigned)s2[x] << 8) | s_idx[s2[x]]);
        }
        __syncthreads();
        bool exact = false;
        if (!seed_off) {  // (unseeded mode only: a seeded band need not hold the main diagonal)
            bool same = mine && len1 <= len2;
            int self = 0;
            if (same)
                for (int x = l; x < len1; x += QP) {
                    const unsigned c1 = s_seq1[g][x], i1 = c1 & 255u;
                    same = same && standard_residue(i1) && (c1 >> 8) == ((unsigned)s_seq2[g][x] >> 8);
                    self += (int)s_mat[(i1 * 33u) & 1023u];
                }
#pragma unroll
            for (int o = 1; o < QP; o <<= 1) {
                same = (__shfl_xor((int)same, o) != 0) && same;
                self += __shfl_xor(self, o);
            }
            exact = same;
            if (exact && l == 0) {
                int32_t *o = out8 + 8 * (size_t)p;
                o[0] = self; o[1] = len1; o[2] = 0; o[3] = 0; o[4] = 0; o[5] = len1; o[6] = 0; o[7] = len1;
            }
        }
        const bool dp = mine && !exact;
        int steps = dp ? len1 + QP - 1 : 0;
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) steps = max(steps, __shfl_xor(steps, o));
        // (wave-uniform choice: every pair of the quad that needs the DP fits 48 diagonals)
        const bool narrow = __all(!dp || 2 * k + 1 <= 3 * QP);
        const Result r = narrow ? protein_quad_registers<3>(s_seq1[g], s_seq2[g], s_mat, dp ? len1 : 0, dp ? len2 : 0, k, shift, l, steps)
                                : protein_quad_registers<4>(s_seq1[g], s_seq2[g], s_mat, dp ? len1 : 0, dp ? len2 : 0, k, shift, l, steps);
        store_result(r, l, out8 + 8 * (size_t)(p < n ? p : 0), QP, dp || empty);
    }
}

// the rest: wide bands (truncated / partial genes) and very long proteins.  A pair here is one long dependent chain of
// steps for a single wave, so the launch lasts as long as its slowest block: the blocks take the pairs off a shared
// counter instead of striding over them, and skip what kp_protein_kernel handles.
constexpr int QUEUE_GRAB = 1;
constexpr int WAVE_NC_MAX = 8;  // diagonals per lane of the wave-register path: bands of up to 512 diagonals

__global__ __launch_bounds__(64) void kp_protein_wide_kernel(const uint8_t *__restrict__ q, const int32_t *__restrict__ q_off,
                                                             const int32_t *__restrict__ q_len,
                                                             const uint8_t *__restrict__ t, const int32_t *__restrict__ t_off,
                                                             const int32_t *__restrict__ t_len, int32_t n_host,
                                                             const int32_t *__restrict__ n_dev,
                                                             const int8_t *__restrict__ blosum, int32_t *__restrict__ out8,
                                                             int32_t *__restrict__ scratch, size_t scratch_ints_per_block,
                                                             int32_t *__restrict__ queue,
                                                             const int32_t *__restrict__ seed_off, int seed_k) {
    __shared__ int s_chunk[RB_FIELDS][RB_CHUNK], s_out[RB_FIELDS][RB_CHUNK];
    __shared__ uint16_t s_seq2[S2_CAP];
    __shared__ uint16_t s_seq1[REG_MAX_LEN];  // (wave-register path)
    __shared__ int8_t s_mat[32 * 32];
    __shared__ uint8_t s_idx[256];
    const int lane = threadIdx.x;
    const int n = n_dev ? *n_dev : n_host;
    bool staged = false;
    for (;;) {
        int p0 = 0;
        if (lane == 0) p0 = atomicAdd(queue, QUEUE_GRAB);
        p0 = __shfl(p0, 0);
        if (p0 >= n) break;
        for (int p = p0; p < min(n, p0 + QUEUE_GRAB); ++p) {
            const int len1 = q_len[p], len2 = t_len[p];
            if (len1 == 0 || len2 == 0) continue;
            int d = len1 - len2;
            if (d < 0) d = -d;
            const int k = seed_off ? seed_k : max(KP_PROT_K, d + 1);
            const int shift = seed_off ? seed_off[p] : 0;
            if (fits_registers(len1, len2, 2 * k + 1)) continue;
            if (!staged) { stage_blosum(blosum, s_mat, s_idx, lane); staged = true; }  // many blocks find nothing to do
            if (!seed_off && len1 <= len2) {  // exact prefix (a gene cut short by a stop or a contig end, otherwise unchanged)
                const uint8_t *s1 = q + q_off[p], *s2 = t + t_off[p];
                bool same = true;
                int self = 0;
                for (int x = lane; x < len1; x += 64) {
                    const unsigned i1 = s_idx[s1[x]];
                    same = same && standard_residue(i1) && s1[x] == s2[x];
                    self += (int)s_mat[(i1 * 33u) & 1023u];
                }
#pragma unroll
                for (int o = 1; o < 64; o <<= 1) {
                    same = (__shfl_xor((int)same, o) != 0) && same;
                    self += __shfl_xor(self, o);
                }
                if (same) {
                    if (lane == 0) {
                        int32_t *o = out8 + 8 * (size_t)p;
                        o[0] = self; o[1] = len1; o[2] = 0; o[3] = 0; o[4] = 0; o[5] = len1; o[6] = 0; o[7] = len1;
                    }
                    continue;
                }
            }
            if (2 * k + 1 <= 64 * WAVE_NC_MAX && len1 <= REG_MAX_LEN && len2 <= S2_CAP) {
                // the whole band in the registers of one wave: len1 + 63 steps instead of (len1 / 64) strips x (window + 63)
                const uint8_t *s1 = q + q_off[p], *s2 = t + t_off[p];
                __syncthreads();  // (a block is one wave: orders the LDS stores below after the previous pair's reads)
                for (int x = lane; x < len1; x += 64) s_seq1[x] = (uint16_t)(((unsigned)s1[x] << 8) | s_idx[s1[x]]);
                for (int x = lane; x < len2; x += 64) s_seq2[x] = (uint16_t)(((unsigned)s2[x] << 8) | s_idx[s2[x]]);
                __syncthreads();
                const int steps = len1 + 63;
                const Result r = 2 * k + 1 <= 64 * 4 ? protein_quad_registers<4, 64>(s_seq1, s_seq2, s_mat, len1, len2, k, shift, lane, steps)
                                                     : protein_quad_registers<WAVE_NC_MAX, 64>(s_seq1, s_seq2, s_mat, len1, len2, k, shift, lane, steps);
                store_result(r, lane, out8 + 8 * (size_t)p);
                continue;
            }
            if (len1 <= 64 * 6 && len2 <= S2_CAP) {  // a fragment against a long protein: the whole matrix, rows per lane
                const uint8_t *s1 = q + q_off[p], *s2 = t + t_off[p];
                __syncthreads();
                for (int x = lane; x < len1; x += 64) s_seq1[x] = (uint16_t)(((unsigned)s1[x] << 8) | s_idx[s1[x]]);
                for (int x = lane; x < len2; x += 64) s_seq2[x] = (uint16_t)(((unsigned)s2[x] << 8) | s_idx[s2[x]]);
                __syncthreads();
                const Result r = len1 <= 64 * 2 ? protein_wave_rows<2>(s_seq1, s_seq2, s_mat, len1, len2, k, shift, lane)
                                 : len1 <= 64 * 4 ? protein_wave_rows<4>(s_seq1, s_seq2, s_mat, len1, len2, k, shift, lane)
                                                  : protein_wave_rows<6>(s_seq1, s_seq2, s_mat, len1, len2, k, shift, lane);
                store_result(r, lane, out8 + 8 * (size_t)p);
                continue;
            }
            const RowBuf rb{scratch + (size_t)blockIdx.x * scratch_ints_per_block, len2 + 1};
            const Result r = protein_pair_strips(rb, s_chunk, s_out, s_seq2, s_idx, s_mat, q + q_off[p], t + t_off[p], len1, len2,
                                                 k, shift, lane);
            store_result(r, lane, out8 + 8 * (size_t)p);
        }
    }
}

}  // namespace

// n pairs; when n_dev is not null the pair count is read from device memory (n is then only an upper bound)
void kp_launch_protein(const uint8_t *q, const int32_t *q_off, const int32_t *q_len, const uint8_t *t,
                       const int32_t *t_off, const int32_t *t_len, int32_t n, const int32_t *n_dev, const int8_t *blosum,
                       int32_t *out8, int32_t *scratch, size_t scratch_ints_per_block, int n_blocks, hipStream_t stream,
                       hipStream_t aux, hipEvent_t fork, hipEvent_t join, const int32_t *seed_off, int seed_k) {
    if (n == 0) return;
    // The wide-band kernel is a handful of long-running waves (its duration is one pair's time-step chain), the
    // register kernel fills the chip: with a second stream they run side by side, joined before `stream` goes on.
    hipStream_t wide = stream;
    if (aux) {
        (void)hipEventRecord(fork, stream);
        (void)hipStreamWaitEvent(aux, fork, 0);
        wide = aux;
    }
    int32_t *queue = scratch + (size_t)n_blocks * scratch_ints_per_block;  // callers leave 64 ints behind the regions
    (void)hipMemsetAsync(queue, 0, sizeof(int32_t), wide);
    hipLaunchKernelGGL(kp_protein_wide_kernel, dim3(n_blocks), dim3(64), 0, wide, q, q_off, q_len, t, t_off, t_len, n, n_dev,
                       blosum, out8, scratch, scratch_ints_per_block, queue, seed_off, seed_k);
    hipLaunchKernelGGL(kp_protein_kernel, dim3(std::min(n_blocks, (n + 3) / 4)), dim3(64), 0, stream, q, q_off, q_len, t,
                       t_off, t_len, n, n_dev, blosum, out8, seed_off, seed_k);
    if (aux) {
        (void)hipEventRecord(join, aux);
        (void)hipStreamWaitEvent(stream, join, 0);
    }
}
