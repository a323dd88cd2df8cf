// kp_sw.hip -- banded local alignment (Smith-Waterman-Gotoh, affine gaps, int32) of every band task.
//
// Stands in for the extension half of rammappy's map_batch (reference call site src/kaptive/serotyping/core.py:154;
// fields consumed at src/kaptive/core/alignment.py:415-446).  Recurrence, tie rules and scores: include/kp_spec.h.
//
// Mapping (wavefront-parallel anti-diagonals, no MFMA -- this is dependent integer DP, not a contraction):
//   * a task's band has W = 4P diagonals; P lanes own it, lane l holds the four adjacent diagonals 4l .. 4l+3 (cells
//     A..D); a 64-lane wave therefore runs 64/P tasks side by side (P = 4/8/16/32 for W = 16/32/64/128).
//   * time is skewed by lane: at step m lane l works on query row r = m - l, cells A, B, C, D in that order.  With
//     that skew A's left neighbour is lane l-1's D of the previous step, D's upper neighbour is lane l+1's A of the
//     same step, and every other neighbour is one of the lane's own registers -- two one-lane shifts per step, both
//     done with DPP (v_mov_b32_dpp row_shr:1 / row_shl:1, wave_* for the 32-lane class), all state stays in registers.
//   * sequences are streamed systolically: the query enters at lane 0 as a per-row score profile (five 6-bit signed
//     fields indexed by the target code, so a substitution score is one v_bfe_i32) and moves up one lane per step, the
//     target code enters at lane P-1 and moves down; both are staged per chunk in LDS (N and outside-contig have their
//     own codes, so N runs and contig ends need no special path).
//   * the gap states are kept pre-charged (H - open - ext, E - ext, F - ext), so E and F of a neighbour are one max.
//   * start coordinates, matches and column counts ride along with the scores ("carry-forward" of the traceback): every
//     state value carries the payload of the predecessor it was derived from, chosen by exactly the tie rules the
//     oracle's stored traceback uses, so no DP matrix is ever written to memory.
//   * steps whose rows and target positions are all inside the task (the bulk) run a variant without boundary masks.
//   * the best cell of a task is found with a wave-level max reduction over (score, first row, first column).
#include <cstdlib>

#include "kp_internal.h"

namespace {

constexpr int CH = 128;  // steps staged per chunk
constexpr int NEG = KP_NEG_INF;
constexpr int OE = KP_GAP_OPEN + KP_GAP_EXT;
constexpr int EX = KP_GAP_EXT;
constexpr unsigned T_OUT = 30u;  // profile field offset standing for "outside the contig" (no such field)
// score profile of a query row: field t (6 bits, signed, at bit 6t) = score against target code t (0..3 ACGT, 4 = N)
constexpr unsigned PROF_N = 0x3FFFFFFFu;  // KP_SC_N in every field
constexpr unsigned PROF_MISMATCH = 0x3Cu | (0x3Cu << 6) | (0x3Cu << 12) | (0x3Cu << 18) | (0x3Fu << 24);
static_assert(KP_SC_MATCH == 2 && KP_SC_MISMATCH == -4 && KP_SC_N == -1, "profile constants encode these scores");

// One-lane shifts.  The lane without a source reads 0 (bound_ctrl); every group-edge lane overrides what it receives
// anyway.  Groups of up to 16 lanes never straddle a DPP row, so the row shifts do; the 32-lane class needs wave shifts.
template <bool ROW>
__device__ __forceinline__ int from_lower(int v) {  // lane i <- lane i-1
    return ROW ? __builtin_amdgcn_mov_dpp(v, 0x111 /*row_shr:1*/, 0xf, 0xf, true)
               : __builtin_amdgcn_mov_dpp(v, 0x138 /*wave_shr:1*/, 0xf, 0xf, true);
}
template <bool ROW>
__device__ __forceinline__ int from_upper(int v) {  // lane i <- lane i+1
    return ROW ? __builtin_amdgcn_mov_dpp(v, 0x101 /*row_shl:1*/, 0xf, 0xf, true)
               : __builtin_amdgcn_mov_dpp(v, 0x130 /*wave_shl:1*/, 0xf, 0xf, true);
}

__device__ __forceinline__ unsigned nibble(unsigned word, int i) { return (word >> (4 * i)) & 15u; }
__device__ __forceinline__ unsigned row_profile(unsigned qcode) {
    return qcode < 4u ? (PROF_MISMATCH ^ (0x3Eu << (6u * qcode))) : PROF_N;  // -4 ^ 0x3E = +2 in the matching field
}

// payload words: lo = matches << 16 | columns, hi = start_row << 8 | start_band_index
struct Cell {
    int h, hmoe, emex, fmex;  // H, H - (open + ext), E - ext, F - ext
    unsigned hlo, hhi, elo, ehi, flo, fhi;
};
struct Gap {  // what a neighbour offers: its pre-charged H and gap state with their payloads
    int hmoe, gmex;
    unsigned hlo, hhi, glo, ghi;
};
struct Best {
    int score, r;
    unsigned lo, hi;
};

template <bool MASKED>
__device__ __forceinline__ void dp_cell(Cell &c, Best &best, int r, unsigned start_key, unsigned prof, unsigned tsh,
                                        bool row_ok, const Gap &left, const Gap &up) {
    // E: gap in the query, arrives from the left; F: gap in the target, arrives from above (open wins ties)
    const bool eo = left.hmoe >= left.gmex;
    const int e = max(left.hmoe, left.gmex);
    const unsigned e_lo = (eo ? left.hlo : left.glo) + 1u, e_hi = eo ? left.hhi : left.ghi;
    const bool fo = up.hmoe >= up.gmex;
    const int f = max(up.hmoe, up.gmex);
    const unsigned f_lo = (fo ? up.hlo : up.glo) + 1u, f_hi = fo ? up.hhi : up.ghi;
    // diagonal: the lane's own previous row
    const int s = __builtin_amdgcn_sbfe((int)prof, tsh, 6u);
    const bool fresh = c.h == 0;
    unsigned p_lo = (fresh ? 0u : c.hlo) + ((unsigned)max(s, 0) * 0x8000u + 1u);  // +1 column, +1 match when s == 2
    unsigned p_hi = fresh ? start_key : c.hhi;
    int bestv = c.h + s;
    if (e > bestv) { bestv = e; p_lo = e_lo; p_hi = e_hi; }
    if (f > bestv) { bestv = f; p_lo = f_lo; p_hi = f_hi; }
    int h = max(bestv, 0), emex = e - EX, fmex = f - EX;
    if (MASKED) {  // outside the task (row out of range, target outside the contig): reads as boundary for its neighbours
        const bool live = row_ok && tsh != T_OUT;
        h = live ? h : 0; emex = live ? emex : NEG; fmex = live ? fmex : NEG;
    }
    c.h = h; c.hmoe = h - OE; c.emex = emex; c.fmex = fmex;
    c.hlo = p_lo; c.hhi = p_hi; c.elo = e_lo; c.ehi = e_hi; c.flo = f_lo; c.fhi = f_hi;
    if (h > best.score) { best.score = h; best.r = r; best.lo = p_lo; best.hi = p_hi; }
}

__device__ __forceinline__ Gap as_left(const Cell &c) { return Gap{c.hmoe, c.emex, c.hlo, c.hhi, c.elo, c.ehi}; }
__device__ __forceinline__ Gap as_up(const Cell &c) { return Gap{c.hmoe, c.fmex, c.hlo, c.hhi, c.flo, c.fhi}; }

struct State {
    Cell A, B, C, D;
    Best bA, bB, bC, bD;
    unsigned qb, t0, t1, t2, t3;
};

template <int P, bool MASKED>
__device__ __forceinline__ void dp_step(State &s, int m, int l, int qlen, unsigned prof_in, unsigned t_in) {
    constexpr bool ROW = P <= 16;
    const unsigned q_shift = (unsigned)from_lower<ROW>((int)s.qb);
    s.qb = (l == 0) ? prof_in : q_shift;  // profile of row m enters at lane 0
    const int r = m - l;
    const bool row_ok = MASKED ? ((unsigned)r < (unsigned)qlen) : true;
    const unsigned key = ((unsigned)r << 8) | (unsigned)(4 * l);

    Gap left;  // lane l-1's D of the previous step
    left.hmoe = from_lower<ROW>(s.D.hmoe); left.gmex = from_lower<ROW>(s.D.emex);
    left.hlo = (unsigned)from_lower<ROW>((int)s.D.hlo); left.hhi = (unsigned)from_lower<ROW>((int)s.D.hhi);
    left.glo = (unsigned)from_lower<ROW>((int)s.D.elo); left.ghi = (unsigned)from_lower<ROW>((int)s.D.ehi);
    if (l == 0) { left.hmoe = -OE; left.gmex = NEG; }
    dp_cell<MASKED>(s.A, s.bA, r, key, s.qb, s.t0, row_ok, left, as_up(s.B));
    dp_cell<MASKED>(s.B, s.bB, r, key + 1u, s.qb, s.t1, row_ok, as_left(s.A), as_up(s.C));
    dp_cell<MASKED>(s.C, s.bC, r, key + 2u, s.qb, s.t2, row_ok, as_left(s.B), as_up(s.D));
    Gap up;  // lane l+1's A of this step
    up.hmoe = from_upper<ROW>(s.A.hmoe); up.gmex = from_upper<ROW>(s.A.fmex);
    up.hlo = (unsigned)from_upper<ROW>((int)s.A.hlo); up.hhi = (unsigned)from_upper<ROW>((int)s.A.hhi);
    up.glo = (unsigned)from_upper<ROW>((int)s.A.flo); up.ghi = (unsigned)from_upper<ROW>((int)s.A.fhi);
    if (l == P - 1) { up.hmoe = -OE; up.gmex = NEG; }
    dp_cell<MASKED>(s.D, s.bD, r, key + 3u, s.qb, s.t3, row_ok, as_left(s.C), up);

    const unsigned t_shift = (unsigned)from_upper<ROW>((int)s.t1);  // lane l+1's x = m + 3l + 4 = this lane's next t3
    s.t0 = s.t1; s.t1 = s.t2; s.t2 = s.t3;
    s.t3 = (l == P - 1) ? t_in : t_shift;  // target code x = m + 1 + 3P enters at lane P-1
}

// target code at window position t (assembly coordinates) for one task
__device__ __forceinline__ unsigned target_code(const uint32_t *__restrict__ asm_words, int32_t t, int32_t cstart,
                                                int32_t cend, const int32_t *__restrict__ runs, int n_runs) {
    if (t < cstart || t >= cend) return 5u;
    if (n_runs > 0) {
        int lo = 0, hi = n_runs;
        while (lo < hi) {
            int mid = (lo + hi) >> 1;
            if (runs[2 * mid + 1] <= t) lo = mid + 1; else hi = mid;
        }
        if (lo < n_runs && runs[2 * lo] <= t) return 4u;
    }
    return (asm_words[t >> 4] >> (2 * (t & 15))) & 3u;
}

constexpr int PROF_WORDS = 16 * CH;                  // LDS of one block: profiles of 64/P groups x CH rows (P >= 4)
constexpr int TCODE_BYTES = 16 * (CH + 3 * 4 + 4);   // ... and their target codes (largest for P = 4)

// all tasks of one band class, seen from block `block` of `n_blocks` that work on the class
template <int P>
__device__ __forceinline__ void sw_class(const KpBatchView &b, const KpGenes &genes, const KpTask *__restrict__ tasks,
                                         uint32_t n_tasks, const uint32_t *__restrict__ order,
                                         KpSwResult *__restrict__ results, uint32_t block, uint32_t n_blocks,
                                         uint32_t *s_prof_raw, uint8_t *s_t_raw) {
    constexpr int G = 64 / P;
    constexpr int W = 4 * P;
    constexpr int TW = CH + 3 * P + 4;  // staged target codes per chunk: window x in [m0, m0 + CH + 3P]
    static_assert(G * CH <= PROF_WORDS && G * TW <= TCODE_BYTES, "LDS carve-up");
    uint32_t(*s_prof)[CH] = reinterpret_cast<uint32_t(*)[CH]>(s_prof_raw);
    uint8_t(*s_t)[TW] = reinterpret_cast<uint8_t(*)[TW]>(s_t_raw);

    const int lane = threadIdx.x;
    const int g = lane / P, l = lane % P;

    for (uint32_t quad = block; (uint64_t)quad * G < n_tasks; quad += n_blocks) {
        const uint32_t slot = quad * G + g;
        const bool have = slot < n_tasks;
        const uint32_t ti = have ? order[slot] : 0u;  // tasks of similar length share a wave (kp_chain.hip)
        KpTask tk;
        tk.asm_id = 0; tk.gs = 0; tk.contig = 0; tk.lo = 0;
        if (have) tk = tasks[ti];
        const int gene = tk.gs >> 1;
        const int qlen = have ? genes.len[gene] : 0;
        const uint32_t *qnib = genes.nib + genes.word_off[(tk.gs & 1) ? genes.n_genes + gene : gene];
        const uint32_t *asm_words = b.words + b.asm_word_off[tk.asm_id];
        const int c_abs = b.asm_first_ctg[tk.asm_id] + tk.contig;
        const int32_t cstart = b.ctg_start[c_abs], cend = cstart + b.ctg_len[c_abs];
        const int r0 = b.asm_first_nrun[tk.asm_id];
        const int n_runs = b.asm_first_nrun[tk.asm_id + 1] - r0;
        const int32_t *runs = b.n_runs + 2 * (size_t)r0;
        const int lo = tk.lo;

        const int steps = have ? qlen + P - 1 : 0;  // steps this group needs
        int max_steps = steps, min_q = qlen;
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) {
            max_steps = max(max_steps, __shfl_xor(max_steps, o));
            min_q = min(min_q, __shfl_xor(min_q, o));
        }
        // every cell of every task of the wave inside its contig: steps P-1 .. min_q-1 need no boundary masks
        const bool all_inside = __all(have && lo >= cstart && lo + qlen + W <= cend);

        State st;
        st.A.h = 0; st.A.hmoe = -OE; st.A.emex = NEG; st.A.fmex = NEG;
        st.A.hlo = st.A.hhi = st.A.elo = st.A.ehi = st.A.flo = st.A.fhi = 0;
        st.B = st.A; st.C = st.A; st.D = st.A;
        st.bA = Best{0, 0, 0, 0}; st.bB = st.bA; st.bC = st.bA; st.bD = st.bA;
        st.qb = PROF_N; st.t0 = st.t1 = st.t2 = st.t3 = T_OUT;

        for (int m0 = 0; m0 < max_steps; m0 += CH) {
            // ---- stage this chunk: profiles of query rows [m0, m0+CH), target codes of window x in [m0, m0+CH+3P] ----
            __syncthreads();
            for (int w = l; w < CH / 8; w += P) {
                const int r = m0 + 8 * w;
                const uint32_t word = (have && r < qlen) ? qnib[r >> 3] : 0x44444444u;
#pragma unroll
                for (int i = 0; i < 8; ++i) s_prof[g][8 * w + i] = row_profile(nibble(word, i));
            }
            for (int x = l; x < TW; x += P) {
                const unsigned code = have ? target_code(asm_words, lo + m0 + x, cstart, cend, runs, n_runs) : 5u;
                s_t[g][x] = (uint8_t)(code < 5u ? 6u * code : T_OUT);
            }
            __syncthreads();
            if (m0 == 0) {  // initial window: cell k of lane l sits on x = 3l + k
                st.t0 = s_t[g][3 * l]; st.t1 = s_t[g][3 * l + 1]; st.t2 = s_t[g][3 * l + 2]; st.t3 = s_t[g][3 * l + 3];
            }
            const int m_end = min(m0 + CH, max_steps);
            int m = m0;
            while (m < m_end) {
                if (all_inside && m >= P - 1 && m + 4 <= min_q && m + 4 <= m_end) {
#pragma unroll
                    for (int u = 0; u < 4; ++u)
                        dp_step<P, false>(st, m + u, l, qlen, s_prof[g][m - m0 + u], s_t[g][m - m0 + u + 3 * P + 1]);
                    m += 4;
                } else {
                    dp_step<P, true>(st, m, l, qlen, s_prof[g][m - m0], s_t[g][m - m0 + 3 * P + 1]);
                    m += 1;
                }
            }
        }

        // ---- best cell of the task: max score, then first row, then first column ---------------------------------
        // (a lane visits its cells in column order and only replaces on a strictly higher score)
        int sc = st.bA.score, er = st.bA.r, eb = 4 * l;
        unsigned plo = st.bA.lo, phi = st.bA.hi;
        if (st.bB.score > sc || (st.bB.score == sc && st.bB.r < er)) { sc = st.bB.score; er = st.bB.r; eb = 4 * l + 1; plo = st.bB.lo; phi = st.bB.hi; }
        if (st.bC.score > sc || (st.bC.score == sc && st.bC.r < er)) { sc = st.bC.score; er = st.bC.r; eb = 4 * l + 2; plo = st.bC.lo; phi = st.bC.hi; }
        if (st.bD.score > sc || (st.bD.score == sc && st.bD.r < er)) { sc = st.bD.score; er = st.bD.r; eb = 4 * l + 3; plo = st.bD.lo; phi = st.bD.hi; }
#pragma unroll
        for (int o = 1; o < P; o <<= 1) {
            const int sc2 = __shfl_xor(sc, o), er2 = __shfl_xor(er, o), eb2 = __shfl_xor(eb, o);
            const unsigned plo2 = __shfl_xor(plo, o), phi2 = __shfl_xor(phi, o);
            const bool take = sc2 > sc || (sc2 == sc && (er2 < er || (er2 == er && eb2 < eb)));
            if (take) { sc = sc2; er = er2; eb = eb2; plo = plo2; phi = phi2; }
        }
        if (have && l == 0) {
            KpSwResult out;
            if (sc > 0) {
                const int sr = (int)(phi >> 8), sb = (int)(phi & 255u);
                out.score = sc; out.q_start = sr; out.q_end = er + 1;
                out.t_start = sr + lo + sb; out.t_end = er + lo + eb + 1;
                out.matches = (int)(plo >> 16); out.block_len = (int)(plo & 0xFFFFu);
            } else {
                out.score = out.q_start = out.q_end = out.t_start = out.t_end = out.matches = out.block_len = 0;
            }
            results[ti] = out;
        }
    }
}

// One launch for all four band classes.  The wide classes hold few tasks but each of their waves runs a long step chain
// (a launch of its own costs ~1 ms of latency at the end of the pass), so they get the first blocks of the grid and
// run underneath the 16-diagonal class that fills the chip.
constexpr uint32_t WIDE_BLOCKS = 512;  // blocks per wide class (they stride over their quads)

// (64, 4): four waves per SIMD (128 VGPRs; what spills sits outside the step loop) -- the cells of a step depend on each
// other, so a wave alone cannot keep the SIMD busy: 23.5 ms against 24.0 ms with three waves
__global__ __launch_bounds__(64, 4) void kp_sw_kernel(KpBatchView b, KpGenes genes, const KpTask *__restrict__ tasks,
                                                   const uint32_t *__restrict__ task_count, uint32_t task_cap,
                                                   const uint32_t *__restrict__ order,
                                                   KpSwResult *__restrict__ results) {
    __shared__ uint32_t s_prof[PROF_WORDS];
    __shared__ uint8_t s_t[TCODE_BYTES];
    // class c (0..3 = 16/32/64/128 diagonals): tasks, order and results at c * task_cap, count at task_count[c]
    const uint32_t blk = blockIdx.x;
    const int c = blk < 3 * WIDE_BLOCKS ? 3 - (int)(blk / WIDE_BLOCKS) : 0;
    const size_t off = (size_t)c * task_cap;
    uint32_t n = task_count[c];
    if (n > task_cap) n = task_cap;
    if (c == 3) sw_class<32>(b, genes, tasks + off, n, order + off, results + off, blk, WIDE_BLOCKS, s_prof, s_t);
    else if (c == 2) sw_class<16>(b, genes, tasks + off, n, order + off, results + off, blk - WIDE_BLOCKS, WIDE_BLOCKS, s_prof, s_t);
    else if (c == 1) sw_class<8>(b, genes, tasks + off, n, order + off, results + off, blk - 2 * WIDE_BLOCKS, WIDE_BLOCKS, s_prof, s_t);
    else sw_class<4>(b, genes, tasks + off, n, order + off, results + off, blk - 3 * WIDE_BLOCKS, gridDim.x - 3 * WIDE_BLOCKS, s_prof, s_t);
}

}  // namespace

void kp_launch_sw(const KpBatchView &b, const KpGenes &genes, const KpTask *tasks, const uint32_t *task_count,
                  uint32_t task_cap, const uint32_t *order, KpSwResult *results, int blocks_per_cu, hipStream_t stream) {
    // many short-lived single-wave blocks (each strides over a quad or two): CU slots turn over every few hundred
    // microseconds, so the tail is even and the high-priority streams of other batches' reductions get their turn
    // (measured: 24.0 ms with 256 blocks per CU against 28-32 ms with 16 persistent ones, K pass of the bench)
    const dim3 grid(3 * WIDE_BLOCKS + 256 * (unsigned)(blocks_per_cu > 0 ? blocks_per_cu : 256)), block(64);
    hipLaunchKernelGGL(kp_sw_kernel, grid, block, 0, stream, b, genes, tasks, task_count, task_cap, order, results);
}
