// kp_sw.hip -- banded local alignment (Smith-Waterman-Gotoh, affine gaps, int32) of every band task.
//
// Stands in for the extension half of rammappy's map_batch (reference call site src/kaptive/serotyping/core.py:154;
// fields consumed at src/kaptive/core/alignment.py:415-446).  Recurrence, tie rules and scores: include/kp_spec.h.
//
// Mapping (wavefront-parallel anti-diagonals, no MFMA -- this is dependent integer DP, not a contraction):
//   * a task's band has W = 2P diagonals; P lanes own it, lane l holds diagonals 2l ("A") and 2l+1 ("B");
//     a 64-lane wave therefore runs 64/P tasks side by side (P = 8/16/32/64 for W = 16/32/64/128).
//   * time is skewed by lane: at macro step m lane l works on query row r = m - l, first cell A then cell B.  With that
//     skew A's left neighbour is lane l-1's B of the previous step, B's upper neighbour is lane l+1's A of the
//     same step, and everything else is the lane's own previous row -- two one-lane wave shifts per macro step, both
//     done with DPP (v_mov_b32_dpp wave_shr:1 / wave_shl:1), all state stays in registers.
//   * sequences are streamed systolically: the query code enters at lane 0 and moves up one lane per step, the
//     target code enters at lane P-1 and moves down; each group stages its query chunk and target window as 4-bit
//     codes in LDS (N = 4, outside-contig = 5), so N runs and contig ends need no special path.
//   * start coordinates, matches and column counts ride along with the scores ("carry-forward" of the traceback): every
//     state value carries the payload of the predecessor it was derived from, chosen by exactly the tie rules the
//     oracle's stored traceback uses, so no DP matrix is ever written to memory.
//   * the best cell of a task is found with a wave-level max reduction over (score, first row, first column).
#include "kp_internal.h"

namespace {

constexpr int CH = 1024;  // macro steps staged per chunk
constexpr int NEG = KP_NEG_INF;
constexpr int OE = KP_GAP_OPEN + KP_GAP_EXT;
constexpr int EX = KP_GAP_EXT;

// One-lane wave shifts.  The lane without a source (0 resp. 63) reads 0 (bound_ctrl); every group-edge lane overrides
// what it receives anyway, so no "old" operand (and no extra v_mov) is needed.
__device__ __forceinline__ int dpp_from_lower(int v) {  // lane i <- lane i-1
    return __builtin_amdgcn_mov_dpp(v, 0x138 /*wave_shr:1*/, 0xf, 0xf, true);
}
__device__ __forceinline__ int dpp_from_upper(int v) {  // lane i <- lane i+1
    return __builtin_amdgcn_mov_dpp(v, 0x130 /*wave_shl:1*/, 0xf, 0xf, true);
}

struct Cell {
    int h, e, f;
    unsigned hp_lo, hp_hi;  // payload of h: lo = matches << 16 | columns, hi = start_row << 8 | start_band_index
    unsigned ep_lo, ep_hi;
    unsigned fp_lo, fp_hi;
};

struct Best {
    int score, end_r;
    unsigned p_lo, p_hi;
};

__device__ __forceinline__ unsigned nibble(unsigned word, int i) { return (word >> (4 * i)) & 15u; }

// one DP cell; `left`/`up` give (h, gap score, payloads) of the neighbours, `diag_h`/`diag_p` the diagonal one
__device__ __forceinline__ void dp_cell(Cell &c, bool live, int r, int band_idx, unsigned qb, unsigned tb, int diag_h,
                                        unsigned diag_lo, unsigned diag_hi, int left_h, int left_e, unsigned left_hlo,
                                        unsigned left_hhi, unsigned left_elo, unsigned left_ehi, int up_h, int up_f,
                                        unsigned up_hlo, unsigned up_hhi, unsigned up_flo, unsigned up_fhi, Best &best) {
    if (!live) {  // outside the task (row out of range, target outside the contig): reads as boundary for its neighbours
        c.h = 0; c.e = NEG; c.f = NEG;
        return;
    }
    // E: gap in the query, arrives from the left
    const int e_open = left_h - OE, e_ext = left_e - EX;
    const bool eo = e_open >= e_ext;
    const int e = eo ? e_open : e_ext;
    const unsigned e_lo = (eo ? left_hlo : left_elo) + 1u, e_hi = eo ? left_hhi : left_ehi;
    // F: gap in the target, arrives from above
    const int f_open = up_h - OE, f_ext = up_f - EX;
    const bool fo = f_open >= f_ext;
    const int f = fo ? f_open : f_ext;
    const unsigned f_lo = (fo ? up_hlo : up_flo) + 1u, f_hi = fo ? up_hhi : up_fhi;
    // diagonal
    const bool known = (qb | tb) < 4u;
    const bool eq = known && (qb == tb);
    const int s = known ? (eq ? KP_SC_MATCH : KP_SC_MISMATCH) : KP_SC_N;
    const bool fresh = diag_h == 0;
    unsigned p_lo = (fresh ? 0u : diag_lo) + (eq ? 0x10001u : 1u);
    unsigned p_hi = fresh ? (((unsigned)r << 8) | (unsigned)band_idx) : diag_hi;
    int bestv = diag_h + s;
    if (e > bestv) { bestv = e; p_lo = e_lo; p_hi = e_hi; }
    if (f > bestv) { bestv = f; p_lo = f_lo; p_hi = f_hi; }
    const int h = bestv > 0 ? bestv : 0;
    c.h = h; c.e = e; c.f = f;
    c.hp_lo = p_lo; c.hp_hi = p_hi;
    c.ep_lo = e_lo; c.ep_hi = e_hi;
    c.fp_lo = f_lo; c.fp_hi = f_hi;
    if (h > best.score) { best.score = h; best.end_r = r; best.p_lo = p_lo; best.p_hi = p_hi; }
}

// target code at window position x (relative to the band's lowest diagonal) for one task
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

template <int P>
__global__ __launch_bounds__(64) void kp_sw_kernel(KpBatchView b, KpGenes genes, const KpTask *__restrict__ tasks,
                                                   const uint32_t *__restrict__ task_count, uint32_t task_cap,
                                                   const uint32_t *__restrict__ order,
                                                   KpSwResult *__restrict__ results) {
    constexpr int G = 64 / P;
    constexpr int TW = (CH + P + 1 + 7) / 8 + 1;  // words of staged target codes per group
    __shared__ uint32_t s_q[G][CH / 8];
    __shared__ uint32_t s_t[G][TW];

    const int lane = threadIdx.x;
    const int g = lane / P, l = lane % P;
    uint32_t n_tasks = *task_count;
    if (n_tasks > task_cap) n_tasks = task_cap;

    for (uint32_t quad = blockIdx.x; (uint64_t)quad * G < n_tasks; quad += gridDim.x) {
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

        int steps = have ? qlen + P - 1 : 0;  // macro steps this group needs
        int max_steps = steps;
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) max_steps = max(max_steps, __shfl_xor(max_steps, o));

        Cell A, B;
        A.h = B.h = 0; A.e = B.e = A.f = B.f = NEG;
        A.hp_lo = A.hp_hi = A.ep_lo = A.ep_hi = A.fp_lo = A.fp_hi = 0;
        B = A;
        Best bestA{0, 0, 0, 0}, bestB{0, 0, 0, 0};
        unsigned qb = 4, t0 = 5, t1 = 5, qword = 0, tword = 0;

        for (int m0 = 0; m0 < max_steps; m0 += CH) {
            // ---- stage this chunk: query rows [m0, m0+CH) and target window x in [m0, m0+CH+P] --------------------
            __syncthreads();
            for (int w = l; w < CH / 8; w += P) {
                const int r = m0 + 8 * w;
                s_q[g][w] = (have && r < qlen) ? qnib[r >> 3] : 0x44444444u;
            }
            for (int w = l; w < TW; w += P) {
                unsigned word = 0;
                if (have) {
#pragma unroll
                    for (int i = 0; i < 8; ++i)
                        word |= target_code(asm_words, lo + m0 + 8 * w + i, cstart, cend, runs, n_runs) << (4 * i);
                } else word = 0x55555555u;
                s_t[g][w] = word;
            }
            __syncthreads();
            if (m0 == 0) {  // initial window: lane l holds target codes x = l and l + 1
                t0 = nibble(s_t[g][l >> 3], l & 7);
                t1 = nibble(s_t[g][(l + 1) >> 3], (l + 1) & 7);
            }
            const int m_end = min(m0 + CH, max_steps);
            for (int m = m0; m < m_end; ++m) {
                const int k = m & 7;
                if (k == 0) {
                    qword = s_q[g][(m - m0) >> 3];
                    tword = s_t[g][(m - m0 + P) >> 3];
                }
                // ---- systolic sequence feeds ----------------------------------------------------------------------
                const unsigned q_in = nibble(qword, k);        // q[m] enters at lane 0
                const unsigned t_in = nibble(tword, k);        // target code x = m + P enters at lane P-1
                const unsigned q_shift = (unsigned)dpp_from_lower((int)qb);
                const unsigned t_shift = (unsigned)dpp_from_upper((int)t1);
                if (m > 0) {
                    t0 = t1;
                    t1 = (l == P - 1) ? t_in : t_shift;
                }
                qb = (l == 0) ? q_in : q_shift;
                const int r = m - l;
                const bool row_ok = (unsigned)r < (unsigned)qlen;

                // ---- cell A (band index 2l): left neighbour = lane l-1's B of the previous step --------------------
                int lh = dpp_from_lower(B.h), le = dpp_from_lower(B.e);
                unsigned lhlo = (unsigned)dpp_from_lower((int)B.hp_lo), lhhi = (unsigned)dpp_from_lower((int)B.hp_hi);
                unsigned lelo = (unsigned)dpp_from_lower((int)B.ep_lo), lehi = (unsigned)dpp_from_lower((int)B.ep_hi);
                if (l == 0) { lh = 0; le = NEG; }
                const int a_dh = A.h; const unsigned a_dlo = A.hp_lo, a_dhi = A.hp_hi;
                const int b_dh = B.h; const unsigned b_dlo = B.hp_lo, b_dhi = B.hp_hi;
                dp_cell(A, row_ok && t0 != 5u, r, 2 * l, qb, t0, a_dh, a_dlo, a_dhi, lh, le, lhlo, lhhi, lelo, lehi,
                        B.h, B.f, B.hp_lo, B.hp_hi, B.fp_lo, B.fp_hi, bestA);

                // ---- cell B (band index 2l+1): upper neighbour = lane l+1's A of this step -------------------------
                int uh = dpp_from_upper(A.h), uf = dpp_from_upper(A.f);
                unsigned uhlo = (unsigned)dpp_from_upper((int)A.hp_lo), uhhi = (unsigned)dpp_from_upper((int)A.hp_hi);
                unsigned uflo = (unsigned)dpp_from_upper((int)A.fp_lo), ufhi = (unsigned)dpp_from_upper((int)A.fp_hi);
                if (l == P - 1) { uh = 0; uf = NEG; }
                dp_cell(B, row_ok && t1 != 5u, r, 2 * l + 1, qb, t1, b_dh, b_dlo, b_dhi, A.h, A.e, A.hp_lo, A.hp_hi,
                        A.ep_lo, A.ep_hi, uh, uf, uhlo, uhhi, uflo, ufhi, bestB);
            }
        }

        // ---- best cell of the task: max score, then first row, then first column ---------------------------------
        int sc = bestA.score, er = bestA.end_r, eb = 2 * l;
        unsigned plo = bestA.p_lo, phi = bestA.p_hi;
        if (bestB.score > sc || (bestB.score == sc && bestB.end_r < er)) {
            sc = bestB.score; er = bestB.end_r; eb = 2 * l + 1; plo = bestB.p_lo; phi = bestB.p_hi;
        }
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

}  // namespace

void kp_launch_sw(const KpBatchView &b, const KpGenes &genes, const KpTask *tasks, const uint32_t *task_count,
                  uint32_t task_cap, const uint32_t *order, int width, KpSwResult *results, hipStream_t stream) {
    // persistent-style grid: enough single-wave blocks to fill 256 CUs several times over; each strides over quads
    const dim3 grid(256 * 16), block(64);
    if (width == 16)
        hipLaunchKernelGGL(kp_sw_kernel<8>, grid, block, 0, stream, b, genes, tasks, task_count, task_cap, order, results);
    else if (width == 32)
        hipLaunchKernelGGL(kp_sw_kernel<16>, grid, block, 0, stream, b, genes, tasks, task_count, task_cap, order, results);
    else if (width == 64)
        hipLaunchKernelGGL(kp_sw_kernel<32>, grid, block, 0, stream, b, genes, tasks, task_count, task_cap, order, results);
    else
        hipLaunchKernelGGL(kp_sw_kernel<64>, grid, block, 0, stream, b, genes, tasks, task_count, task_cap, order, results);
}
