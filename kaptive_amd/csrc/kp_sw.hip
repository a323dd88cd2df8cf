// kp_sw.hip -- banded local alignment (Smith-Waterman-Gotoh, affine gaps, int32) of every band task.
//
// Stands in for the extension half of rammappy's map_batch (reference call site src/kaptive/serotyping/core.py:154;
// fields consumed at src/kaptive/core/alignment.py:415-446).  Recurrence, tie rules and scores: include/kp_spec.h.
//
// Two kernels.  kp_sw_kernel fills the band and leaves, per task, the best cell and four direction bits per cell in a
// trace buffer in HBM (288 GB: a pass of 1000 assemblies writes ~12 GB of them); kp_sw_traceback_kernel walks the path
// of every task that reaches the score cut-off, one lane per task, and writes the hit coordinates, matches and columns.
// The fill kernel is bound by integer VALU issue (SQ counters: profiles/), so it carries nothing but scores: ~21 vector
// instructions per cell, against ~70 executed per cell by the round-1 kernel that carried start / matches / columns
// through the recurrence.
//
// Fill kernel mapping (wavefront-parallel anti-diagonals, no MFMA -- this is dependent integer DP, not a contraction):
//   * a task's band has W = 4P diagonals; P lanes own it, lane l holds the four adjacent diagonals 4l .. 4l+3 (cells
//     A..D); a 64-lane wave therefore runs 64/P tasks side by side (P = 4/8/16/32 for W = 16/32/64/128).
//   * time is skewed by lane: at step m lane l works on query row r = m - l, cells A, B, C, D in that order.  With
//     that skew A's left neighbour is lane l-1's D of the previous step, D's upper neighbour is lane l+1's A of the
//     same step, and every other neighbour is one of the lane's own registers -- two one-lane shifts per step, both
//     done with DPP (v_mov_b32_dpp row_shr:1 / row_shl:1, wave_* for the 32-lane class), all state stays in registers.
//   * sequences are streamed systolically: the query enters at lane 0 as a per-row score profile (five 6-bit signed
//     fields indexed by the target code, so a substitution score is one v_bfe_i32) and moves up one lane per step, the
//     target code enters at lane P-1 and moves down; both are staged per chunk in LDS and read four steps at a time.
//   * no boundary masks: rows outside the gene carry a profile of -32 in every field and columns outside the contig score
//     like N (-1).  Substitution scores <= 0 are all it takes: cells before the contig or above the gene then hold H = 0
//     and gap states <= -(open + ext), which is what kp_spec.h prescribes for their neighbours inside (H = 0, E = F = -inf
//     gives the same E, F and diagonal there); cells past the contig's end or below the gene only ever feed further such
//     cells, hold values strictly below the inside cell they derive from, so none becomes the best cell, and the
//     traceback, which only moves up and left, cannot reach them.
//   * the gap states are kept pre-charged (H - open - ext, E - ext, F - ext), so E and F of a neighbour are one max.
//   * best cell: per lane and cell one v_max_u32 on (score << 15 | 32767 - row): first maximum in row order for free.
//   * direction bits: the compares that decide a cell (open/extend for E and F, which of diagonal/E/F wins, "the diagonal
//     predecessor is a restart cell") stay in SGPR lane masks, are combined by scalar instructions, and each bit is
//     shifted into the lane's trace word by one v_addc_co_u32 (x + x + carry-in): 4 bits per cell, 16 bytes per lane
//     per 8 steps, written as one global_store_dwordx4 into the lane's own stream of the task's trace block.
#include "kp_internal.h"

namespace {

constexpr int CH = 64;  // steps staged per chunk (multiple of 8)
constexpr int NEG = KP_NEG_INF;
constexpr int OE = KP_GAP_OPEN + KP_GAP_EXT;
constexpr int EX = KP_GAP_EXT;
constexpr unsigned T_OUT = 24u;  // columns outside the contig read the N field: any score <= 0 does (see below)
// score profile of a query row: field t (6 bits, signed, at bit 6t) = score against target code t (0..3 ACGT, 4 = N)
constexpr unsigned PROF_N = 0x3FFFFFFFu;  // KP_SC_N in every field
constexpr unsigned PROF_MISMATCH = 0x3Cu | (0x3Cu << 6) | (0x3Cu << 12) | (0x3Cu << 18) | (0x3Fu << 24);
constexpr unsigned PROF_OUT = 0x20u | (0x20u << 6) | (0x20u << 12) | (0x20u << 18) | (0x20u << 24);  // -32 everywhere
static_assert(KP_SC_MATCH == 2 && KP_SC_MISMATCH == -4 && KP_SC_N == -1, "profile constants encode these scores");
static_assert(KP_MAX_GENE_LEN <= 32767, "the best-cell key holds the row in 15 bits and the score (<= 2 * length) in 16");

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

// acc = 2 * acc + (this lane's bit of `mask`): one VALU instruction, the mask stays in SGPRs
__device__ __forceinline__ void push_bit(unsigned &acc, unsigned long long mask) {
    unsigned long long carry_out;
    asm("v_addc_co_u32 %0, %1, %2, %2, %3" : "=v"(acc), "=s"(carry_out) : "v"(acc), "s"(mask));
}

struct Cell {
    int h, hmoe, emex, fmex;  // H, H - (open + ext), E - ext, F - ext
    unsigned best;            // max over the rows so far of (H << 15) | (32767 - row)
};

// Direction nibble of a cell, most significant bit first: [source:2][E opened][F opened];
// source 0 = diagonal, 1 = diagonal from a restart cell (the path starts here), 2 = E, 3 = F.
// e / f: the gap states arriving from the left / from above, eo / fo: lane masks "the gap was opened there" (open wins
// ties) -- computed by the caller, which also knows the band's edge lanes.
// (Measured and rejected: one bit plane per compare, shifted in without the scalar mask arithmetic in between -- one more
// vector instruction per cell, 14.8 ms against 13.5 ms per pass.)
__device__ __forceinline__ void dp_cell(Cell &c, unsigned &acc, unsigned inv_r, unsigned prof, unsigned tsh, int e,
                                        unsigned long long eo, int f, unsigned long long fo) {
    const int s = __builtin_amdgcn_sbfe((int)prof, tsh, 6u);
    const unsigned long long fresh = __builtin_amdgcn_ballot_w64(c.h == 0);
    const int d = c.h + s;  // diagonal: the lane's own previous row
    const int m = max(max(d, e), f);
    const unsigned long long from_d = __builtin_amdgcn_ballot_w64(d == m);  // the diagonal wins ties, then E
    const unsigned long long from_e = __builtin_amdgcn_ballot_w64(e == m);
    const int h = max(m, 0);
    c.h = h; c.hmoe = h - OE; c.emex = e - EX; c.fmex = f - EX;
    c.best = max(c.best, ((unsigned)h << 15) | inv_r);
    push_bit(acc, ~from_d);
    push_bit(acc, (from_d & fresh) | ~(from_d | from_e));
    push_bit(acc, eo);
    push_bit(acc, fo);
}

struct State {
    Cell A, B, C, D;
    unsigned qb, t0, t1, t2, t3;
};

template <int P>
__device__ __forceinline__ void dp_step(State &s, unsigned &acc, int m, int l, unsigned prof_in, unsigned t_in,
                                        unsigned long long first_lanes, unsigned long long last_lanes) {
    constexpr bool ROW = P <= 16;
    const unsigned q_shift = (unsigned)from_lower<ROW>((int)s.qb);
    s.qb = (l == 0) ? prof_in : q_shift;  // profile of row m enters at lane 0
    const unsigned inv_r = (unsigned)(32767 - (m - l)) & 32767u;

    // A: the left neighbour is lane l-1's D of the previous step; left of the band's first diagonal H = 0, E = -inf, so
    // the band's first lane takes E = -(open + ext), "opened" (one select on the result instead of one per operand)
    const int l_hmoe = from_lower<ROW>(s.D.hmoe), l_gmex = from_lower<ROW>(s.D.emex);
    const int eA = (l == 0) ? -OE : max(l_hmoe, l_gmex);
    const unsigned long long eoA = __builtin_amdgcn_ballot_w64(l_hmoe >= l_gmex) | first_lanes;
    dp_cell(s.A, acc, inv_r, s.qb, s.t0, eA, eoA, max(s.B.hmoe, s.B.fmex), __builtin_amdgcn_ballot_w64(s.B.hmoe >= s.B.fmex));
    dp_cell(s.B, acc, inv_r, s.qb, s.t1, max(s.A.hmoe, s.A.emex), __builtin_amdgcn_ballot_w64(s.A.hmoe >= s.A.emex),
            max(s.C.hmoe, s.C.fmex), __builtin_amdgcn_ballot_w64(s.C.hmoe >= s.C.fmex));
    dp_cell(s.C, acc, inv_r, s.qb, s.t2, max(s.B.hmoe, s.B.emex), __builtin_amdgcn_ballot_w64(s.B.hmoe >= s.B.emex),
            max(s.D.hmoe, s.D.fmex), __builtin_amdgcn_ballot_w64(s.D.hmoe >= s.D.fmex));
    // D: the upper neighbour is lane l+1's A of this step; above the band's last diagonal the same boundary applies
    const int u_hmoe = from_upper<ROW>(s.A.hmoe), u_gmex = from_upper<ROW>(s.A.fmex);
    const int fD = (l == P - 1) ? -OE : max(u_hmoe, u_gmex);
    const unsigned long long foD = __builtin_amdgcn_ballot_w64(u_hmoe >= u_gmex) | last_lanes;
    dp_cell(s.D, acc, inv_r, s.qb, s.t3, max(s.C.hmoe, s.C.emex), __builtin_amdgcn_ballot_w64(s.C.hmoe >= s.C.emex), fD, foD);

    const unsigned t_shift = (unsigned)from_upper<ROW>((int)s.t1);  // lane l+1's x = m + 3l + 4 = this lane's next t3
    s.t0 = s.t1; s.t1 = s.t2; s.t2 = s.t3;
    s.t3 = (l == P - 1) ? t_in : t_shift;  // target code x = m + 1 + 3P enters at lane P-1
}

constexpr int PROF_WORDS = 16 * CH;                      // LDS of one block: profiles of 64/P groups x CH rows (P >= 4)
constexpr int TCODE_BYTES = 16 * (CH + 3 * 4 + 4 + 4);   // ... and their target codes (largest for P = 4), rows padded
constexpr int TWORD_WORDS = 128;                         // ... and the packed words those codes are cut from

// all tasks of one band class, seen from block `block` of `n_blocks` that work on the class
template <int P>
__device__ __forceinline__ void sw_class(const KpBatchView &b, const KpGenes &genes, const KpTask *__restrict__ tasks,
                                         uint32_t n_tasks, const uint32_t *__restrict__ order,
                                         KpSwEnd *__restrict__ ends, uint4 *__restrict__ trace,
                                         unsigned long long *__restrict__ trace_top, uint64_t trace_cap, uint32_t block,
                                         uint32_t n_blocks, uint32_t *s_prof_raw, uint8_t *s_t_raw, uint32_t *s_tw_raw) {
    constexpr int G = 64 / P;
    constexpr int TW = CH + 3 * P + 4;  // staged target codes per chunk: window x in [m0, m0 + CH + 3P]
    // the codes a step pulls in start at x = step + 3P + 1: the row is shifted by PAD so that every fourth step's lies
    // on a 4-byte boundary (one ds_read_b32 feeds four steps)
    constexpr int PAD = (4 - (3 * P + 1) % 4) % 4;
    constexpr int TROW = (TW + PAD + 3) & ~3;
    static_assert(G * CH <= PROF_WORDS && G * TROW <= TCODE_BYTES, "LDS carve-up");
    uint32_t(*s_prof)[CH] = reinterpret_cast<uint32_t(*)[CH]>(s_prof_raw);
    uint8_t(*s_t)[TROW] = reinterpret_cast<uint8_t(*)[TROW]>(s_t_raw);

    const int lane = threadIdx.x;
    const int g = lane / P, l = lane % P;
    const unsigned long long first_lanes = __builtin_amdgcn_ballot_w64(l == 0), last_lanes = __builtin_amdgcn_ballot_w64(l == P - 1);

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
        // 8-step trace pieces per lane (an even number: task blocks then start on 128-byte lines)
        const int n_chunks = (((steps + 7) >> 3) + 1) & ~1;
        int max_steps = steps;
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) max_steps = max(max_steps, __shfl_xor(max_steps, o));
        // the task's trace block: P lane streams of n_chunks 16-byte pieces each
        unsigned long long toff = 0;
        if (have && l == 0) toff = atomicAdd(trace_top, (unsigned long long)(P * n_chunks));
        toff = ((unsigned long long)__shfl((unsigned)(toff >> 32), g * P) << 32) | __shfl((unsigned)toff, g * P);
        const bool fits = have && toff + (unsigned long long)(P * n_chunks) <= trace_cap;  // else: counted, host reruns
        // piece j of lane l at [j][l]: the P lanes of a task write P * 16 contiguous bytes per store, and a 128-byte line
        // is complete after 8 * 8 / P steps -- with a stream per lane a line stayed open for 64 steps, more open lines
        // than the L2 holds, and HBM saw three times the bytes (WRITE_SIZE, profiles/)
        uint4 *my_trace = trace + toff + l;

        State st;
        st.A.h = 0; st.A.hmoe = -OE; st.A.emex = NEG; st.A.fmex = NEG; st.A.best = 0;
        st.B = st.A; st.C = st.A; st.D = st.A;
        st.qb = PROF_OUT; st.t0 = st.t1 = st.t2 = st.t3 = T_OUT;
        unsigned acc[4] = {0, 0, 0, 0};
        bool saw_n = false;  // an N in the gene or in the target window: the traceback then compares bases itself

        const int steps8 = (max_steps + 7) & ~7;
        // Staging of a chunk (profiles of its query rows, codes of its target window) works from packed words that were
        // requested one chunk earlier: NQ gene words (8 rows each) and NT assembly words (16 bases each) per lane.
        constexpr int NQ = (CH / 8 + P - 1) / P, NTW = (TW + 15) / 16 + 1, NT = (NTW + P - 1) / P;
        static_assert(G * NTW <= TWORD_WORDS, "LDS carve-up");
        uint32_t(*s_tw)[NTW] = reinterpret_cast<uint32_t(*)[NTW]>(s_tw_raw);
        const int64_t asm_n_words = b.asm_word_off[tk.asm_id + 1] - b.asm_word_off[tk.asm_id];
        uint32_t qreg[NQ], treg[NT];
        auto request = [&](int m0) {  // global loads only; nothing waits for them here
#pragma unroll
            for (int i = 0; i < NQ; ++i) {
                const int w = l + i * P, r = m0 + 8 * w;
                qreg[i] = (have && w < CH / 8 && r < qlen) ? qnib[r >> 3] : 0u;
            }
            const int w0 = (lo + m0) >> 4;
#pragma unroll
            for (int i = 0; i < NT; ++i) {
                const int64_t wi = (int64_t)w0 + l + i * P;
                treg[i] = (have && l + i * P < NTW && wi >= 0 && wi < asm_n_words) ? asm_words[wi] : 0u;
            }
        };
        request(0);
        for (int m0 = 0; m0 < steps8; m0 += CH) {
            // ---- stage this chunk: profiles of query rows [m0, m0+CH), target codes of window x in [m0, m0+CH+3P] ----
            __syncthreads();
#pragma unroll
            for (int i = 0; i < NQ; ++i) {
                const int w = l + i * P, r = m0 + 8 * w;
                if (w < CH / 8) {
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const bool live = have && r + j < qlen;
                        s_prof[g][8 * w + j] = live ? row_profile(nibble(qreg[i], j)) : PROF_OUT;
                        saw_n |= live && nibble(qreg[i], j) >= 4u;
                    }
                }
            }
#pragma unroll
            for (int i = 0; i < NT; ++i)
                if (l + i * P < NTW) s_tw[g][l + i * P] = treg[i];
            __syncthreads();
            {
                // codes of the window, four at a time (one 32-bit LDS store): byte y of the padded row = position
                // p0 + y - PAD.  Whole groups inside the contig of an assembly without N runs -- nearly all -- are cut
                // out of the packed words with one funnel shift and spread to bytes; the rest go base by base.
                const int p0 = lo + m0, w0 = p0 >> 4;
                for (int y = 4 * l; y < TROW; y += 4 * P) {
                    const int t0 = p0 + y - PAD;
                    uint32_t four;
                    if (have && n_runs == 0 && y >= PAD && t0 >= cstart && t0 + 3 < cend) {
                        const int wi = (t0 >> 4) - w0;
                        const uint32_t lo_w = s_tw[g][wi], hi_w = wi + 1 < NTW ? s_tw[g][wi + 1] : 0u;
                        const uint32_t v = __builtin_amdgcn_alignbit(hi_w, lo_w, 2 * (t0 & 15)) & 255u;
                        four = ((v & 3u) | ((v & 0xCu) << 6) | ((v & 0x30u) << 12) | ((v & 0xC0u) << 18)) * 6u;
                    } else {
                        four = 0;
                        for (int i = 0; i < 4; ++i) {
                            const int t = t0 + i;
                            unsigned code = 5u;
                            if (have && t >= p0 && t >= cstart && t < cend) {  // (bytes before the window are never read)
                                code = (s_tw[g][(t >> 4) - w0] >> (2 * (t & 15))) & 3u;
                                if (n_runs > 0) {  // rare: assemblies with scaffold gaps
                                    int a = 0, z = n_runs;
                                    while (a < z) {
                                        const int mid = (a + z) >> 1;
                                        if (runs[2 * mid + 1] <= t) a = mid + 1; else z = mid;
                                    }
                                    if (a < n_runs && runs[2 * a] <= t) code = 4u;
                                }
                            }
                            four |= (code < 5u ? 6u * code : T_OUT) << (8 * i);
                            saw_n |= code == 4u;
                        }
                    }
                    *reinterpret_cast<uint32_t *>(&s_t[g][y]) = four;
                }
            }
            __syncthreads();
            if (m0 + CH < steps8) request(m0 + CH);  // the next chunk's words arrive while this chunk is computed
            if (m0 == 0) {  // initial window: cell k of lane l sits on x = 3l + k
                st.t0 = s_t[g][PAD + 3 * l]; st.t1 = s_t[g][PAD + 3 * l + 1]; st.t2 = s_t[g][PAD + 3 * l + 2];
                st.t3 = s_t[g][PAD + 3 * l + 3];
            }
            const int m_end = min(m0 + CH, steps8);
            for (int m = m0; m < m_end; m += 8) {
#pragma unroll
                for (int half = 0; half < 2; ++half) {
                    const int mm = m + 4 * half;
                    const uint4 pr = *reinterpret_cast<const uint4 *>(&s_prof[g][mm - m0]);
                    const uint32_t tc = *reinterpret_cast<const uint32_t *>(&s_t[g][mm - m0 + 3 * P + 1 + PAD]);
                    dp_step<P>(st, acc[2 * half], mm, l, pr.x, tc & 255u, first_lanes, last_lanes);
                    dp_step<P>(st, acc[2 * half], mm + 1, l, pr.y, (tc >> 8) & 255u, first_lanes, last_lanes);
                    dp_step<P>(st, acc[2 * half + 1], mm + 2, l, pr.z, (tc >> 16) & 255u, first_lanes, last_lanes);
                    dp_step<P>(st, acc[2 * half + 1], mm + 3, l, pr.w, tc >> 24, first_lanes, last_lanes);
                }
                const int j = m >> 3;
                if (fits && j < n_chunks) my_trace[(size_t)j * P] = make_uint4(acc[0], acc[1], acc[2], acc[3]);
            }
        }

        // ---- best cell of the task: max score, then first row, then first column ---------------------------------
        unsigned key = st.A.best;
        int eb = 4 * l;
        if (st.B.best > key) { key = st.B.best; eb = 4 * l + 1; }
        if (st.C.best > key) { key = st.C.best; eb = 4 * l + 2; }
        if (st.D.best > key) { key = st.D.best; eb = 4 * l + 3; }
#pragma unroll
        for (int o = 1; o < P; o <<= 1) {
            const unsigned key2 = (unsigned)__shfl_xor((int)key, o);
            const int eb2 = __shfl_xor(eb, o);
            if (key2 > key || (key2 == key && eb2 < eb)) { key = key2; eb = eb2; }
            saw_n |= __shfl_xor((int)saw_n, o) != 0;
        }
        if (have && l == 0) {
            KpSwEnd out;
            out.score = fits ? (int)(key >> 15) : 0;
            out.er = 32767 - (int)(key & 32767u);
            out.eb = eb | (saw_n ? KP_SWEND_HAS_N : 0);
            out.trace_off = (uint32_t)toff;
            ends[ti] = out;
        }
    }
}

// One launch for all four band classes.  The wide classes hold few tasks but each of their waves runs a long step chain
// (a launch of its own costs ~1 ms of latency at the end of the pass), so they get the first blocks of the grid and
// run underneath the 16-diagonal class that fills the chip.
constexpr uint32_t WIDE_BLOCKS = 512;  // blocks per wide class (they stride over their quads)

#ifndef KP_SW_WAVES
#define KP_SW_WAVES 5  // waves per SIMD the register budget is set for (96 VGPRs; measured against 4, 6 and 8: profiles/)
#endif
__global__ __launch_bounds__(64, KP_SW_WAVES) void kp_sw_kernel(KpBatchView b, KpGenes genes, const KpTask *__restrict__ tasks,
                                                   const uint32_t *__restrict__ task_count, uint32_t task_cap,
                                                   const uint32_t *__restrict__ order, KpSwEnd *__restrict__ ends,
                                                   uint4 *__restrict__ trace, unsigned long long *__restrict__ trace_top,
                                                   uint64_t trace_cap) {
    __shared__ __attribute__((aligned(16))) uint32_t s_prof[PROF_WORDS];
    __shared__ __attribute__((aligned(16))) uint8_t s_t[TCODE_BYTES];
    __shared__ uint32_t s_tw[TWORD_WORDS];
    // class c (0..3 = 16/32/64/128 diagonals): tasks, order and results at c * task_cap, count at task_count[c]
    const uint32_t blk = blockIdx.x;
    const int c = blk < 3 * WIDE_BLOCKS ? 3 - (int)(blk / WIDE_BLOCKS) : 0;
    const size_t off = (size_t)c * task_cap;
    uint32_t n = task_count[c];
    if (n > task_cap) n = task_cap;
    if (c == 3) sw_class<32>(b, genes, tasks + off, n, order + off, ends + off, trace, trace_top, trace_cap, blk, WIDE_BLOCKS, s_prof, s_t, s_tw);
    else if (c == 2) sw_class<16>(b, genes, tasks + off, n, order + off, ends + off, trace, trace_top, trace_cap, blk - WIDE_BLOCKS, WIDE_BLOCKS, s_prof, s_t, s_tw);
    else if (c == 1) sw_class<8>(b, genes, tasks + off, n, order + off, ends + off, trace, trace_top, trace_cap, blk - 2 * WIDE_BLOCKS, WIDE_BLOCKS, s_prof, s_t, s_tw);
    else sw_class<4>(b, genes, tasks + off, n, order + off, ends + off, trace, trace_top, trace_cap, blk - 3 * WIDE_BLOCKS, gridDim.x - 3 * WIDE_BLOCKS, s_prof, s_t, s_tw);
}

// ---- traceback: one lane per task -------------------------------------------------------------------------------------------
// Cell (row r, band index bi) sits on target position lo + r + bi; a diagonal step keeps bi, a step to the left (E, gap
// in the query) lowers it, a step up (F, gap in the target) raises it.  The nibble of (r, bi) is in lane stream bi / 4,
// step r + bi / 4: piece (step / 8), word (step % 8) / 2, upper half for even steps, cell A first.
//
// A path runs along a diagonal most of the time: it stays in one lane stream and walks it backwards.  The walk therefore
// works on whole 16-byte pieces (8 steps x 4 cells) held in registers: when it stands on the last step of a piece and
// all eight nibbles of its cell say "diagonal, not the start", it takes the eight steps at once; everything else (gaps,
// the first and last steps of a path, tasks with an N, whose matches are counted base by base) goes step by step from
// the same registers.  The piece after the current one is requested a whole piece ahead, so the 160 or so dependent loads
// of a path overlap with other waves' work; the direction bits are read about once (a quarter of what the fill wrote).
// Matches: without an N in the gene or the window every diagonal step scores +2 or -4, so
// score = 6 * matches - 4 * diagonal_steps - gap_costs gives the matches in closed form; tasks that saw an N (flagged by
// the fill kernel) compare the bases of every diagonal step instead.
constexpr int TB_THREADS = 256;

__device__ __forceinline__ uint32_t piece_word(const uint4 &v, int w) {  // word w of a piece, w in registers' terms
    const uint32_t lo = (w & 1) ? v.y : v.x, hi = (w & 1) ? v.w : v.z;
    return (w & 2) ? hi : lo;
}

__global__ __launch_bounds__(TB_THREADS) void kp_sw_traceback_kernel(KpBatchView b, KpGenes genes, const KpTask *__restrict__ tasks,
                                                              const uint32_t *__restrict__ task_count, uint32_t task_cap,
                                                              const uint32_t *__restrict__ order,
                                                              const KpSwEnd *__restrict__ ends,
                                                              const uint32_t *__restrict__ trace,
                                                              KpSwResult *__restrict__ results) {
    const int cls = blockIdx.y;
    uint32_t n = task_count[cls];
    if (n > task_cap) n = task_cap;
    const int P = 4 << cls;
    const uint32_t n_iter = (n + 63u) & ~63u;  // whole waves iterate together
    for (uint32_t slot = blockIdx.x * blockDim.x + threadIdx.x; slot < n_iter; slot += gridDim.x * blockDim.x) {
        const bool have = slot < n;
        const uint32_t ti = have ? order[(size_t)cls * task_cap + slot] : 0u;  // neighbours in this order have similar lengths
        const size_t at = (size_t)cls * task_cap + ti;
        KpSwEnd e;
        e.score = 0; e.er = 0; e.eb = 0; e.trace_off = 0;
        if (have) e = ends[at];
        KpSwResult out;
        out.score = e.score; out.q_start = out.q_end = out.t_start = out.t_end = out.matches = out.block_len = 0;
        bool walking = have && e.score >= KP_MIN_DP_SCORE;  // the others are dropped by the hit filter anyway
        KpTask tk;
        tk.asm_id = 0; tk.gs = 0; tk.lo = 0;
        if (walking) tk = tasks[at];
        const int gene = tk.gs >> 1;
        const int qlen = walking ? genes.len[gene] : 0;
        const bool has_n = (e.eb & KP_SWEND_HAS_N) != 0;
        const int eb = e.eb & 255;
        const uint32_t *qnib = genes.nib + genes.word_off[(tk.gs & 1) ? genes.n_genes + gene : gene];
        const uint32_t *asm_words = b.words + b.asm_word_off[tk.asm_id];
        const int r0 = b.asm_first_nrun[tk.asm_id];
        const int n_runs = b.asm_first_nrun[tk.asm_id + 1] - r0;
        const int32_t *runs = b.n_runs + 2 * (size_t)r0;
        const uint4 *tw = reinterpret_cast<const uint4 *>(trace) + e.trace_off;  // piece j of lane l at [j * P + l]
        int r = e.er, bi = eb, state = 0, cols = 0, matches = 0, diag = 0, gap_cost = 0, gap = 0, credit = 0;
        int sr = r, sb = bi;
        // cur = the piece the walk stands in, nxt = the one before it in the same lane stream (requested ahead)
        uint4 cur = make_uint4(0, 0, 0, 0), nxt = cur;
        int cur_tag = -1, nxt_tag = -1;  // (stream << 20) | piece index
        while (__any(walking)) {
            if (!walking) continue;
            const int l = bi >> 2, k = bi & 3, step = r + l;
            const int pc = step >> 3, tag = (l << 20) | pc;
            if (tag != cur_tag) {
                const uint4 *stream = tw + l;
                if (tag == nxt_tag) cur = nxt;
                else cur = stream[(size_t)pc * P];
                cur_tag = tag;
                if (pc > 0) { nxt = stream[(size_t)(pc - 1) * P]; nxt_tag = tag - 1; }  // used one piece from now at the earliest
            }
            if (state == 0 && !has_n && (step & 7) == 7) {
                // src bits (the nibble's upper two) of cell k in both steps of a word
                const uint32_t pure = (0xC000C000u >> (4 * k));
                if (((cur.x | cur.y | cur.z | cur.w) & pure) == 0u) {  // eight plain diagonal steps
                    cols += 8; diag += 8; r -= 8;
                    continue;
                }
            }
            const uint32_t word = piece_word(cur, (step & 7) >> 1);
            const uint32_t nib = (word >> ((step & 1 ? 0 : 16) + 12 - 4 * k)) & 15u;
            const uint32_t src = nib >> 2;
            if (state == 0) {
                if (src <= 1u) {  // diagonal: one column
                    ++cols; ++diag;
                    if (has_n) {  // a match when both bases are the same unambiguous base
                        const int t = tk.lo + r + bi;
                        const uint32_t qc = nibble(qnib[r >> 3], r & 7);
                        uint32_t tc = (asm_words[t >> 4] >> (2 * (t & 15))) & 3u;
                        if (n_runs > 0) {
                            int lo = 0, hi = n_runs;
                            while (lo < hi) {
                                const int mid = (lo + hi) >> 1;
                                if (runs[2 * mid + 1] <= t) lo = mid + 1; else hi = mid;
                            }
                            if (lo < n_runs && runs[2 * lo] <= t) tc = 4u;
                        }
                        matches += (qc == tc && qc < 4u) ? 1 : 0;  // N against N scores KP_SC_N: not a match
                    }
                    if (src == 1u) { sr = r; sb = bi; walking = false; }
                    --r;
                } else {
                    state = (int)src - 1;  // 1 = E, 2 = F: the gap's columns are counted in that state
                }
            } else if (state == 1) {  // E: gap in the query; this cell's E came from H (opened) or E (extended) of the left cell
                ++cols; ++gap; gap_cost += EX;
                --bi;
                if (nib & 2u) { state = 0; gap_cost += KP_GAP_OPEN; credit += max(gap - KP_GAP_LONG, 0); gap = 0; }
            } else {  // F: gap in the target
                ++cols; ++gap; gap_cost += EX;
                --r; ++bi;
                if (nib & 1u) { state = 0; gap_cost += KP_GAP_OPEN; credit += max(gap - KP_GAP_LONG, 0); gap = 0; }
            }
        }
        if (!have) continue;
        if (e.score < KP_MIN_DP_SCORE) { results[at] = out; continue; }
        if (!has_n) matches = (e.score + 4 * diag + gap_cost) / 6;
        out.score = e.score + credit;  // the path under the two-piece gap cost (kp_spec.h): long gaps get their credit
        out.q_start = sr; out.q_end = e.er + 1;
        out.t_start = sr + tk.lo + sb; out.t_end = e.er + tk.lo + eb + 1;
        out.matches = matches; out.block_len = cols;
        results[at] = out;
    }
}

}  // namespace

void kp_launch_sw(const KpBatchView &b, const KpGenes &genes, const KpTask *tasks, const uint32_t *task_count,
                  uint32_t task_cap, const uint32_t *order, KpSwEnd *ends, void *trace, unsigned long long *trace_top,
                  uint64_t trace_cap_units, KpSwResult *results, int blocks_per_cu, hipStream_t stream,
                  hipEvent_t after_fill) {
    // many short-lived single-wave blocks (each strides over a quad or two): CU slots turn over every few hundred
    // microseconds, so the tail is even and the high-priority streams of other batches' reductions get their turn
    const dim3 grid(3 * WIDE_BLOCKS + 256 * (unsigned)(blocks_per_cu > 0 ? blocks_per_cu : 256)), block(64);
    hipLaunchKernelGGL(kp_sw_kernel, grid, block, 0, stream, b, genes, tasks, task_count, task_cap, order, ends,
                       reinterpret_cast<uint4 *>(trace), trace_top, trace_cap_units);
    if (after_fill) (void)hipEventRecord(after_fill, stream);
    hipLaunchKernelGGL(kp_sw_traceback_kernel, dim3(2048, KP_N_CLASSES), dim3(TB_THREADS), 0, stream, b, genes, tasks, task_count,
                       task_cap, order, ends, reinterpret_cast<const uint32_t *>(trace), results);
}
