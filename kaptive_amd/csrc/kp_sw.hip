// kp_sw.hip -- banded local alignment (Smith-Waterman-Gotoh, affine gaps) of every band task.
//
// Stands in for the extension half of rammappy's map_batch (reference call site src/kaptive/serotyping/core.py:154;
// fields consumed at src/kaptive/core/alignment.py:415-446).  Recurrence, tie rules and scores: include/kp_spec.h.
//
// Two kernels.  kp_sw_kernel fills the band and leaves, per task, the best cell and four direction bits per cell in a
// trace buffer in HBM (288 GB: a pass of 1000 assemblies writes ~12 GB of them); kp_sw_traceback_kernel walks the path
// of every task that reaches the score cut-off, one lane per task, and writes the hit coordinates, matches and columns.
//
// What the fill kernel is built around (measured on the MI355X, tools/microbench/valu_rate*.hip, profiles/valu_rate*_r2.txt):
// a gfx950 SIMD issues a wave64 v_add_u32 / v_sub_u32 / v_and / v_or / v_xor / v_lshrrev / v_mov (VGPR or constant
// operands) every 2.35 cycles, but v_max / v_min, every compare, v_addc, v_cndmask, v_bfe, every VOP3 and VOP3P (packed
// 16-bit) instruction, DPP and SDWA forms, and anything with an SGPR operand every 4.  The round's first fill kernel (int32
// scores, one compare plus one carry push per direction bit) priced at 82 cycles per cell of a lane by that table and ran
// at exactly that.
// This one keeps TWO tasks in every register -- task X in the low 16 bits, task Y in the high 16 -- so that
//   * additions and subtractions of scores are one cheap v_add_u32 for two cells (all values are kept biased into
//     [0, 32767], so no carry or borrow ever crosses bit 16),
//   * maxima are one v_pk_max_u16 -- or, three ways, one v_pk_maximum3_f16 (see pk_max3) -- for two cells,
//   * a comparison a >= b is (a | 0x80008000) - b: two cheap instructions leave the answer for both cells in bits 15
//     and 31 (the guard bit survives exactly when no borrow reaches it) -- no v_cmp, no lane mask, no carry push,
//   * the four direction bits of a cell are gathered from those words byte-wise (v_perm_b32, one v_bfi),
//   * every cross-lane move (DPP) and every band-edge select serves two cells.
// 41.8 issue cycles per cell of a lane (2674 per 8 steps of 8 cells, tools/isa_cost.py; round 2: 52) instead of 82.
//
// Mapping (wavefront-parallel anti-diagonals, no MFMA -- this is dependent integer DP, not a contraction):
//   * a task's band has W = 4P diagonals; P lanes own a PAIR of tasks of that width (neighbours in the length-ordered task
//     list), lane l holds the four adjacent diagonals 4l .. 4l+3 (cells A..D) of both; a wave runs 2 * 64/P tasks.
//   * time is skewed by lane: at step m lane l works on query row r = m - l, cells A, B, C, D in that order.  With
//     that skew A's left neighbour is lane l-1's D of the previous step, D's upper neighbour is lane l+1's A of the
//     same step, and every other neighbour is one of the lane's own registers -- two one-lane shifts per step, both
//     done with DPP (v_mov_b32_dpp row_shr:1 / row_shl:1, wave_* for the 32-lane class), all state stays in registers.
//   * sequences: the query is a per-row score profile (five 3-bit fields per task, score + 4, indexed by the target code:
//     a substitution score is one v_pk_lshrrev_b16 and one v_and for two cells), the target a shift amount 3 * code per
//     column; both are staged per chunk in LDS -- the profiles ready-made from the database's stream of them
//     (KpGenes::prof), the shift amounts cut from the packed words four at a time and spread by a 256-entry table -- and
//     every lane reads its own row and column there (the LDS pipe is idle otherwise).
//   * no boundary masks: rows outside the gene carry a profile of -4 in every field and columns outside the contig score
//     like N (-1).  Substitution scores <= 0 are all it takes: cells before the contig or above the gene then hold H = 0
//     and gap states <= -(open + ext), which is what kp_spec.h prescribes for their neighbours inside (H = 0, E = F = -inf
//     gives the same E, F and diagonal there); cells past the contig's end or below the gene only ever feed further such
//     cells, hold values strictly below the inside cell they derive from, so none becomes the best cell, and the
//     traceback, which only moves up and left, cannot reach them.
//   * only the rows of a band that can touch the contig are filled (kp_task_rows, kp_internal.h): a gene that runs off
//     a contig end costs the rows that are on the contig; steps, trace pieces and the traceback count from that first row.
//   * biases: the pre-charged gap states (H - open - ext, E - ext, F - ext) are kept as value + 12 and a cell's three
//     candidates are compared two below their value (d - 2 is the predecessor's H - open - ext plus the score + 4; e - 2
//     and f - 2 are the cell's own E - ext and F - ext), so H itself is never formed.  The smallest value the recurrence can
//     produce is -(open + 2 ext) = -8, and "no gap yet" is represented by exactly that (it loses every maximum it takes
//     part in, like -inf), so everything is >= 0; the largest is 2 * length + 12 < 32768 (KP_FILL16_MAX_GENE_LEN).
//   * the restart at 0 costs nothing: the gap state a cell hands to its right neighbour is floored at 0 inside the
//     three-way maximum that forms it, so one of a cell's three candidates is always >= 0 (dp_cell).
//   * best cell: per lane and task the running maximum of its cells, the PAIR of steps in which it last rose and the eight
//     H of that pair (packed three-way maxima, one sign compare, v_pk_ashrrev_i16 to a half-word mask, v_bfi selects):
//     first maximum in row order, then the first of the lane's cells that holds it.
//   * direction bits, four steps per 16 bits, eight steps per 32-bit word and cell: the two tasks' halves are
//     separated with v_perm_b32 every eight steps and each task's 16 bytes go to its own trace block.
#include "kp_internal.h"

namespace {

constexpr int CH = 64;  // steps staged per chunk (multiple of 8)
#ifndef KP_TRACE_GROUP
#define KP_TRACE_GROUP 2
#endif
constexpr int TG = KP_TRACE_GROUP;  // consecutive 8-step trace pieces of a lane that are contiguous in memory (2 or 4)
constexpr int OE = KP_GAP_OPEN + KP_GAP_EXT;
constexpr int EX = KP_GAP_EXT;
static_assert(KP_SC_MATCH == 2 && KP_SC_MISMATCH == -4 && KP_SC_N == -1 && OE == 6 && EX == 2,
              "the profile fields and the biases below encode these scores");
static_assert(2 * KP_FILL16_MAX_GENE_LEN + 14 < 0x7C00, "biased scores must stay below the half-precision infinity pattern (pk_max3) "
                                                 "and leave bit 15 of a half-word free for the guard");

constexpr unsigned K1 = 0x00010001u;                        // a value in both halves: x * K1
constexpr unsigned GUARD = 0x80008000u;
constexpr unsigned CB = 12;                                 // bias of the gap states and of H - (open + ext)
constexpr unsigned ZB = CB - OE;                            // H = 0 as the kernel holds it: H - (open + ext) + 12
constexpr unsigned GAP_NONE = (CB - OE - EX) * K1;          // "no gap": -(open + 2 ext), biased
constexpr unsigned GAP_EDGE = (CB - OE) * K1;               // the gap state a band-edge cell sees: opened from H = 0
constexpr unsigned T_OUT = 12u;                             // shift amount of the N field: columns outside the contig
// profile of a query row, per task 15 bits: field t (3 bits at 3t) = score against target code t (0..3 ACGT, 4 = N) + 4
// (kp_row_profile, kp_internal.h: the database holds every gene as a stream of them)
constexpr unsigned PROF_OUT = 0u;  // -4 everywhere
__device__ __forceinline__ unsigned nibble(unsigned word, int i) { return (word >> (4 * i)) & 15u; }

// One-lane shifts.  The lane without a source reads 0 (bound_ctrl); every group-edge lane overrides what it receives
// anyway.  Groups of up to 16 lanes never straddle a DPP row, so the row shifts do; the 32-lane class needs wave shifts.
template <bool ROW>
__device__ __forceinline__ unsigned from_lower(unsigned v) {  // lane i <- lane i-1
    return (unsigned)(ROW ? __builtin_amdgcn_mov_dpp((int)v, 0x111 /*row_shr:1*/, 0xf, 0xf, true)
                          : __builtin_amdgcn_mov_dpp((int)v, 0x138 /*wave_shr:1*/, 0xf, 0xf, true));
}
template <bool ROW>
__device__ __forceinline__ unsigned from_upper(unsigned v) {  // lane i <- lane i+1
    return (unsigned)(ROW ? __builtin_amdgcn_mov_dpp((int)v, 0x101 /*row_shl:1*/, 0xf, 0xf, true)
                          : __builtin_amdgcn_mov_dpp((int)v, 0x130 /*wave_shl:1*/, 0xf, 0xf, true));
}


// The instructions of the cell, spelled out so that constants stay literals of 2-cycle VOP2 encodings (an SGPR operand
// makes them 4-cycle) and packed operations are not taken apart.  Plain asm (not volatile): the compiler schedules them.
template <unsigned LIT>
__device__ __forceinline__ unsigned add_k(unsigned a) {  // a + LIT (also a - x as a + (2^32 - x): same 32-bit result)
    unsigned r;
    asm("v_add_u32_e32 %0, %2, %1" : "=v"(r) : "v"(a), "n"(LIT));
    return r;
}
template <unsigned LIT>
__device__ __forceinline__ unsigned or_k(unsigned a) {
    unsigned r;
    asm("v_or_b32_e32 %0, %2, %1" : "=v"(r) : "v"(a), "n"(LIT));
    return r;
}
template <unsigned LIT>
__device__ __forceinline__ unsigned and_k(unsigned a) {
    unsigned r;
    asm("v_and_b32_e32 %0, %2, %1" : "=v"(r) : "v"(a), "n"(LIT));
    return r;
}
__device__ __forceinline__ unsigned add_v(unsigned a, unsigned b) {
    unsigned r;
    asm("v_add_u32_e32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ unsigned sub_v(unsigned a, unsigned b) {  // a - b
    unsigned r;
    asm("v_sub_u32_e32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ unsigned or_v(unsigned a, unsigned b) {
    unsigned r;
    asm("v_or_b32_e32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
template <int N>
__device__ __forceinline__ unsigned shr_k(unsigned a) {
    unsigned r;
    asm("v_lshrrev_b32_e32 %0, %2, %1" : "=v"(r) : "v"(a), "n"(N));
    return r;
}
__device__ __forceinline__ unsigned pk_max(unsigned a, unsigned b) {
    unsigned r;
    asm("v_pk_max_u16 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
// Three-way maximum of both halves in one instruction: v_pk_maximum3_f16 (new in gfx950).  Positive half-precision bit
// patterns below 0x7C00 order like unsigned integers and the instruction returns the winning operand's bits unchanged,
// denormal patterns included (tools/microbench/pk_max3.hip: all 6.0e9 checked triples exact, same issue cost as
// v_pk_max_u16) -- and every biased score of this kernel is in [0, 2 * KP_FILL16_MAX_GENE_LEN + 14] < 0x7C00.
__device__ __forceinline__ unsigned pk_max3(unsigned a, unsigned b, unsigned c) {
    unsigned r;
#ifdef KP_SW_NO_MAX3  // (A/B builds: KAPTIVE_AMD_EXTRA_FLAGS=-DKP_SW_NO_MAX3)
    asm("v_pk_max_u16 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    asm("v_pk_max_u16 %0, %1, %2" : "=v"(r) : "v"(r), "v"(c));
#else
    asm("v_pk_maximum3_f16 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
#endif
    return r;
}
__device__ __forceinline__ unsigned pk_shr(unsigned v, unsigned by) {  // each half of v shifted right by the same half of `by`
    unsigned r;
    asm("v_pk_lshrrev_b16 %0, %1, %2" : "=v"(r) : "v"(by), "v"(v));
    return r;
}
__device__ __forceinline__ unsigned pk_sign_mask(unsigned v, unsigned fifteen) {  // 0xFFFF in the halves whose bit 15 is set
    unsigned r;
    asm("v_pk_ashrrev_i16 %0, %1, %2" : "=v"(r) : "v"(fifteen), "v"(v));
    return r;
}
__device__ __forceinline__ unsigned bfi(unsigned mask, unsigned a, unsigned b) {  // (mask & a) | (~mask & b)
    unsigned r;
    asm("v_bfi_b32 %0, %1, %2, %3" : "=v"(r) : "s"(mask), "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ unsigned bfi_v(unsigned mask, unsigned a, unsigned b) {
    unsigned r;
    asm("v_bfi_b32 %0, %1, %2, %3" : "=v"(r) : "v"(mask), "v"(a), "v"(b));
    return r;
}
// a >= b for both halves: the answer in bits 15 and 31, the bits below are of no use to anyone
__device__ __forceinline__ unsigned ge_word(unsigned a, unsigned b) { return sub_v(or_k<GUARD>(a), b); }
// a < b for both halves, one instruction: the sign of the 16-bit difference (values are below 2^15, it cannot wrap)
__device__ __forceinline__ unsigned lt_word(unsigned a, unsigned b) {
    unsigned r;
    asm("v_pk_sub_i16 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
template <unsigned LIT>
__device__ __forceinline__ unsigned lit_minus(unsigned a) {  // LIT - a
    unsigned r;
    asm("v_sub_u32_e32 %0, %2, %1" : "=v"(r) : "v"(a), "n"(LIT));
    return r;
}

struct Cell {
    // H - (open + ext) + 12 (= H + 6), the same with the guard bits set, E - ext + 12, F - ext + 12 -- task X low, task Y high
    unsigned hmoe, hg, emex, fmex;
};

// Direction nibble of a cell, most significant bit first: [D][L][E opened][F opened].  D: the diagonal won (it wins
// ties, then E).  L: with D, "the diagonal predecessor holds H > 0" (0 = the path starts in this cell); without D,
// 1 = E, 0 = F.  e / f: the gap states arriving from the left / from above, eo / fo: comparison words "the gap was
// opened there" (open wins ties) -- computed by the caller, which also knows the band's edge lanes.  In the trace D and L
// are stored INVERTED (the comparisons that make them are one instruction cheaper that way; the traceback flips them).
// The three candidates are compared two below their value (d - 2, e - 2, f - 2: the last two are the cell's own
// pre-charged gap states, and d - 2 is the predecessor's H - (open + ext) plus the biased score), so H itself is never
// formed.  Four steps share 16 bits per task: a byte of [D, E opened] pairs above a byte of [L, F opened] pairs, the first
// step's pair lowest in each.
template <bool FIRST>
__device__ __forceinline__ void dp_cell(Cell &c, unsigned &acc, unsigned prof, unsigned tsh, unsigned e, unsigned eo, unsigned f,
                                        unsigned fo) {
    const unsigned s = and_k<7u * K1>(pk_shr(prof, tsh));        // score + 4
    // The D and L answers are produced INVERTED (one v_pk_sub_i16 each instead of an OR and a subtraction; the traceback reads
    // them that way): nzn = "H of the diagonal predecessor is 0", dwn = "the diagonal lost", ewn = "E lost too".
    const unsigned nzn = lit_minus<(CB - OE) * K1 | GUARD>(c.hmoe);
    const unsigned d = add_v(c.hmoe, s);                         // diagonal candidate - 2, biased by 12
    const unsigned en = add_k<0u - (unsigned)EX * K1>(e), fn = add_k<0u - (unsigned)EX * K1>(f);
    // H + 10: the caller floors the arriving E at 0 (e >= 12, so en >= 10 = "H is 0" on this scale) and that makes the
    // restart at 0 part of the three-way maximum.  A floored gap state changes no H anywhere (H >= 0 beats it as it beats
    // the true, negative one), and the direction bits of a path with a positive score only ever come from positive states.
    const unsigned m = pk_max3(d, en, fn);
    const unsigned dwn = lt_word(d, m), ewn = lt_word(en, m);    // (neither exceeds m)
    c.hmoe = add_k<0u - 4u * K1>(m);
    c.hg = or_k<GUARD>(c.hmoe);
    c.emex = en;
    c.fmex = fn;
    const unsigned lwn = bfi_v(dwn, ewn, nzn);
    // the bytes that hold the four answers of both tasks, side by side: [D_Y L_Y D_X L_X] and [EO_Y FO_Y EO_X FO_X], each
    // answer in bit 7 of its byte; the second set goes to bit 6, and a step's two bits per byte move down as steps follow
    const unsigned p1 = (unsigned)__builtin_amdgcn_perm(dwn, lwn, 0x07030501u), p2 = (unsigned)__builtin_amdgcn_perm(eo, fo, 0x07030501u);
    // bit 7 of every byte from p1, bit 6 from p2 (the bits below are leftovers); the accumulator takes exactly those two
    // bits of each byte and moves what it holds two down, so after four steps every bit of it has come through the mask
    const unsigned n = bfi(0x80808080u, p1, shr_k<1>(p2));
    acc = FIRST ? n : bfi(0xC0C0C0C0u, n, shr_k<2>(acc));
}

struct State {
    Cell A, B, C, D;
    unsigned qb, t0, t1, t2, t3;
    // running maximum of the lane's cells (H + 6), the PAIR of steps in which it last rose (its first step) and the eight
    // cells of that pair
    unsigned best, brow;
    unsigned s0[4], s1[4];
};

// Best cell of the lane so far, once per two steps (the comparison, the mask and the row select are shared by eight cells
// instead of four; the selects of the cells themselves cost the same): a strict rise keeps the first pair, and the saved
// cells tell the step of the pair and the column later.
__device__ __forceinline__ void track_pair(State &s, const unsigned (&h0)[4], const unsigned (&h1)[4], unsigned step_k, unsigned fifteen) {
    const unsigned risen = pk_max3(pk_max3(h0[0], h0[1], h0[2]), pk_max3(h0[3], h1[0], h1[1]), pk_max3(h1[2], h1[3], s.best));
    const unsigned rose = pk_sign_mask(lt_word(s.best, risen), fifteen);  // halves that rose (best <= risen everywhere)
    s.best = risen;
    s.brow = bfi_v(rose, step_k, s.brow);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        s.s0[k] = bfi_v(rose, h0[k], s.s0[k]);
        s.s1[k] = bfi_v(rose, h1[k], s.s1[k]);
    }
}

template <int P, bool FIRST>
__device__ __forceinline__ void dp_step(State &s, unsigned (&acc)[4], unsigned (&hs)[4], unsigned prof_mine, unsigned t_next,
                                        unsigned first_all, unsigned last_all, unsigned twelve) {
    constexpr bool ROW = P <= 16;
    s.qb = prof_mine;  // profile of this lane's row m - l, read from LDS by the lane itself (the LDS pipe is idle otherwise;
                       // streaming the rows up the lanes cost a DPP move and a select per step on the busy vector pipe)

    // The gap state a cell hands to its right neighbour is max(H - open - ext, E - ext, 0): the floor at 0 (twelve on the
    // biased scale) is what lets dp_cell leave out the restart (see there); it rides in the three-way maximum for free.
    // A: the left neighbour is lane l-1's D of the previous step -- that lane forms what it hands over (it has the guard
    // word of its D already) and the two results cross; left of the band's first diagonal H = 0, so the band's first lane
    // takes E = 0 (any E <= 0 does), "opened" or not (nobody ever asks: that E loses to the restart and no path with a
    // positive score stands in state E on the band's first diagonal)
    const unsigned eA = bfi_v(first_all, twelve, from_lower<ROW>(pk_max3(s.D.hmoe, s.D.emex, twelve)));
    const unsigned eoA = from_lower<ROW>(sub_v(s.D.hg, s.D.emex));  // (only bits 15 and 31 are looked at)
    dp_cell<FIRST>(s.A, acc[0], s.qb, s.t0, eA, eoA, pk_max(s.B.hmoe, s.B.fmex), sub_v(s.B.hg, s.B.fmex));
    dp_cell<FIRST>(s.B, acc[1], s.qb, s.t1, pk_max3(s.A.hmoe, s.A.emex, twelve), sub_v(s.A.hg, s.A.emex),
                   pk_max(s.C.hmoe, s.C.fmex), sub_v(s.C.hg, s.C.fmex));
    dp_cell<FIRST>(s.C, acc[2], s.qb, s.t2, pk_max3(s.B.hmoe, s.B.emex, twelve), sub_v(s.B.hg, s.B.emex),
                   pk_max(s.D.hmoe, s.D.fmex), sub_v(s.D.hg, s.D.fmex));
    // D: the upper neighbour is lane l+1's A of this step; above the band's last diagonal H = 0 as well (F = 0 there
    // stands for any F <= 0, as above)
    const unsigned fD = bfi_v(last_all, twelve, from_upper<ROW>(pk_max(s.A.hmoe, s.A.fmex)));
    const unsigned foD = from_upper<ROW>(sub_v(s.A.hg, s.A.fmex));
    dp_cell<FIRST>(s.D, acc[3], s.qb, s.t3, pk_max3(s.C.hmoe, s.C.emex, twelve), sub_v(s.C.hg, s.C.emex), fD, foD);

    hs[0] = s.A.hmoe; hs[1] = s.B.hmoe; hs[2] = s.C.hmoe; hs[3] = s.D.hmoe;
    s.t0 = s.t1; s.t1 = s.t2; s.t2 = s.t3;
    s.t3 = t_next;  // the code of x = m + 3l + 4, this lane's cell D at the next step (also read by the lane itself)
}

constexpr int PROF_WORDS = 16 * (CH + 4);                // LDS of one block: profile pairs of 64/P groups x (P + CH) rows (P >= 4)
constexpr int TCODE_HALVES = 16 * (CH + 3 * 4 + 4 + 4);  // ... their target shift amounts, a byte per task (largest for P = 4)
constexpr int TWORD_WORDS = 2 * 128;                     // ... and the packed words those are cut from
constexpr int SPREAD_WORDS = 256;                        // four 2-bit codes -> four shift amounts (3 * code), a byte each

// all tasks of one band class, seen from block `block` of `n_blocks` that work on the class
template <int P>
__device__ __forceinline__ void sw_class(const KpBatchView &b, const KpGenes &genes, const KpTask *__restrict__ tasks,
                                         uint32_t n_tasks, const uint32_t *__restrict__ order,
                                         KpSwEnd *__restrict__ ends, uint4 *__restrict__ trace,
                                         unsigned long long *__restrict__ trace_top, uint64_t trace_cap, uint32_t block,
                                         uint32_t n_blocks, unsigned int *__restrict__ next_quad, uint32_t *s_prof_raw,
                                         uint16_t *s_t_raw, uint32_t *s_tw_raw, const uint32_t *s_spread) {
    constexpr int G = 64 / P;
    constexpr int TW = CH + 3 * P + 4;  // staged target codes per chunk: window x in [m0, m0 + CH + 3P]
    // the codes a step pulls in start at x = step + 3P + 1: the row is shifted by PAD so that every fourth step's lies
    // on an 8-byte boundary (one ds_read_b64 feeds four steps)
    constexpr int PAD = (4 - (3 * P + 1) % 4) % 4;
    constexpr int TROW = (TW + PAD + 3) & ~3;  // byte pair y of the row = position lo + m0 - PAD + y
    // profiles: entry i of a group's row is query row m0 - P + i of the chunk being computed: the P rows before the chunk
    // stay, because lane l works on row m - l
    constexpr int PROW = CH + P;
    static_assert(G * PROW <= PROF_WORDS && G * TROW <= TCODE_HALVES, "LDS carve-up");
    uint32_t(*s_prof)[PROW] = reinterpret_cast<uint32_t(*)[PROW]>(s_prof_raw);
    uint16_t(*s_t)[TROW] = reinterpret_cast<uint16_t(*)[TROW]>(s_t_raw);  // byte 0: task X, byte 1: task Y

    const int lane = threadIdx.x;
    const int g = lane / P, l = lane % P;
    // all ones in the band's first / last lane: the selects of the band edges are bitwise (a ternary around the asm
    // statements becomes a branch)
    unsigned first_all = l == 0 ? ~0u : 0u, last_all = l == P - 1 ? ~0u : 0u;
    // (VGPR copies of constants that VOP3 / VOP3P instructions cannot take as literals)
    unsigned twelve = CB * K1, fifteen = 15u * K1;  // twelve: a gap state of 0 on the biased scale
    asm volatile("" : "+v"(twelve), "+v"(fifteen), "+v"(first_all), "+v"(last_all));

    // The narrow class hands its quads out by a counter (longest tasks first) to a fixed number of single-wave blocks that
    // keep taking until it is exhausted: 32 per CU, twice what is resident, so the dispatcher always has a block to put
    // into a slot another kernel gives back, and the tail is as even as the last quads are short.  (One block per quad --
    // short-lived blocks, slots turning over every millisecond for other streams' kernels -- was measured too: 10.7 ms
    // against 10.3 ms for this launch and 36.1 k against 37.5 k assemblies/s.)  The wide classes have counters of their own (kp_sw_kernel).
    auto take = [&]() -> uint32_t {
        uint32_t q = 0;
        if (lane == 0) q = atomicAdd(next_quad, 1u);
        return (uint32_t)__builtin_amdgcn_readfirstlane((int)q);
    };
    for (uint32_t quad = next_quad ? take() : block; (uint64_t)quad * (2 * G) < n_tasks;
         quad = next_quad ? take() : quad + n_blocks) {
        // the pair of this group: two neighbours in the length-ordered list (kp_chain.hip)
        bool have[2];
        uint32_t ti[2];
        KpTask tk[2];
        int qlen[2], lo[2], q0[2], n_runs[2], steps[2], n_chunks[2];  // lo: column of (row q0, band index 0); q0: first row filled
        int32_t cstart[2], cend[2];
        // (32-bit offsets from the batch's and the genes' arrays rather than pointers: registers)
        uint32_t asm_n_words[2], q_off[2], w_off[2], run_off[2];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const uint32_t slot = (quad * G + g) * 2 + h;
            have[h] = slot < n_tasks;
            ti[h] = have[h] ? order[slot] : 0u;
            tk[h].asm_id = 0; tk[h].gs = 0; tk[h].contig = 0; tk[h].lo = 0;
            if (have[h]) {  // the first four fields are all the fill needs (one 16-byte load; the whole record is eight words)
                const int4 head = *reinterpret_cast<const int4 *>(&tasks[ti[h]]);
                tk[h].asm_id = head.x; tk[h].gs = head.y; tk[h].contig = head.z; tk[h].lo = head.w;
            }
            const int gene = tk[h].gs >> 1;
            qlen[h] = have[h] ? genes.len[gene] : 0;
            q_off[h] = (uint32_t)genes.word_off[(tk[h].gs & 1) ? genes.n_genes + gene : gene];
            w_off[h] = (uint32_t)b.asm_word_off[tk[h].asm_id];
            asm_n_words[h] = (uint32_t)(b.asm_word_off[tk[h].asm_id + 1] - b.asm_word_off[tk[h].asm_id]);
            const int c_abs = b.asm_first_ctg[tk[h].asm_id] + tk[h].contig;
            cstart[h] = b.ctg_start[c_abs]; cend[h] = cstart[h] + b.ctg_len[c_abs];
            const int r0 = b.asm_first_nrun[tk[h].asm_id];
            n_runs[h] = b.asm_first_nrun[tk[h].asm_id + 1] - r0;
            run_off[h] = 2u * (uint32_t)r0;
            // only the rows whose band cells can lie inside the contig are filled (kp_task_rows): step m of lane l is row
            // q0 + m - l, and everything below works in that shifted frame
            int r_hi;
            kp_task_rows(tk[h].lo, 4 * P, cstart[h], cend[h], qlen[h], &q0[h], &r_hi);
            lo[h] = tk[h].lo + q0[h];
            steps[h] = have[h] ? (r_hi - q0[h]) + P - 1 : 0;  // steps the task needs
            // 8-step trace pieces per lane, in whole groups of four
            n_chunks[h] = (((steps[h] + 7) >> 3) + 3) & ~3;  // (a multiple of four whatever TG: task blocks start on 128-byte lines)
        }
        int max_steps = max(steps[0], steps[1]);
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) max_steps = max(max_steps, __shfl_xor(max_steps, o));
        // the pair's trace blocks, X's then Y's: P lane streams of n_chunks 16-byte pieces each
        const unsigned long long want = (unsigned long long)P * (unsigned)((have[0] ? n_chunks[0] : 0) + (have[1] ? n_chunks[1] : 0));
        unsigned long long toff = 0;
        if (have[0] && l == 0) toff = atomicAdd(trace_top, want);
        toff = ((unsigned long long)__shfl((unsigned)(toff >> 32), g * P) << 32) | __shfl((unsigned)toff, g * P);
        const bool fits = have[0] && toff + want <= trace_cap;  // else: counted, host reruns
        // piece j of lane l at [j / TG][l][j % TG]: TG consecutive pieces of a lane are contiguous, so the traceback -- it
        // follows one lane's pieces backwards -- fetches them in one go.  With TG = 2 a 128-byte line (four lanes) is
        // complete after 16 steps; TG = 4 halves the traceback's fetches again (1.75 ms against 2.75 ms with a fetch per
        // piece) but keeps lines open for 32 steps, and the fill kernel pays more for the partial write-backs than the
        // traceback gains (11.9 ms against 11.0 ms); with a whole stream per lane a line stayed open for 64 steps, more open
        // lines than the L2 holds, and HBM saw three times the bytes (WRITE_SIZE, profiles/).
        const unsigned long long toff_y = toff + (unsigned long long)P * (unsigned)n_chunks[0];
        uint4 *trace_x = trace + toff + TG * l, *trace_y = trace + toff_y + TG * l;

        State st;
        st.A.hmoe = GAP_EDGE; st.A.hg = GAP_EDGE | GUARD; st.A.emex = GAP_NONE; st.A.fmex = GAP_NONE;
        st.B = st.A; st.C = st.A; st.D = st.A;
        st.qb = PROF_OUT; st.t0 = st.t1 = st.t2 = st.t3 = T_OUT * K1;
        st.best = GAP_EDGE; st.brow = 0;  // H = 0
#pragma unroll
        for (int k = 0; k < 4; ++k) st.s0[k] = st.s1[k] = GAP_EDGE;
        unsigned acc[4] = {0, 0, 0, 0}, held[4] = {0, 0, 0, 0};
        // an N in the gene or in the target window: the traceback then compares bases itself
        bool saw_n[2] = {have[0] && genes.has_n[tk[0].gs >> 1] != 0, have[1] && genes.has_n[tk[1].gs >> 1] != 0};

        const int steps8 = (max_steps + 7) & ~7;
        // Staging of a chunk (profiles of its query rows, codes of its target window) works from packed words that were
        // requested one chunk earlier: NQ gene words (8 rows each) and NT assembly words (16 bases each) per lane and task.
        constexpr int NQ = (CH / 8 + P - 1) / P, NTW = (TROW + 15) / 16 + 1, NT = (NTW + P - 1) / P;
        static_assert(2 * G * NTW <= TWORD_WORDS, "LDS carve-up");
        uint32_t(*s_tw)[G][NTW] = reinterpret_cast<uint32_t(*)[G][NTW]>(s_tw_raw);
        uint4 qreg[2][NQ];  // eight row profiles each (KpGenes::prof: nothing to compute, rows past the gene's end are 0 there)
        uint32_t treg[2][NT];
        auto request = [&](int m0) {  // global loads only; nothing waits for them here
#pragma unroll
            for (int h = 0; h < 2; ++h) {
#pragma unroll
                for (int i = 0; i < NQ; ++i) {
                    const int w = l + i * P, r = q0[h] + m0 + 8 * w;
                    qreg[h][i] = (have[h] && w < CH / 8 && r < qlen[h]) ? genes.prof[q_off[h] + (uint32_t)(r >> 3)] : make_uint4(0u, 0u, 0u, 0u);
                }
                const int w0 = (lo[h] + m0 - PAD) >> 4;
#pragma unroll
                for (int i = 0; i < NT; ++i) {
                    const int wi = w0 + l + i * P;
                    treg[h][i] = (have[h] && l + i * P < NTW && wi >= 0 && (uint32_t)wi < asm_n_words[h]) ? b.words[w_off[h] + (uint32_t)wi] : 0u;
                }
            }
        };
        request(0);
        for (int m0 = 0; m0 < steps8; m0 += CH) {
            // ---- stage this chunk: profiles of query rows [m0-P, m0+CH), target codes of window x in [m0, m0+CH+3P] ----
            __syncthreads();
            {   // the last P rows of the previous chunk move to the front (a block is one wave: every lane has read its
                // entry before any lane's stores below are issued); before the first chunk they are rows outside the gene
                const uint32_t keep = m0 == 0 ? PROF_OUT : s_prof[g][CH + l];
                s_prof[g][l] = keep;
            }
#pragma unroll
            for (int i = 0; i < NQ; ++i) {
                const int w = l + i * P;
                if (w < CH / 8) {  // rows 8w .. 8w + 7 of the chunk, task X in the low halves and task Y in the high ones
                    const uint4 x = qreg[0][i], y = qreg[1][i];
                    uint4 *dst = reinterpret_cast<uint4 *>(&s_prof[g][P + 8 * w]);
                    dst[0] = make_uint4(__builtin_amdgcn_perm(y.x, x.x, 0x05040100u), __builtin_amdgcn_perm(y.x, x.x, 0x07060302u),
                                        __builtin_amdgcn_perm(y.y, x.y, 0x05040100u), __builtin_amdgcn_perm(y.y, x.y, 0x07060302u));
                    dst[1] = make_uint4(__builtin_amdgcn_perm(y.z, x.z, 0x05040100u), __builtin_amdgcn_perm(y.z, x.z, 0x07060302u),
                                        __builtin_amdgcn_perm(y.w, x.w, 0x05040100u), __builtin_amdgcn_perm(y.w, x.w, 0x07060302u));
                }
            }
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int i = 0; i < NT; ++i)
                    if (l + i * P < NTW) s_tw[h][g][l + i * P] = treg[h][i];
            __syncthreads();
            {
                // shift amounts (3 * code) of the window, four positions at a time: byte pair y of the padded row =
                // position pw + y with pw = lo + m0 - PAD (the PAD entries before the window are never read; they get real
                // codes like the rest).  Whole groups inside the contig of an assembly without N runs -- nearly all -- are
                // cut out of the packed words with one funnel shift and spread to bytes by a table; the rest go base by base.
                for (int y = 4 * l; y < TROW; y += 4 * P) {
                    uint32_t four[2];
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        const int pw = lo[h] + m0 - PAD, w0 = pw >> 4;
                        const int t0 = pw + y;
                        if (have[h] && n_runs[h] == 0 && t0 >= cstart[h] && t0 + 3 < cend[h]) {
                            const int wi = (t0 >> 4) - w0;
                            const uint32_t lo_w = s_tw[h][g][wi], hi_w = wi + 1 < NTW ? s_tw[h][g][wi + 1] : 0u;
                            four[h] = s_spread[__builtin_amdgcn_alignbit(hi_w, lo_w, 2 * (t0 & 15)) & 255u];
                        } else {
                            four[h] = 0;
                            for (int i = 0; i < 4; ++i) {
                                const int t = t0 + i;
                                unsigned code = 5u;
                                if (have[h] && t >= cstart[h] && t < cend[h]) {
                                    code = (s_tw[h][g][(t >> 4) - w0] >> (2 * (t & 15))) & 3u;
                                    if (n_runs[h] > 0) {  // rare: assemblies with scaffold gaps
                                        int a = 0, z = n_runs[h];
                                        while (a < z) {
                                            const int mid = (a + z) >> 1;
                                            if (b.n_runs[run_off[h] + 2 * mid + 1] <= t) a = mid + 1; else z = mid;
                                        }
                                        if (a < n_runs[h] && b.n_runs[run_off[h] + 2 * a] <= t) code = 4u;
                                    }
                                }
                                four[h] |= (code < 5u ? 3u * code : T_OUT) << (8 * i);
                                saw_n[h] |= code == 4u;
                            }
                        }
                    }
                    // bytes X0 Y0 X1 Y1 | X2 Y2 X3 Y3 (v_perm_b32 selects: bytes 0-3 = second operand, 4-7 = first)
                    const uint32_t lo2 = __builtin_amdgcn_perm(four[1], four[0], 0x05010400u);
                    const uint32_t hi2 = __builtin_amdgcn_perm(four[1], four[0], 0x07030602u);
                    *reinterpret_cast<uint2 *>(&s_t[g][y]) = make_uint2(lo2, hi2);
                }
            }
            __syncthreads();
            if (m0 + CH < steps8) request(m0 + CH);  // the next chunk's words arrive while this chunk is computed
            if (m0 == 0) {  // initial window: cell k of lane l sits on x = 3l + k
                const unsigned a0 = s_t[g][PAD + 3 * l], a1 = s_t[g][PAD + 3 * l + 1], a2 = s_t[g][PAD + 3 * l + 2],
                               a3 = s_t[g][PAD + 3 * l + 3];
                st.t0 = (a0 & 255u) | ((a0 >> 8) << 16); st.t1 = (a1 & 255u) | ((a1 >> 8) << 16);
                st.t2 = (a2 & 255u) | ((a2 >> 8) << 16); st.t3 = (a3 & 255u) | ((a3 >> 8) << 16);
            }
            const int m_end = min(m0 + CH, steps8);
            for (int m = m0; m < m_end; m += 8) {
#pragma unroll
                for (int half = 0; half < 2; ++half) {
                    const int mm = m + 4 * half;
                    // this lane's rows mm - l .. mm - l + 3 and the codes of x = mm + 3l + 4 .. + 3, straight from LDS
                    const uint32_t *pp = &s_prof[g][mm - m0 + P - l];
                    const uint16_t *tp = &s_t[g][mm - m0 + 3 * l + 4 + PAD];
                    const unsigned p0 = pp[0], p1 = pp[1], p2 = pp[2], p3 = pp[3];
                    const unsigned c0 = tp[0], c1 = tp[1], c2 = tp[2], c3 = tp[3];
                    // byte pairs -> (X | Y << 16): selector bytes 0x0c are zeros
                    const unsigned ta = __builtin_amdgcn_perm(0u, c0, 0x0c010c00u), tb = __builtin_amdgcn_perm(0u, c1, 0x0c010c00u);
                    const unsigned tcw = __builtin_amdgcn_perm(0u, c2, 0x0c010c00u), td = __builtin_amdgcn_perm(0u, c3, 0x0c010c00u);
                    unsigned h0[4], h1[4];
                    dp_step<P, true>(st, acc, h0, p0, ta, first_all, last_all, twelve);
                    dp_step<P, false>(st, acc, h1, p1, tb, first_all, last_all, twelve);
                    track_pair(st, h0, h1, (unsigned)mm * K1, fifteen);
                    dp_step<P, false>(st, acc, h0, p2, tcw, first_all, last_all, twelve);
                    dp_step<P, false>(st, acc, h1, p3, td, first_all, last_all, twelve);
                    track_pair(st, h0, h1, (unsigned)(mm + 2) * K1, fifteen);
                    if (half == 0) {
#pragma unroll
                        for (int c = 0; c < 4; ++c) held[c] = acc[c];
                    }
                }
                // eight steps of both tasks per cell: X = low halves, Y = high halves (steps 0-3 low, 4-7 high)
                const int j = m >> 3;
                if (fits && j < n_chunks[0])
                    trace_x[(size_t)(j / TG) * (TG * P) + (j % TG)] = make_uint4(__builtin_amdgcn_perm(acc[0], held[0], 0x05040100u), __builtin_amdgcn_perm(acc[1], held[1], 0x05040100u),
                                                        __builtin_amdgcn_perm(acc[2], held[2], 0x05040100u), __builtin_amdgcn_perm(acc[3], held[3], 0x05040100u));
                if (fits && have[1] && j < n_chunks[1])
                    trace_y[(size_t)(j / TG) * (TG * P) + (j % TG)] = make_uint4(__builtin_amdgcn_perm(acc[0], held[0], 0x07060302u), __builtin_amdgcn_perm(acc[1], held[1], 0x07060302u),
                                                        __builtin_amdgcn_perm(acc[2], held[2], 0x07060302u), __builtin_amdgcn_perm(acc[3], held[3], 0x07060302u));
            }
        }

        // ---- best cell of each task: max score, then first row, then first column -------------------------------------
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const unsigned v = (st.best >> (16 * h)) & 0xFFFFu;  // H + 6
            // the first step of the saved pair that holds the maximum, then the first of its cells
            int mstar = (int)((st.brow >> (16 * h)) & 0xFFFFu), cell = 3;
            bool in0 = false;
#pragma unroll
            for (int k = 3; k >= 0; --k)
                if (((st.s0[k] >> (16 * h)) & 0xFFFFu) == v) { cell = k; in0 = true; }
            if (!in0) {
                ++mstar;
#pragma unroll
                for (int k = 2; k >= 0; --k)
                    if (((st.s1[k] >> (16 * h)) & 0xFFFFu) == v) cell = k;
            }
            int eb = 4 * l + cell;
            unsigned key = v > ZB ? ((v - ZB) << 15) | (unsigned)(32767 - (q0[h] + mstar - l)) : 0u;
            if (key == 0u) eb = 4 * l;
            bool sn = saw_n[h];
#pragma unroll
            for (int o = 1; o < P; o <<= 1) {
                const unsigned key2 = (unsigned)__shfl_xor((int)key, o);
                const int eb2 = __shfl_xor(eb, o);
                if (key2 > key || (key2 == key && eb2 < eb)) { key = key2; eb = eb2; }
                sn |= __shfl_xor((int)sn, o) != 0;
            }
            if (have[h] && l == 0) {
                KpSwEnd out;
                out.score = fits ? (int)(key >> 15) : 0;
                out.er = 32767 - (int)(key & 32767u);
                out.eb = eb | (sn ? KP_SWEND_HAS_N : 0);
                out.trace_off = (uint32_t)(h ? toff_y : toff);
                ends[ti[h]] = out;
            }
        }
    }
}

// One launch for all four band classes.  On the headline workload the wide classes hold few tasks but each of their waves
// runs a long step chain (a launch of its own costs ~1 ms of latency at the end of the pass), so they get the first blocks of
// the grid and run underneath the 16-diagonal class that fills the chip.
#ifndef KP_SW_WIDE_BLOCKS
#define KP_SW_WIDE_BLOCKS 512
#endif
constexpr uint32_t WIDE_BLOCKS = KP_SW_WIDE_BLOCKS;  // blocks per wide class at the front of the grid
#ifndef KP_SW_HELP_BLOCKS
#define KP_SW_HELP_BLOCKS 3072
#endif
constexpr uint32_t HELP_BLOCKS = KP_SW_HELP_BLOCKS;  // ... and behind the narrow class's blocks (as many as the chip holds waves of this kernel)
#ifndef KP_SW_NARROW_BLOCKS_PER_CU
#define KP_SW_NARROW_BLOCKS_PER_CU 32
#endif
constexpr uint32_t NARROW_BLOCKS_PER_CU = KP_SW_NARROW_BLOCKS_PER_CU;  // blocks of the 16-diagonal class (they take quads off a counter)

#ifndef KP_SW_WAVES
#define KP_SW_WAVES 3  // waves per SIMD the register budget is set for: 145 VGPRs, nothing spilled to scratch (at 4 -- 128 VGPRs --
                       // the per-quad set-up spilled 17 VGPRs; round 4, same box: 46.8-47.0 k assemblies/s at 3 against 46.2-46.4 k at 4,
                       // fill alone 7.36 ms either way: the kernel is bound by vector issue, not by latency)
#endif
#ifndef KP_SW_LDS_PAD
#define KP_SW_LDS_PAD 0
#endif
__global__ __launch_bounds__(64, KP_SW_WAVES) void kp_sw_kernel(KpBatchView b, KpGenes genes, const KpTask *__restrict__ tasks,
                                                   const uint32_t *__restrict__ task_count, uint32_t task_cap,
                                                   const uint32_t *__restrict__ order, KpSwEnd *__restrict__ ends,
                                                   uint4 *__restrict__ trace, unsigned long long *__restrict__ trace_top,
                                                   uint64_t trace_cap) {
    __shared__ __attribute__((aligned(16))) uint32_t s_prof[PROF_WORDS];
    __shared__ __attribute__((aligned(16))) uint16_t s_t[TCODE_HALVES];
    __shared__ uint32_t s_tw[TWORD_WORDS];
    __shared__ uint32_t s_spread[SPREAD_WORDS];
#if KP_SW_LDS_PAD
    // LDS that is never used: it limits the blocks of this kernel a CU holds (160 KB / (9 KB + pad)), so that a SIMD per CU keeps
    // room for a wave of the kernels that run beside it -- the persistent blocks of this one give their slots up only at its end
    __shared__ uint32_t s_pad[KP_SW_LDS_PAD / 4];
    if (task_cap == 0xFFFFFFFFu) { s_pad[threadIdx.x] = blockIdx.x; __syncthreads(); trace_top[3] = s_pad[63 - threadIdx.x]; }  // (never)
#endif
    for (uint32_t v = threadIdx.x; v < SPREAD_WORDS; v += 64)
        s_spread[v] = 3u * ((v & 3u) | ((v & 0xCu) << 6) | ((v & 0x30u) << 12) | ((v & 0xC0u) << 18));
    // (every use comes after the first chunk's barriers)
    // class c (0..3 = 16/32/64/128 diagonals): tasks, order and results at c * task_cap, count at task_count[c].  Every class
    // hands its quads out by a counter of its own (trace_top[1..2]: four 32-bit words, zeroed with trace_top before the
    // launch).  The grid: WIDE_BLOCKS blocks per wide class first (they run underneath the narrow class), the narrow class's
    // blocks, then HELP_BLOCKS more per wide class, widest first.  Those are dispatched as the narrow blocks retire and
    // leave at once when their class has nothing left; on a workload rich in wide bands (diverged relatives with indels:
    // `bench.py --background paralog`) they are what finishes the wide classes on the whole chip instead of on the 3 x 512
    // wave slots of the first blocks -- that launch ran at 0.44 of its issue roof.  (A block that goes on to the next class
    // itself, a loop around the four instantiations, costs 23 more VGPRs and 124 bytes of scratch per lane.)
    const uint32_t blk = blockIdx.x, narrow_end = 3 * WIDE_BLOCKS + 256u * NARROW_BLOCKS_PER_CU;
    const int c = blk < 3 * WIDE_BLOCKS ? 3 - (int)(blk / WIDE_BLOCKS) : blk < narrow_end ? 0 : 3 - (int)((blk - narrow_end) / (HELP_BLOCKS ? HELP_BLOCKS : 1u));
    const size_t off = (size_t)c * task_cap;
    uint32_t n = task_count[c];
    if (n > task_cap) n = task_cap;
    unsigned int *next_quad = reinterpret_cast<unsigned int *>(trace_top + 1) + c;
    if (c == 3) sw_class<32>(b, genes, tasks + off, n, order + off, ends + off, trace, trace_top, trace_cap, 0, 1, next_quad, s_prof, s_t, s_tw, s_spread);
    else if (c == 2) sw_class<16>(b, genes, tasks + off, n, order + off, ends + off, trace, trace_top, trace_cap, 0, 1, next_quad, s_prof, s_t, s_tw, s_spread);
    else if (c == 1) sw_class<8>(b, genes, tasks + off, n, order + off, ends + off, trace, trace_top, trace_cap, 0, 1, next_quad, s_prof, s_t, s_tw, s_spread);
    else sw_class<4>(b, genes, tasks + off, n, order + off, ends + off, trace, trace_top, trace_cap, 0, 1, next_quad, s_prof, s_t, s_tw, s_spread);
}


// ---- fill with 32-bit scores: the tasks of genes longer than KP_FILL16_MAX_GENE_LEN ------------------------------------------
// The reference compiles any CDS (src/kaptive/db/core.py:402-404, 478); the packed kernel above holds scores in 16 bits.
// The rare longer gene's tasks are left out of its order (kp_chain.hip) and filled here instead: the same band mapping (P
// lanes per task, lane l holds diagonals 4l .. 4l + 3 and works on row m - l at step m), the recurrence of kp_spec.h
// written out in plain 32-bit arithmetic, ONE task per lane group, sequences read straight from the packed words --
// several times slower per cell, on a handful of tasks.  It leaves what kp_sw_kernel leaves: the best cell (KpSwEnd) and the
// direction words in the same layout (a cell's word: eight steps, per four steps a byte of [not D, E opened] pairs above a
// byte of [not L, F opened] pairs), so the traceback kernel does not know which kernel filled a task.
constexpr int LONG_NEG = -(1 << 29);

template <int P>
__device__ __forceinline__ void sw_long_class(const KpBatchView &b, const KpGenes &genes, const KpTask *__restrict__ tasks,
                                              const uint32_t *__restrict__ order, uint32_t n_tasks, KpSwEnd *__restrict__ ends,
                                              uint4 *__restrict__ trace, unsigned long long *__restrict__ trace_top, uint64_t trace_cap,
                                              uint32_t block, uint32_t n_blocks) {
    constexpr int G = 64 / P;
    const int lane = threadIdx.x, g = lane / P, l = lane % P;
    for (uint32_t quad = block; (uint64_t)quad * G < n_tasks; quad += n_blocks) {
        const uint32_t slot = quad * G + g;
        const bool have = slot < n_tasks;
        const uint32_t ti = have ? order[slot] : 0u;
        KpTask tk;
        tk.asm_id = 0; tk.gs = 0; tk.contig = 0; tk.lo = 0;
        if (have) tk = tasks[ti];
        const int gene = tk.gs >> 1;
        const int qlen = have ? genes.len[gene] : 0;
        const uint32_t *qnib = genes.nib + genes.word_off[(tk.gs & 1) ? genes.n_genes + gene : gene];
        const uint32_t *asm_words = b.words + b.asm_word_off[tk.asm_id];
        const int c_abs = b.asm_first_ctg[tk.asm_id] + tk.contig;
        const int cstart = b.ctg_start[c_abs], cend = cstart + b.ctg_len[c_abs];
        const int r0n = b.asm_first_nrun[tk.asm_id], n_runs = b.asm_first_nrun[tk.asm_id + 1] - r0n;
        const int32_t *runs = b.n_runs + 2 * (size_t)r0n;
        int q0, r_hi;
        kp_task_rows(tk.lo, 4 * P, cstart, cend, qlen, &q0, &r_hi);
        const int steps = have ? (r_hi - q0) + P - 1 : 0;
        const int n_chunks = (((steps + 7) >> 3) + 3) & ~3;
        int max_steps = steps;
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) max_steps = max(max_steps, __shfl_xor(max_steps, o));
        const unsigned long long want = (unsigned long long)P * (unsigned)(have ? n_chunks : 0);
        unsigned long long toff = 0;
        if (have && l == 0) toff = atomicAdd(trace_top, want);
        toff = ((unsigned long long)__shfl((unsigned)(toff >> 32), g * P) << 32) | __shfl((unsigned)toff, g * P);
        const bool fits = have && toff + want <= trace_cap;
        uint4 *trace_x = trace + toff + TG * l;

        auto target_code = [&](int t) -> int {  // 0..3, 4 = N, 5 = outside the contig
            if (t < cstart || t >= cend) return 5;
            int code = (int)((asm_words[t >> 4] >> (2 * (t & 15))) & 3u);
            if (n_runs > 0) {
                int a = 0, z = n_runs;
                while (a < z) {
                    const int mid = (a + z) >> 1;
                    if (runs[2 * mid + 1] <= t) a = mid + 1; else z = mid;
                }
                if (a < n_runs && runs[2 * a] <= t) code = 4;
            }
            return code;
        };
        int H[4] = {0, 0, 0, 0}, E[4], F[4], tc[4];
        bool saw_n = have && genes.has_n[gene] != 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) { E[k] = F[k] = LONG_NEG; tc[k] = 5; }
        int best = 0, best_r = 0, best_b = 4 * l;
        uint32_t word[4] = {0, 0, 0, 0};
        const int steps8 = (max_steps + 7) & ~7;
        for (int m = 0; m < steps8; ++m) {
            const int r = q0 + m - l;  // this lane's row
            const bool row_ok = have && r >= q0 && r < r_hi;
            const int qc = row_ok ? (int)nibble(qnib[r >> 3], r & 7) : 4;
            // columns of the lane's cells: t0 + k, t0 = lo + r + 4l
            const int t0 = tk.lo + r + 4 * l;
#pragma unroll
            for (int k = 0; k < 4; ++k) tc[k] = row_ok ? target_code(t0 + k) : 5;
            // left neighbour of A: lane l - 1's D as the previous step left it; upper neighbour of D: lane l + 1's A of this step
            int hl = __shfl_up(H[3], 1), el = __shfl_up(E[3], 1);
            if (l == 0) { hl = 0; el = LONG_NEG; }
            const int oldH[4] = {H[0], H[1], H[2], H[3]}, oldF[4] = {F[0], F[1], F[2], F[3]};
            const int sh = 2 * (m & 3) + 16 * ((m >> 2) & 1);
            int hu_d = 0, fu_d = LONG_NEG;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                if (k == 3) {  // (every lane has computed its A by now)
                    hu_d = __shfl_down(H[0], 1); fu_d = __shfl_down(F[0], 1);
                    if (l == P - 1) { hu_d = 0; fu_d = LONG_NEG; }
                }
                const int hleft = k == 0 ? hl : H[k - 1], eleft = k == 0 ? el : E[k - 1];
                const int hup = k == 3 ? hu_d : oldH[k + 1], fup = k == 3 ? fu_d : oldF[k + 1];
                const int hd = oldH[k];
                const int code = tc[k];
                const bool inside = row_ok && code < 5;
                saw_n |= inside && code == 4;
                const int e_open = hleft - OE, e_ext = eleft - EX, f_open = hup - OE, f_ext = fup - EX;
                const int e = e_open >= e_ext ? e_open : e_ext, f = f_open >= f_ext ? f_open : f_ext;
                const int s = (qc > 3 || code > 3) ? KP_SC_N : (qc == code ? KP_SC_MATCH : KP_SC_MISMATCH);
                int bv = hd + s, tb = 0;
                if (e > bv) { bv = e; tb = 1; }
                if (f > bv) { bv = f; tb = 2; }
                const uint32_t not_d = tb != 0, not_l = tb == 0 ? (hd == 0) : (tb == 2);
                const uint32_t eo = e_open >= e_ext, fo = f_open >= f_ext;
                word[k] |= ((not_d << 9) | (eo << 8) | (not_l << 1) | fo) << sh;
                if (inside) {
                    E[k] = e; F[k] = f;
                    H[k] = bv > 0 ? bv : 0;
                    if (bv > best) { best = bv; best_r = r; best_b = 4 * l + k; }
                } else {
                    H[k] = 0; E[k] = LONG_NEG; F[k] = LONG_NEG;
                }
            }
            if ((m & 7) == 7) {
                const int j = m >> 3;
                if (fits && j < n_chunks) trace_x[(size_t)(j / TG) * (TG * P) + (j % TG)] = make_uint4(word[0], word[1], word[2], word[3]);
                word[0] = word[1] = word[2] = word[3] = 0;
            }
        }
        // best cell of the task: the largest score, then the first row, then the first column
        bool sn = saw_n;
#pragma unroll
        for (int o = 1; o < P; o <<= 1) {
            const int s2 = __shfl_xor(best, o), r2 = __shfl_xor(best_r, o), b2 = __shfl_xor(best_b, o);
            if (s2 > best || (s2 == best && (r2 < best_r || (r2 == best_r && b2 < best_b)))) { best = s2; best_r = r2; best_b = b2; }
            sn |= __shfl_xor((int)sn, o) != 0;
        }
        if (have && l == 0) {
            KpSwEnd out;
            out.score = fits ? best : 0;
            out.er = best > 0 ? best_r : q0;
            out.eb = (best > 0 ? best_b : 0) | (sn ? KP_SWEND_HAS_N : 0);
            out.trace_off = (uint32_t)toff;
            ends[ti] = out;
        }
    }
}

__global__ __launch_bounds__(64) void kp_sw_long_kernel(KpBatchView b, KpGenes genes, const KpTask *__restrict__ tasks,
                                                        const uint32_t *__restrict__ order_counts, uint32_t task_cap,
                                                        const uint32_t *__restrict__ order, KpSwEnd *__restrict__ ends,
                                                        uint4 *__restrict__ trace, unsigned long long *__restrict__ trace_top,
                                                        uint64_t trace_cap) {
    // class c's long tasks follow its ordinary ones in the order: [counts[c], counts[c] + counts[KP_N_CLASSES + c])
    const int c = blockIdx.y;
    const uint32_t first = order_counts[c], n = order_counts[KP_N_CLASSES + c];
    if (n == 0) return;
    const size_t off = (size_t)c * task_cap;
    if (c == 3) sw_long_class<32>(b, genes, tasks + off, order + off + first, n, ends + off, trace, trace_top, trace_cap, blockIdx.x, gridDim.x);
    else if (c == 2) sw_long_class<16>(b, genes, tasks + off, order + off + first, n, ends + off, trace, trace_top, trace_cap, blockIdx.x, gridDim.x);
    else if (c == 1) sw_long_class<8>(b, genes, tasks + off, order + off + first, n, ends + off, trace, trace_top, trace_cap, blockIdx.x, gridDim.x);
    else sw_long_class<4>(b, genes, tasks + off, order + off + first, n, ends + off, trace, trace_top, trace_cap, blockIdx.x, gridDim.x);
}

// ---- traceback: one lane per task -------------------------------------------------------------------------------------------
// Cell (row r, band index bi) sits on target position lo + r + bi; a diagonal step keeps bi, a step to the left (E, gap
// in the query) lowers it, a step up (F, gap in the target) raises it.  The nibble of (r, bi) is in lane stream bi / 4,
// step r + bi / 4: piece j = step / 8 (at [j / TG][lane][j % TG] of the task's block), word bi % 4 (the cell); bit layout of a word: below.
//
// A path runs along a diagonal most of the time: it stays in one lane stream and walks it backwards.  The walk therefore
// works on whole 16-byte pieces (8 steps x 4 cells) held in registers: when it stands on the last step of a piece and
// all eight nibbles of its cell say "diagonal, not the start", it takes the eight steps at once; everything else (gaps,
// the first and last steps of a path, tasks with an N, whose matches are counted base by base) goes step by step from
// the same registers.  Pieces are fetched TG at a time (contiguous bytes) and the group
// after the current one is requested a whole group ahead, so the dependent fetches of a path overlap with other
// waves' work; the direction bits are read about once (a quarter of what the fill wrote).
// Matches: without an N in the gene or the window every diagonal step scores +2 or -4, so
// score = 6 * matches - 4 * diagonal_steps - gap_costs gives the matches in closed form; tasks that saw an N (flagged by
// the fill kernel) compare the bases of every diagonal step instead.
constexpr int TB_THREADS = 256;

__device__ __forceinline__ uint32_t piece_word(const uint4 &v, int k) {  // cell k's word of a piece, in registers' terms
    const uint32_t lo = (k & 1) ? v.y : v.x, hi = (k & 1) ? v.w : v.z;
    return (k & 2) ? hi : lo;
}

__global__ __launch_bounds__(TB_THREADS) void kp_sw_traceback_kernel(KpBatchView b, KpGenes genes, const KpTask *__restrict__ tasks,
                                                              const uint32_t *__restrict__ task_count, uint32_t task_cap,
                                                              const uint32_t *__restrict__ order,
                                                              const KpSwEnd *__restrict__ ends,
                                                              const uint32_t *__restrict__ trace,
                                                              KpSwResult *__restrict__ results) {
    const int cls = blockIdx.y;
    uint32_t n = task_count[cls] + task_count[KP_N_CLASSES + cls];  // the class's order: ordinary tasks, then those of long genes
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
        int q0 = 0;  // first row the fill kernel computed for the task: its steps count from there
        if (walking) {
            const int c_abs = b.asm_first_ctg[tk.asm_id] + tk.contig;
            const int cstart = b.ctg_start[c_abs];
            int r_hi;
            kp_task_rows(tk.lo, 4 * P, cstart, cstart + b.ctg_len[c_abs], qlen, &q0, &r_hi);
        }
        const uint4 *tw = reinterpret_cast<const uint4 *>(trace) + e.trace_off;  // piece j of lane l at [(j / TG) * TG * P + TG * l + j % TG]
        int r = e.er, bi = eb, state = 0, cols = 0, matches = 0, diag = 0, gap_cost = 0, gap = 0, credit = 0;
        int sr = r, sb = bi;
        // curq = the TG pieces (contiguous bytes) of the lane stream the walk stands in, nxtq = the TG before them
        // (requested a whole group ahead).  A group is fetched with loads in a row: one trip to memory --
        // with a load per piece the line had left the L2 by the time the walk came back for the next one (75 % misses,
        // one random 64-byte fetch per piece: the kernel ran at the rate HBM serves those, tools/microbench/l2_gather.hip)
        uint4 curq[TG], nxtq[TG];
#pragma unroll
        for (int i = 0; i < TG; ++i) curq[i] = nxtq[i] = make_uint4(0, 0, 0, 0);
        int cur_tag = -1, nxt_tag = -1;  // (stream << 20) | group index
        while (__any(walking)) {
            if (!walking) continue;
            const int l = bi >> 2, k = bi & 3, step = r - q0 + l;
            const int pc = step >> 3, grp = pc / TG, tag = (l << 20) | grp;
            if (tag != cur_tag) {
                const uint4 *stream = tw + TG * l;
                if (tag == nxt_tag) {
#pragma unroll
                    for (int i = 0; i < TG; ++i) curq[i] = nxtq[i];
                } else {
#pragma unroll
                    for (int i = 0; i < TG; ++i) curq[i] = stream[(size_t)grp * (TG * P) + i];
                }
                cur_tag = tag;
                if (grp > 0) {
#pragma unroll
                    for (int i = 0; i < TG; ++i) nxtq[i] = stream[(size_t)(grp - 1) * (TG * P) + i];
                    nxt_tag = tag - 1;
                }
            }
            uint4 cur = (pc & 1) ? curq[1] : curq[0];
            if (TG == 4) {
                const uint4 hi2 = (pc & 1) ? curq[TG - 1] : curq[TG - 2];
                cur = (pc & 2) ? hi2 : cur;
            }
            const uint32_t word = piece_word(cur, k);
            if (state == 0 && !has_n && (step & 7) == 7 && (word & 0xAAAAAAAAu) == 0u) {  // eight plain diagonal steps (D and L are stored inverted)
                // ... and the eight before them when they are the same cell's other piece of the group in hand (a diagonal step
                // stays on its lane and cell): a wave goes round this loop as often as its slowest lane, and most paths are plain
                int n = 8;
#ifndef KP_TB_NO16
                if (TG == 2 && (pc & 1) && (piece_word(curq[0], k) & 0xAAAAAAAAu) == 0u) n = 16;
#endif
                cols += n; diag += n; r -= n;
                continue;
            }
            // a cell's word: steps 0-3 in the low half, 4-7 in the high half; per half a byte of [L, F opened] pairs below
            // a byte of [D, E opened] pairs, step j's pair at bits 2j+1, 2j
            const uint32_t half = word >> (16 * ((step >> 2) & 1)), sh = 2 * (step & 3);
            const uint32_t de = ((half >> (8 + sh)) & 3u) ^ 2u, lf = ((half >> sh) & 3u) ^ 2u;  // (D, L: stored inverted)
            const uint32_t nib = ((de & 2u) << 2) | ((lf & 2u) << 1) | ((de & 1u) << 1) | (lf & 1u);  // [D][L][EO][FO]
            // -> source 0 = diagonal, 1 = diagonal and the path starts here, 2 = E, 3 = F
            const uint32_t src = (nib & 8u) ? ((nib & 4u) ? 0u : 1u) : ((nib & 4u) ? 2u : 3u);
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

void kp_launch_sw(const KpBatchView &b, const KpGenes &genes, const KpTask *tasks, const uint32_t *task_count /* of each class's order */,
                  uint32_t task_cap, const uint32_t *order, KpSwEnd *ends, void *trace, unsigned long long *trace_top,
                  uint64_t trace_cap_units, KpSwResult *results, bool has_long_genes, hipStream_t stream,
                  hipEvent_t after_fill) {
    const dim3 grid(3 * WIDE_BLOCKS + 256u * NARROW_BLOCKS_PER_CU + 3 * HELP_BLOCKS), block(64);
    hipLaunchKernelGGL(kp_sw_kernel, grid, block, 0, stream, b, genes, tasks, task_count, task_cap, order, ends,
                       reinterpret_cast<uint4 *>(trace), trace_top, trace_cap_units);
    if (has_long_genes)  // (a database property: the Kaptive-shaped ones have none and never launch it)
        hipLaunchKernelGGL(kp_sw_long_kernel, dim3(256, KP_N_CLASSES), block, 0, stream, b, genes, tasks, task_count, task_cap, order, ends,
                           reinterpret_cast<uint4 *>(trace), trace_top, trace_cap_units);
    if (after_fill) (void)hipEventRecord(after_fill, stream);
    hipLaunchKernelGGL(kp_sw_traceback_kernel, dim3(2048, KP_N_CLASSES), dim3(TB_THREADS), 0, stream, b, genes, tasks, task_count,
                       task_cap, order, ends, reinterpret_cast<const uint32_t *>(trace), results);
}
