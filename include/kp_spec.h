/* kp_spec.h -- constants of the kaptive_amd nucleotide aligner ("kp-align v5") and of the packed data layout.
 *
 * The reference delegates gene-vs-contig alignment to the third-party rammappy 0.1.3 wheel
 * (src/kaptive/serotyping/core.py:147-155), whose source is not in the reference tree; parity at that stage is
 * UNPINNED. This header is therefore the specification of the replacement: the HIP kernels
 * (kaptive_amd/csrc) and the CPU restatement (oracle/kp_oracle.c) both include it and must agree bit for bit.
 * Parameters follow minimap2's documented defaults where the design has an equivalent (k, scoring, -s 80 peak score,
 * >=3 seeds and >=40 seeded bases per chain, chaining of anchors across diagonal jumps of up to 500); DESIGN.md lists every deviation.
 */
#ifndef KP_SPEC_H
#define KP_SPEC_H

#include <stdint.h>

/* ---- base codes -------------------------------------------------------------------------------------------------
 * A C G T(U) = 0 1 2 3, case-insensitive; every other byte is "N" (code 4).  Packed streams hold 2 bits per base,
 * 16 bases per little-endian uint32 (base i of a word at bits 2i..2i+1); N positions are stored as code 0 and listed
 * separately as sorted, disjoint [start,end) runs in the same coordinate space. */
#define KP_CODE_N 4

/* ---- assembly layout --------------------------------------------------------------------------------------------
 * An assembly is one padded coordinate space: contig c occupies [ctg_start[c], ctg_start[c]+ctg_len[c]) and every
 * ctg_start is a multiple of KP_CONTIG_ALIGN; the gap up to the next contig is code 0.  Assemblies of a batch are
 * laid back to back, each starting on a KP_ASM_ALIGN boundary. */
#define KP_CONTIG_ALIGN 32u
#define KP_ASM_ALIGN 64u

/* ---- seeding (kp-align v3): minimap2's (w = 10, k = 15) minimizers on BOTH sides ----------------------------------------
 * The reference indexes the assembly and queries it with the genes (src/kaptive/core/genome.py:177-191,
 * src/kaptive/serotyping/core.py:147-155: Aligner(index, preset=None)); with no preset minimap2 samples both with
 * mm_sketch(w = 10, k = 15).  This section restates that sampling; it is the one place where v2 (a context-free rule at
 * density 1/4) measurably left the published algorithm (profiles/concordance_r3.md).
 *
 * Value of the 15-mer that STARTS at base p of a sequence (codes c[p..p+14], all in 0..3):
 *     fwd = sum c[p+j] << 2(14-j)   (first base in the high bits)      rev = sum (3 - c[p+j]) << 2j   (its reverse complement)
 *     z = fwd < rev ? 0 : 1  (k is odd: fwd != rev)                    x(p) = kp_hash30(z ? rev : fwd)
 * A 15-mer with an ambiguous base has no value (x = +inf).  kp_hash30 is minimap2's hash64 under the 30-bit mask; it is a
 * bijection of 30-bit values, so equal x means equal canonical 15-mers.
 *
 * Which positions are SEEDS is defined by the state machine of mm_sketch, restated here (KP_W = 10, steps over the bases
 * i = 0..len-1 of one contig or gene; the 15-mer ending at base i starts at i - 14):
 *     l(i)   = number of consecutive unambiguous bases ending at i (0 at an ambiguous base)
 *     info(i) = x(i - 14) if l(i) >= KP_K else +inf;   buf = info of the last KP_W steps;   min = (+inf) initially
 *     step i:  (1) if l(i) == KP_W + KP_K - 1 and min != +inf: every buffered entry other than this step's and other than
 *                  `min` itself whose value equals min's is a seed                      [ties of the first full window]
 *              (2) if info(i) <= min:  `min` is a seed if l(i) >= KP_W + KP_K and min != +inf;  min = info(i)
 *                  else if `min` is the entry that leaves the buffer at this step:
 *                       `min` is a seed if l(i) >= KP_W + KP_K - 1;  min = the LAST smallest buffered entry;
 *                       if l(i) >= KP_W + KP_K - 1 and min != +inf: every other buffered entry equal to it is a seed
 *     after the last base: `min` is a seed if it is not +inf.
 * Away from sequence ends and ambiguous bases this is: p is a seed iff x(p) is a smallest value (ties included) of at
 * least one of the KP_W windows of KP_W consecutive 15-mers that contain it -- the form the scan kernel evaluates for
 * every position whose 15-mer starts at least KP_W bases after the start of its clean stretch and ends at least KP_W bases
 * before the end of it; the remaining positions (contig ends, the flanks of N runs) are decided by running the state
 * machine itself (kp_scan.hip: kp_edge_kernel).  Expected density 2 / (KP_W + 1).
 *
 * A gene seed (start qs on the gene's forward strand, strand bit zq) and a contig seed (start ts, strand bit zt) with the
 * same x make one ANCHOR: same strand if zq == zt, query position qs; otherwise the gene's reverse complement, query
 * position gene_len - KP_K - qs.
 *
 * OCCURRENCE CUT (v4; round 6: the quantile).  minimap2 drops a query seed that occurs more than mid_occ times in the index --
 * here: among the assembly's minimizers, both strands -- with mid_occ = max(min_mid_occ = 10, 1 + the count at position
 * (int)((1 - 2e-4f) n) of the sorted occurrence counts of the index's n distinct minimizers) (mm_idx_cal_max_occ).  A gene
 * seed with more than mid_occ anchors in an assembly (one anchor per occurrence, whichever strand) loses all of them.  The
 * quantile can matter only where some gene seed has more than KP_MID_OCC anchors (mid_occ >= 10 whatever the assembly holds),
 * and it is WORKED OUT only for an assembly in which some gene has at least KP_MIN_ANCHORS such seeds (kp_chain.hip: the block
 * that meets them sketches its assembly once more and counts every minimizer); in every other assembly the cut is the floor,
 * KP_MID_OCC.  (One or two seeds of a gene beyond the floor -- in practice a 15-mer it shares by chance with a repeat of the
 * genome: on a background with IS-like repeats 7 in 10 assemblies have one -- cannot make a chain on their own, minimap2's
 * -n 3; cutting them at the floor where minimap2 would keep some of them takes at most two anchors off a chain of the gene's
 * own copy.  Counting 0.9 M minimizers per assembly for them costs more than the rest of the pass.)  Counts are capped at KP_MID_OCC_HIST - 1 (minimap2's own cap,
 * max_mid_occ, is 10^6; an assembly whose 2e-4 quantile is a minimizer in 2 000 copies is not a bacterial genome).
 * minimap2's rescue of high-occurrence seeds in seed-poor stretches (mm_seed_select) is not restated. */
#define KP_MID_OCC 10
#define KP_MID_OCC_FRAC 2e-4f
#define KP_MID_OCC_HIST 2048
#define KP_K 15
#define KP_W 10
#define KP_KMER_MASK 0x3FFFFFFFu
#ifndef KP_SPEC_FN
#ifdef __HIPCC__ /* the HIP sources use the same statement on the device */
#define KP_SPEC_FN __host__ __device__ inline
#else
#define KP_SPEC_FN static inline
#endif
#endif
KP_SPEC_FN uint32_t kp_hash30(uint32_t key) { /* minimap2 sketch.c: hash64(key, (1 << 30) - 1), all in 32 bits */
    key = (~key + (key << 21)) & KP_KMER_MASK;
    key = key ^ key >> 24;
    key = ((key + (key << 3)) + (key << 8)) & KP_KMER_MASK;
    key = key ^ key >> 14;
    key = ((key + (key << 2)) + (key << 4)) & KP_KMER_MASK;
    key = key ^ key >> 28;
    return key; /* the last step, key + (key << 31), adds nothing below bit 30 */
}
#define KP_MAX_GENE_LEN 65535       /* query positions are 16-bit fields of the anchor key and of the hit order's keys */
#define KP_FILL16_MAX_GENE_LEN 15800 /* longest gene the packed 16-bit fill kernel takes: biased scores (<= 2 * length + 14) stay below
                                        0x7C00 (kp_sw.hip); tasks of longer genes are filled with 32-bit scores (kp_sw_long_kernel) */
#define KP_MAX_GENES 131071   /* (gene*2+strand) is stored in 18 bits */
#define KP_MAX_ASM_LEN ((1u << 30) - 65536u)

/* anchor key (sortable): [gs:18][diag:30][qpos:16], diag = tpos - qpos + 65536, gs = gene*2 + (strand<0) */
#define KP_DIAG_BIAS 65536
#define KP_ANCHOR_KEY(gs, diag, qpos) (((uint64_t)(gs) << 46) | ((uint64_t)(diag) << 16) | (uint64_t)(qpos))
#define KP_KEY_GS(k) ((uint32_t)((k) >> 46))
#define KP_KEY_DIAG(k) ((uint32_t)(((k) >> 16) & 0x3FFFFFFFu))
#define KP_KEY_QPOS(k) ((uint32_t)((k) & 0xFFFFu))

/* ---- anchors -> DP tasks ----------------------------------------------------------------------------------------
 * Anchors of one assembly are sorted by key and cut into clusters: a new cluster starts when gs changes, the contig
 * changes, the diagonal jumps by more than KP_DIAG_GAP, or the cluster would span more than KP_MAX_SPREAD diagonals.
 *
 * Which clusters become tasks follows minimap2's chaining thresholds -n 3 -m 40 (mm_chain_dp + mg_chain_backtrack):
 *   a cluster of n <= KP_CHAIN_DP_MAX anchors is chained exactly as minimap2 chains that anchor set (with so few anchors
 *   its max_skip / max_iter heuristics cannot act): anchors in (target, query) order, f[i] = max(KP_K, max over earlier j
 *   of f[j] + sc(i, j)) with the FIRST maximum met going backwards from i - 1, where for dq = q_i - q_j, dr = t_i - t_j:
 *       sc = invalid if dq <= 0, dr == 0 (dr > 0 then holds: target order) or dq > KP_CHAIN_MAX_DIST;
 *       dd = |dr - dq|, dg = min(dr, dq), sc = min(KP_K, dg), minus kp_chain_pen[dd] if dd != 0 or dg > KP_K
 *   (kp_chain_pen[dd] = int(0.12 dd + 0.5 mg_log2(dd + 1)), minimap2's gap cost for k = 15, tabulated below; dd <= the
 *   cluster's spread).  The chain is the one ending at the largest f (the later anchor on ties), walked back along the
 *   chosen predecessors and cut where the score counted from its end peaks (mg_chain_backtrack's max_i); its score is that
 *   peak.  The cluster becomes a task iff the score is >= KP_MIN_CHAIN_SCORE (which needs >= KP_MIN_ANCHORS anchors); the
 *   task's n_anchors is the chain's anchor count and chain_score its score.
 *   A cluster of more anchors becomes a task iff its anchors cover >= KP_MIN_SEED_SPAN query bases (qmax - qmin + K);
 *   n_anchors = n, chain_score = min(KP_K * n, qmax - qmin + K) (what a co-linear chain of them scores).
 * The band is the cluster's diagonal range (all its anchors) widened on both sides and centred: by
 * KP_BAND_MARGIN_NARROW when that fits 16 diagonals (anchors on one or two adjacent diagonals: no indel seen), otherwise by
 * KP_BAND_MARGIN, rounded up to 32, 64 or 128 diagonals. */
#define KP_DIAG_GAP 32
#define KP_BAND_MARGIN 15
#define KP_BAND_MARGIN_NARROW 7 /* clusters whose anchors span at most 2 diagonals get a 16-diagonal band */
#define KP_MAX_BAND 128
#define KP_MAX_SPREAD (KP_MAX_BAND - 2 * KP_BAND_MARGIN - 1)
#define KP_MIN_ANCHORS 3
#define KP_MIN_SEED_SPAN 40
#define KP_CHAIN_DP_MAX 24
#define KP_CHAIN_MAX_DIST 5000 /* minimap2's max_gap */
#define KP_CHAIN_PEN_SIZE 512 /* dd 0..500 (minimap2's bw), padded */
#define KP_CHAIN_PEN_TABLE                                                                                            \
    {0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 2, 3, 3, 3, 3, 3, 3, 4, 4, 4, 4, 4, 4, 5, 5, 5, 5, 5, 5, 5, 6, 6, \
     6, 6, 6, 6, 6, 7, 7, 7, 7, 7, 7, 7, 8, 8, 8, 8, 8, 8, 8, 8, 9, 9, 9, 9, 9, 9, 9, 10, 10, 10, 10, 10, \
     10, 10, 10, 11, 11, 11, 11, 11, 11, 11, 11, 12, 12, 12, 12, 12, 12, 12, 13, 13, 13, 13, 13, 13, 13, 13, 14, 14, 14, 14, 14, 14, \
     14, 14, 15, 15, 15, 15, 15, 15, 15, 15, 16, 16, 16, 16, 16, 16, 16, 16, 17, 17, 17, 17, 17, 17, 17, 17, 18, 18, 18, 18, 18, 18, \
     18, 18, 19, 19, 19, 19, 19, 19, 19, 19, 20, 20, 20, 20, 20, 20, 20, 20, 21, 21, 21, 21, 21, 21, 21, 21, 22, 22, 22, 22, 22, 22, \
     22, 22, 23, 23, 23, 23, 23, 23, 23, 23, 24, 24, 24, 24, 24, 24, 24, 24, 25, 25, 25, 25, 25, 25, 25, 25, 26, 26, 26, 26, 26, 26, \
     26, 26, 27, 27, 27, 27, 27, 27, 27, 27, 28, 28, 28, 28, 28, 28, 28, 28, 29, 29, 29, 29, 29, 29, 29, 29, 30, 30, 30, 30, 30, 30, \
     30, 30, 31, 31, 31, 31, 31, 31, 31, 31, 32, 32, 32, 32, 32, 32, 32, 32, 33, 33, 33, 33, 33, 33, 33, 33, 33, 34, 34, 34, 34, 34, \
     34, 34, 34, 35, 35, 35, 35, 35, 35, 35, 35, 36, 36, 36, 36, 36, 36, 36, 36, 37, 37, 37, 37, 37, 37, 37, 37, 38, 38, 38, 38, 38, \
     38, 38, 38, 39, 39, 39, 39, 39, 39, 39, 39, 39, 40, 40, 40, 40, 40, 40, 40, 40, 41, 41, 41, 41, 41, 41, 41, 41, 42, 42, 42, 42, \
     42, 42, 42, 42, 43, 43, 43, 43, 43, 43, 43, 43, 44, 44, 44, 44, 44, 44, 44, 44, 45, 45, 45, 45, 45, 45, 45, 45, 45, 46, 46, 46, \
     46, 46, 46, 46, 46, 47, 47, 47, 47, 47, 47, 47, 47, 48, 48, 48, 48, 48, 48, 48, 48, 49, 49, 49, 49, 49, 49, 49, 49, 50, 50, 50, \
     50, 50, 50, 50, 50, 50, 51, 51, 51, 51, 51, 51, 51, 51, 52, 52, 52, 52, 52, 52, 52, 52, 53, 53, 53, 53, 53, 53, 53, 53, 54, 54, \
     54, 54, 54, 54, 54, 54, 55, 55, 55, 55, 55, 55, 55, 55, 55, 56, 56, 56, 56, 56, 56, 56, 56, 57, 57, 57, 57, 57, 57, 57, 57, 58, \
     58, 58, 58, 58, 58, 58, 58, 59, 59, 59, 59, 59, 59, 59, 59, 59, 60, 60, 60, 60, 60, 60, 60, 60, 61, 61, 61, 61, 61, 61, 61, 61, \
     62, 62, 62, 62, 62, 62, 62, 62, 63, 63, 63, 63, 63, 63, 63, 63, 63, 64, 64, 64, 64, 64, 64, 64, 64, 64, 64, 64, 64, 64, 64, 64}

/* ---- banded local alignment (Smith-Waterman-Gotoh; scores <= 2 * KP_MAX_GENE_LEN) ----------------------------------------------------------
 * H = max(0, Hdiag + s, E, F);  E (gap in query, moves along the target) = max(Hleft - (O+X), Eleft - X);
 * F (gap in target, moves along the query) = max(Hup - (O+X), Fup - X).  Opening wins ties against extending; the
 * diagonal wins ties against E, E against F; a cell whose best is <= 0 is a restart cell.  The reported cell is the
 * first maximum in (query row, target column) order.  Cells outside the band or the contig read as H=0, E=F=-inf.
 * Hits whose cell score is below KP_MIN_DP_SCORE are dropped.
 *
 * The recurrence charges a gap of length n with KP_GAP_OPEN + n * KP_GAP_EXT (minimap2's first affine piece).  The score a
 * hit REPORTS is that of its path under minimap2's two-piece cost min(4 + 2n, 24 + n): every gap longer than
 * KP_GAP_LONG = 20 columns is credited n - 20 during the traceback (minimap2 itself re-scores the finished path with the
 * two-piece cost for its reported DP score).  Gaps that long need a 32-diagonal band or wider and are rare; the path is
 * still the one-piece optimum. */
#define KP_SC_MATCH 2
#define KP_SC_MISMATCH (-4)
#define KP_SC_N (-1) /* either base is N */
#define KP_GAP_OPEN 4
#define KP_GAP_EXT 2
#define KP_GAP_OPEN2 24
#define KP_GAP_EXT2 1
#define KP_GAP_LONG ((KP_GAP_OPEN2 - KP_GAP_OPEN) / (KP_GAP_EXT - KP_GAP_EXT2)) /* from here on the second piece is cheaper */
#define KP_MIN_DP_SCORE 80
#define KP_MIN_CHAIN_SCORE 40 /* minimap2 -m: floor of the secondary score in the mapping quality (kp_mapq.h) */
#define KP_MASK_LEVEL_NUM 1   /* a hit is secondary to a better hit of its gene that covers more than */
#define KP_MASK_LEVEL_DEN 2   /* KP_MASK_LEVEL_NUM / KP_MASK_LEVEL_DEN of the shorter one's query span (minimap2 -M 0.5) */
#define KP_NEG_INF (-(1 << 29))

/* ---- kp-align v5: chains of ANCHORS across diagonal jumps of up to KP_JOIN_BW (minimap2's bw = 500) --------------------------
 * minimap2 chains anchors whose diagonals differ by up to bw = 500 and aligns through the gap; a gene with an insertion or
 * deletion of 33-500 bases is ONE hit there -- also when the stretch beyond the event is too short to be a chain of its own
 * (an event next to a gene's end), and when two events nearly cancel so that the stretches before and after them share a
 * diagonal.  The clusters above stop at KP_DIAG_GAP; the JOIN sits on top of them and leaves everything else as it was: every
 * accepted cluster is still a band task with a hit of its own unless a chain CONSUMES it (below).
 *
 * GROUPS.  Every cluster counts, weak ones (fewer than KP_MIN_ANCHORS anchors or than KP_MIN_SEED_SPAN query bases: no band
 * task) included; a cluster is PROVISIONAL when it has >= KP_MIN_ANCHORS anchors covering >= KP_MIN_SEED_SPAN query bases (it
 * may still be rejected by its chain score).  Per gene/strand AND contig, take the clusters in the order of the sorted
 * anchors (diagonals are in the assembly's padded coordinates: near a contig boundary the clusters of two contigs interleave
 * in that order, which is why the contig is part of the key): a cluster continues the open sequence of its contig iff its
 * lowest diagonal is at most KP_JOIN_BW above the highest diagonal of the sequence's last cluster and the sequence holds
 * fewer than KP_JOIN_GROUP_MAX clusters; otherwise it closes that sequence and opens a new one.  At most KP_JOIN_OPEN
 * sequences of a gene/strand are open at a time: the cluster of a further contig closes the one whose last cluster ends on
 * the lowest diagonal (the lowest contig on ties).  A closed sequence of two or more clusters, at least one of them
 * provisional, whose clusters hold KP_MIN_ANCHORS..KP_JOIN_ANCHOR_MAX anchors in all, is a GROUP.
 *
 * CHAINS OF ANCHORS.  All anchors of the group's clusters, (t, q) = (target, query) position, t = q + diagonal, in (t, q)
 * order, are chained as mm_chain_dp chains them -- exactly, i.e. without its max_skip / max_iter shortcuts:
 *     f[i] = max(KP_K, max over earlier j of f[j] + sc(i, j)),  the FIRST maximum met going backwards from i - 1,
 * sc as for a cluster's chain (above: dq = q_i - q_j, dr = t_i - t_j; invalid if dq <= 0, dr == 0, dq or dr > KP_CHAIN_MAX_DIST;
 * dd = |dr - dq|, dg = min(dr, dq); sc = min(KP_K, dg) minus kp_chain_pen[dd] if dd != 0 or dg > KP_K) and also invalid if
 * dd > KP_JOIN_BW.  Backtracking as mg_chain_backtrack: repeatedly take the unused anchor with the largest f >=
 * KP_MIN_CHAIN_SCORE (the later one on ties) and walk its predecessors until a used anchor or the start, stopping early once
 * the score counted from the end has fallen KP_JOIN_BW below its peak; the chain is cut at that peak, its score is the peak
 * and its anchors become used -- also when it is then discarded for scoring below KP_MIN_CHAIN_SCORE or holding fewer than
 * KP_MIN_ANCHORS anchors.
 *
 * PIECES.  A chain's anchors in query order fall into pieces: a new piece starts where the diagonal differs from the anchor
 * before by more than KP_DIAG_GAP, or where the piece would come to span more than KP_MAX_SPREAD diagonals.  A chain of one
 * piece is what its cluster's band task already aligns; a chain of 2..KP_JOIN_MAX_PIECES pieces is a JOIN (more pieces: none).
 * Every piece gets a band of the same width W and margin M -- those of a cluster's band task for the widest piece's diagonal
 * range: 16 diagonals and KP_BAND_MARGIN_NARROW when that holds it, otherwise the smallest of 32, 64, 128 with KP_BAND_MARGIN
 * (kp_piece_margin / kp_piece_width below) --, centred on its own range:  lo[k] = dmin[k] - M - (W - need[k]) / 2,
 * need[k] = dmax[k] - dmin[k] + 1 + 2 M.  n_anchors and chain score of the join are the chain's.
 *
 * JOINED FILL.  Every piece is filled as a band task (the recurrence above: local, H >= 0, restarts) over the ROWS between its
 * neighbours' anchors: piece k covers rows [R0, R1), R0 = 0 for the first piece, otherwise the query position of the LAST anchor
 * of piece k - 1 rounded down to a multiple of 8; R1 = the gene's length for the last piece, otherwise the query position of
 * the FIRST anchor of piece k + 1 plus KP_K (a chain's alignment runs through its anchors: a piece is left after its last
 * anchor and entered before its first); rows outside read as outside the contig (H = 0, E = F = -inf).  In a
 * piece k > 0, H has two more candidates, the CROSS gaps from piece k - 1.  When piece k lies on higher diagonals
 * (lo[k] > lo[k-1]: an insertion in the contig) a gap along row r from a live cell (r, t') of piece k - 1 left of piece k's
 * band in that row (t' < lo[k] + r):
 *     X1 = max (H(r, t') + 2 t') - 4 - 2 t        X2 = max (H(r, t') + t') - 24 - t      (first / second piece of the gap cost)
 * otherwise (a deletion) a gap down column t from a live cell (r', t) of piece k - 1 on a diagonal above piece k's band
 * (t - r' > lo[k] + W - 1):   X1 = max (H(r', t) + 2 r') - 4 - 2 r,   X2 = max (H(r', t) + r') - 24 - r.
 * (Live: H > 0.)  A cross gap starts and ends in rows of the JUNCTION ZONE, the rows the two pieces share: r (and r') in
 * [R0 of piece k, R1 of piece k - 1) -- the stretch of the query between the last anchor of the piece before and the first
 * anchor of this one.  The source cell of a maximum is the first one in increasing t' (r').  H = max(diagonal, E, F, X1, X2) with
 * ties in that order (a candidate must beat everything before it); H <= 0 is a restart cell.  A path therefore crosses a gap
 * only where that pays -- which is what minimap2 comes to: it cuts a chain's end off when it is shorter than twice the gap
 * behind it (mm_fix_bad_ends) and lets the extension, a free local decision, reach across or not; an end long enough to be
 * kept always pays for its gap at the identities Kaptive works with.
 *
 * THE JOINED PATH.  The best cell of a piece is its first maximum of H in (row, column) order.  The pieces are tried in the
 * order of their best cells' scores (the earlier piece on ties): nothing is reported from a best cell below KP_MIN_DP_SCORE;
 * otherwise walk back from it (through cross gaps into earlier pieces, to where the path starts).  On the way: suf = score of
 * the part of the path behind the current cell with the costs of the cross gaps left out, sufmax = its largest value at a cell
 * in state H so far; at a cell that takes a cross gap, and at every cell in state H once a gap has been crossed, if
 * sufmax - suf > KP_JOIN_DROP the path is REJECTED (minimap2's z-drop of 400 against the open cost of the second gap piece,
 * which forgives a gap its length: mm_test_zdrop would have split the chain there), the piece is set aside and the next one
 * tried.  The first path that is not
 * rejected settles the chain: if it crosses at least one gap it is the JOINED HIT -- score = H(best cell) + the long-gap
 * credit of its in-band gaps, coordinates from its first and last cell, columns = cells + gap columns, matches counted base
 * by base, n_seeds and chain score those of the chain, plus the BONUS of its order score (below) --; if it crosses none, the
 * band task of that piece's cluster already reports it.
 *
 * CONSUMED PIECES.  minimap2 reports one alignment per chain (two where it splits one), and it cuts a chain's end off before
 * aligning when that end is shorter than twice the gap behind it (mm_fix_bad_ends): such an end is reported only if the
 * extension reaches across the gap.  So: the band tasks of the clusters that hold an anchor of a piece the joined hit runs
 * through report no hit of their own; and, unless a path of the chain was rejected, neither do those of the pieces of the
 * chain's WEAK ENDS (kp_weak_ends below) that the settled path does not run through.
 *
 * Not restated: minimap2's max_skip heuristic in the chaining DP, its mm_filter_bad_seeds (anchors between gaps that nearly
 * cancel are skipped as fill boundaries: the joined fill has no fill boundaries inside a piece), an extension that reaches
 * across a gap to a stretch WITHOUT any anchor (its band of 751 diagonals finds it, a piece needs an anchor to exist), the
 * re-chaining with bw_long = 20 000. */
#define KP_JOIN_BW 500
/* WEAK ENDS of a chain of n pieces (kp_spec.h, CONSUMED PIECES; minimap2's mm_fix_bad_ends restated on pieces).  qlo / qhi:
 * query position of every piece's first / last anchor; jump_before[k]: the diagonal jump between piece k - 1 and k.  Going in
 * from the first piece, L = query bases from the chain's first anchor to the end of piece k: pieces 0..k are a weak end if the
 * jump behind piece k exceeds L / 2; the walk ends once L >= KP_JOIN_BW or 2 L >= the chain's query extent.  Likewise from
 * the last piece.  Returns the mask of weak-end pieces. */
KP_SPEC_FN int kp_weak_ends(int n, const int *qlo, const int *qhi, const int *jump_before) {
    const int extent = qhi[n - 1] + KP_K - qlo[0];
    int mask = 0;
    for (int k = 0; k + 1 < n; ++k) {
        const int L = qhi[k] + KP_K - qlo[0];
        if (2 * jump_before[k + 1] > L) mask |= (1 << (k + 1)) - 1;
        if (L >= KP_JOIN_BW || 2 * L >= extent) break;
    }
    for (int k = n - 1; k >= 1; --k) {
        const int L = qhi[n - 1] + KP_K - qlo[k];
        if (2 * jump_before[k] > L) mask |= ((1 << n) - 1) & ~((1 << k) - 1);
        if (L >= KP_JOIN_BW || 2 * L >= extent) break;
    }
    return mask;
}
/* ORDER SCORE.  minimap2 orders a query's hits, filters them (-s) and computes mapping qualities with dp_max -- the best
 * running score of the path when a gap of n columns is charged KP_GAP_OPEN + 2 log2(1 + n) -- not with the alignment score.
 * For band tasks the two are as good as equal; a joined path's cross gaps make them differ by hundreds.  A joined hit
 * therefore carries a BONUS = sum over its cross gaps of (cost charged - (KP_GAP_OPEN + kp_log2x2(n))), floored at 0 per
 * gap and capped at KP_HIT_BONUS_MAX; order score = score + bonus takes the place of the score in the emission order
 * and in the mapping quality.  Until a hit's mapping quality is set the bonus rides in the bits of `score` from
 * KP_HIT_BONUS_SHIFT up; finished hit records hold the plain score. */
#define KP_HIT_BONUS_SHIFT 20
#define KP_HIT_BONUS_MAX 2047
#define KP_HIT_SCORE(field) ((int32_t)((uint32_t)(field) & ((1u << KP_HIT_BONUS_SHIFT) - 1u)))
#define KP_HIT_OSCORE(field) (KP_HIT_SCORE(field) + (int32_t)((uint32_t)(field) >> KP_HIT_BONUS_SHIFT))
KP_SPEC_FN int kp_log2x2(uint32_t n) { /* 2 log2(1 + n) to the nearest integer or so: exponent + the top mantissa bits, integers only */
    const uint32_t m = n + 1u;
    int e = 0;
    while ((m >> e) > 1u) ++e;
    const uint32_t mant = e >= 4 ? (m >> (e - 4)) & 15u : (m << (4 - e)) & 15u;
    return 2 * e + (mant >= 3u) + (mant >= 11u);
}
/* band of a join's pieces from the widest piece's diagonal range `spread` (largest minus smallest diagonal of its anchors): the
 * rule of a cluster's band task -- KP_BAND_MARGIN_NARROW either side when that fits 16 diagonals, KP_BAND_MARGIN otherwise */
KP_SPEC_FN int kp_piece_margin(int spread) { return spread + 1 + 2 * KP_BAND_MARGIN_NARROW <= 16 ? KP_BAND_MARGIN_NARROW : KP_BAND_MARGIN; }
KP_SPEC_FN int kp_piece_width(int spread) {
    const int need = spread + 1 + 2 * kp_piece_margin(spread);
    return need <= 16 ? 16 : (need <= 32 ? 32 : (need <= 64 ? 64 : 128));
}
#define KP_JOIN_GROUP_MAX 16
#define KP_JOIN_ANCHOR_MAX 4096 /* groups with more anchors than this are not chained (a gene beyond ~22 kb whose group holds them all) */
#define KP_JOIN_OPEN 4
#define KP_JOIN_MAX_PIECES 8
#define KP_JOIN_DROP (400 - KP_GAP_OPEN2)

/* ---- protein alignment (restates src/kaptive/core/pairwise.py:395-584) ------------------------------------------------ */
#define KP_PROT_GAP_OPEN 11
#define KP_PROT_GAP_EXT 1
#define KP_PROT_K 20
#define KP_PROT_NEG_INF (-1000000000)
#define KP_PROT_FILL (-128) /* BLOSUM62 lookup value for bytes outside ARNDCQEGHILKMFPSTWYVBJZX* */

/* ---- hit record (one per reported alignment; emission order: gene asc, score desc, contig asc, t_start asc,
 * forward strand first, q_start asc, q_end asc, t_end asc, matches desc, block_len asc, seeds desc; hits with the same
 * span -- gene, contig, strand and both intervals -- are emitted once, the first of them in that order).
 * Mapping quality: walking a gene's hits in emission order, a hit is SECONDARY (mapq 0) when its query span overlaps
 * that of an earlier primary hit of the gene by more than KP_MASK_LEVEL of the shorter span; it then counts towards that
 * primary's n_sub and best secondary score.  Every other hit is PRIMARY, mapq = kp_mapq_value (kp_mapq.h). ------------ */
typedef struct kp_hit {
    int32_t gene;    /* database gene index */
    int32_t contig;  /* contig index within the assembly */
    int32_t q_start; /* on the gene's forward strand, 0-based half-open */
    int32_t q_end;
    int32_t t_start; /* on the contig's forward strand, 0-based half-open */
    int32_t t_end;
    int32_t score;
    int32_t matches;   /* identical aligned bases */
    int32_t block_len; /* alignment columns */
    int8_t strand;     /* +1 / -1 */
    uint8_t mapq;      /* kp_mapq.h; 0 for secondary hits */
    uint8_t n_seeds;   /* anchors of the band task behind the hit, capped at 255 */
    uint8_t pad_;
} kp_hit;

/* ---- records of the batched reduction (one assembly = one summary, its kept hits and its locus pieces) ------------------
 * They carry what SerotypingResult needs (src/kaptive/serotyping/models.py:513-536) minus strings and sequences. */
#define KP_F_EXPECTED 1u /* gene belongs to the best locus and is not an extra gene (core.py:226-228) */
#define KP_F_INSIDE 2u   /* overlaps a locus piece (core.py:273-278) */
#define KP_F_EXTRA 4u    /* gene of an "Extra genes" record */
#define KP_F_PARTIAL 8u  /* clipped by a contig edge (alignment.py:774-809) */
#define KP_F_SPURIOUS 16u /* outside the locus and below the identity threshold: dropped from the result (core.py:383) */
#define KP_F_PRIMARY 32u /* top-scoring kept hit of an expected gene (core.py:236-245) */

#define KP_STATE_NORMAL 0
#define KP_STATE_PARTIAL 1
#define KP_STATE_TRUNCATED 2
#define KP_STATE_NOVEL 3

#define KP_MAX_LOCUS_GENES 256 /* width of the missing-gene mask */

typedef struct kp_kept { /* one hit that survived the overlap cull, emission order */
    int32_t gene, contig, q_start, q_end, t_start, t_end, score;
    int32_t prot_off, prot_len; /* translated protein inside the assembly's protein buffer */
    int32_t cluster;            /* spatial cluster id */
    int32_t dp[8];              /* protein DP: score, matches, mismatches, gaps, q_start, q_end, t_start, t_end */
    float pident, coverage;     /* float32, as the reference stores them */
    int8_t strand, state;
    uint8_t flags, pad_;
} kp_kept;

typedef struct kp_piece {
    int32_t contig, start, end, strand;
    double mean_pos; /* mean expected position of the piece's primary hits; the host orders pieces by it */
} kp_piece;

typedef struct kp_asm_summary {
    int32_t n_hits, n_kept, n_final, n_pieces;
    int32_t best_locus, n_expected, n_missing;
    int32_t overflow; /* bit0 kept list, bit1 pieces, bit2 locus wider than the mask, bit3 protein buffer */
    uint64_t missing_mask[KP_MAX_LOCUS_GENES / 64]; /* bit j: gene locus_gene_off + j not found inside the locus */
    float ident_sum;   /* float32 sum of the identities of NORMAL genes, associated exactly as np.add.reduce does */
    int32_t n_normal;  /* how many; mean identity = float32(float64(ident_sum) / n_normal) (core.py:395-396) */
} kp_asm_summary;

typedef struct kp_typing_params {
    double min_gene_coverage; /* Serotyper.min_gene_coverage */
    float id_threshold;       /* np.float32(metadata.id_threshold): identities are compared as float32 */
    int32_t max_locus_length; /* Database.max_locus_length, the clustering tolerance */
    int32_t edge_tolerance;   /* Serotyper.partial_edge_tolerance */
} kp_typing_params;

#endif /* KP_SPEC_H */
