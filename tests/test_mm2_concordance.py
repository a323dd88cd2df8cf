"""kp-align next to an independent model of the published minimap2 pipeline (oracle/mm2_model.c).  CPU only.

The reference's aligner is a closed wheel, so kp-align cannot be compared with it.  What can be
checked here, on every machine: (1) the model itself behaves like the published algorithm on known answers; (2) on the
config-2/3/4 generators both aligners, fed through the SAME reduction (`Serotyper.reduce`, pinned to the reference by
tests/golden/typing_*.npz), produce BYTE-IDENTICAL report rows for every assembly, report the same raw hits (at most 1 %
one-sided either way) with equal scores, equal chain anchor counts and, on at least 98 % of them, equal mapping quality;
(3) the oracle's seeds (kp_oracle.c, the state machine of include/kp_spec.h) equal the model's mm_sketch on random,
ambiguous, low-complexity and periodic sequences -- two independent restatements of the same published routine.  The
560-assembly run of the same comparison through the REFERENCE's own Serotyper is tools/concordance.py ->
profiles/concordance_r4.md (build container only).
"""

from __future__ import annotations

import numpy as np
import pytest

from kaptive_amd.core.pairwise import PairwiseAlignments
from kaptive_amd.pack import pack_sequences_flat
from kaptive_amd.serotyping.core import Serotyper
from kaptive_amd.serotyping.io import KaptiveRow
from kaptive_amd.synth import make_assembly, make_db, mutate, random_dna
from oracle import mm2
from oracle import oracle as O
from tests.golden_util import hits_to_alignments


# ---- the model on known answers ----------------------------------------------------------------------------------------
def test_ksw_known_answers():
    assert mm2.ksw(b"ACGTACGTAC", b"ACGTACGTAC")[::5] == (20, "10M")
    # one substitution: 17 matches * 2 - 4
    assert mm2.ksw(b"ACGTACGTACGGTTAACC", b"ACGTACGAACGGTTAACC")[::5] == (30, "18M")
    # a 3-base deletion is charged 4 + 2 * 3 and is left-aligned
    assert mm2.ksw(b"ACGTACGTACGGTTAACC", b"ACGTACGTACTTTGGTTAACC")[::5] == (26, "10M3D8M")
    rng = np.random.default_rng(1)
    a = random_dna(rng, 300, 0.5).tobytes()
    b = a[:150] + random_dna(rng, 60, 0.5).tobytes() + a[150:]
    score, *_, cigar = mm2.ksw(a, b)
    import re

    ops = re.findall(r"(\d+)([MID])", cigar)  # one 60-base deletion (left-aligned: it may slide over chance-equal bases)
    assert score == 600 - (24 + 60) and [o for _, o in ops] == ["M", "D", "M"] and int(ops[1][0]) == 60  # second piece: 24 + n
    assert int(ops[0][0]) + int(ops[2][0]) == 300 and int(ops[0][0]) <= 150
    # extension stops at the best cell; a z-drop ends the search
    _, mx, mt, mq, dropped, cigar = mm2.ksw(b"ACGTACGTACGGTTAACC" + b"T" * 30, b"ACGTACGTACGGTTAACC" + b"GA" * 15, zdrop=400, ext_only=True)
    assert (mx, mt, mq, dropped, cigar) == (36, 17, 17, 0, "18M")
    junk_q, junk_t = random_dna(rng, 600, 0.5).tobytes(), random_dna(rng, 600, 0.5).tobytes()
    _, mx, mt, mq, dropped, _ = mm2.ksw(a[:100] + junk_q, a[:100] + junk_t, zdrop=100, ext_only=True)
    assert dropped == 1 and mx >= 200 and 99 <= mt <= 110 and 99 <= mq <= 110


def test_sketch_density_strand_symmetry_and_window_guarantee():
    rng = np.random.default_rng(2)
    seq = random_dna(rng, 20_000, 0.5).tobytes()
    x, y = mm2.sketch(seq)
    assert abs(len(x) / len(seq) - 2 / 11) < 0.01  # (w = 10: 2 / (w + 1))
    pos = (y & 0xFFFFFFFF) >> 1
    assert np.all(np.diff(pos.astype(np.int64)) <= 10) and np.all(np.diff(pos.astype(np.int64)) >= 0)
    # the reverse complement has the same minimizers (hashes of canonical k-mers), mirrored
    comp = bytes.maketrans(b"ACGT", b"TGCA")
    xr, yr = mm2.sketch(seq.translate(comp)[::-1])
    assert sorted((x >> 8).tolist()) == sorted((xr >> 8).tolist())
    posr = (yr & 0xFFFFFFFF) >> 1
    assert sorted((len(seq) - 1 - posr.astype(np.int64) + 14).tolist()) == sorted(pos.astype(np.int64).tolist())
    # an N resets the window
    xn, _ = mm2.sketch(seq[:500] + b"N" + seq[501:1000])
    assert 0 < len(xn) <= len(mm2.sketch(seq[:1000])[0]) + 2


def test_planted_gene_maps_full_length_on_both_strands():
    db = make_db("kpsc_k", seed=7, n_loci=3)
    rng = np.random.default_rng(5)
    g = 4
    o, n = int(db.genes.offsets[g]), int(db.genes.lengths[g])
    gene = db.genes.seqs[o : o + n]
    comp = bytes.maketrans(b"ACGT", b"TGCA")
    fwd = np.concatenate([random_dna(rng, 3000, 0.5), gene, random_dna(rng, 3000, 0.5)])
    rev = np.concatenate([random_dna(rng, 2000, 0.5), np.frombuffer(gene.tobytes().translate(comp)[::-1], np.uint8),
                          random_dna(rng, 500, 0.5)])  # fmt: skip
    seqs = np.concatenate([fwd, rev])
    idx = mm2.Mm2Index(seqs, [0, len(fwd)], [len(fwd), len(rev)])
    assert idx.mid_occ == 10  # -f 2e-4 on a small index falls to min_mid_occ
    hits = idx.map(db.genes)
    mine = hits[hits["gene"] == g]
    assert len(mine) == 2
    for h, (ctg, strand, t0) in zip(sorted(mine, key=lambda r: r["contig"]), ((0, 1, 3000), (1, -1, 2000))):
        assert (h["contig"], h["strand"], h["q_start"], h["q_end"], h["t_start"], h["t_end"]) == (ctg, strand, 0, n, t0, t0 + n)
        assert h["score"] == 2 * n and h["matches"] == n and h["block_len"] == n
        assert h["mapq"] == 0  # two equally good placements of the query: mm_set_mapq gives both 0


# ---- concordance of the calls ------------------------------------------------------------------------------------------------
def _oracle_proteins(q, t):
    return PairwiseAlignments.from_table(O.protein_align(q.seqs, q.offsets, q.lengths, t.seqs, t.offsets, t.lengths))


CONFIGS = {
    # SURVEY.md section 8d generators at reduced background length (the iid background only adds chance seeds)
    "config2": dict(db=("kpsc_k", 100), also=None, kw=dict(length=300_000, median_contigs=8)),
    "config3": dict(db=("kpsc_k", 100), also=("kpsc_o", 101), kw=dict(length=300_000, median_contigs=8)),
    "config4": dict(db=("ab_k", 102), also=None, kw=dict(length=300_000, median_contigs=110, min_contig=200, force_split=True)),
}


@pytest.fixture(scope="module")
def dbs():
    return {}


def _db(cache, kind, seed):
    if (kind, seed) not in cache:
        db = make_db(kind, seed=seed)
        cache[(kind, seed)] = (db, O.OracleDB(*pack_sequences_flat(db.genes)),
                               Serotyper(db, aligner=lambda g: None, protein_aligner=_oracle_proteins))  # fmt: skip
    return cache[(kind, seed)]


def _span(h):
    return (int(h["gene"]), int(h["contig"]), int(h["strand"]), int(h["q_start"]), int(h["q_end"]), int(h["t_start"]), int(h["t_end"]))


@pytest.mark.parametrize("config", sorted(CONFIGS))
def test_locus_type_and_confidence_agree_with_the_minimap2_model(dbs, config):
    cfg = CONFIGS[config]
    main = _db(dbs, *cfg["db"])
    typed = [main] + ([_db(dbs, *cfg["also"])] if cfg["also"] else [])
    shared = scores_equal = only_k = only_m = mapq_equal = seeds_equal = 0
    for i in range(4):
        genome = make_assembly(main[0], seed=7000 + 37 * i, also=tuple(t[0] for t in typed[1:]), **cfg["kw"])
        packed = genome.packed()
        index = mm2.Mm2Index.from_contigs(genome.contigs)
        for db, odb, typer in typed:
            hk, hm = odb.align(packed), index.map(db.genes)
            rk = typer.reduce(genome, hits_to_alignments(db, genome, hk))
            rm = typer.reduce(genome, hits_to_alignments(db, genome, hm))
            for field in ("best_locus_name", "phenotype", "typeable"):
                assert getattr(rk, field) == getattr(rm, field), (config, genome.id, db.metadata.keyword, field)
            fk = bytes(KaptiveRow.from_result(rk)).split(b"\t")
            fm = bytes(KaptiveRow.from_result(rm)).split(b"\t")
            assert fk == fm, (config, genome.id, [i for i in range(len(fk)) if fk[i] != fm[i]])  # the whole row
            sk, sm = {_span(h): h for h in hk}, {_span(h): h for h in hm}
            both = set(sk) & set(sm)
            shared += len(both)
            only_k += len(sk) - len(both)
            only_m += len(sm) - len(both)
            scores_equal += sum(int(sk[s]["score"]) == int(sm[s]["score"]) for s in both)
            mapq_equal += sum(int(sk[s]["mapq"]) == int(sm[s]["mapq"]) for s in both)
            seeds_equal += sum(int(sk[s]["n_seeds"]) == int(sm[s]["n_seeds"]) for s in both)
    # same seeds, same chains: what is left are alignment ends of equal score and the best-segment score (dp_max) that
    # minimap2's mapq reads where kp-align has the alignment score
    assert shared > 300 and scores_equal >= 0.995 * shared, (shared, scores_equal)
    assert only_m <= 0.01 * (shared + only_m), "hits of the minimap2 model that kp-align does not report with the same span"
    assert only_k <= 0.01 * (shared + only_k), "hits of kp-align that the minimap2 model does not report with the same span"
    assert seeds_equal >= 0.999 * shared and mapq_equal >= 0.98 * shared, (shared, seeds_equal, mapq_equal)


@pytest.mark.parametrize("kind", ["del", "ins"])
def test_mid_size_indels_give_the_models_rows(dbs, kind):
    """kp-align v4 / v5 (kp_spec.h): a gene with an insertion or deletion of 33-450 bases is ONE hit through the gap, as in the
    minimap2 model (bw = 500), not two half-gene hits -- byte-identical report rows, and every joined hit of the oracle is
    a hit of the model with the same span, score and anchor count."""
    db, odb, typer = _db(dbs, "kpsc_k", 100)
    joined = narrow_off = 0
    for i, size in ((0, 33), (2, 64), (4, 150), (5, 300), (6, 450)):  # (all seven sizes, both databases: tools/concordance.py)
        # (seed 7351 plants its 300-base insertion 390 bases into a 696-base gene: the model's left extension gets a target window
        # of twice the query it has left, too short for the insertion plus the bases before the first anchor -- its hit starts 6
        # bases late; profiles/concordance_r6.md, "extension windows")
        seed = 7300 + 10 * i + (kind == "ins") + (100 if (size, kind) == (300, "ins") else 0)
        genome = make_assembly(db, seed=seed, length=300_000, median_contigs=8, p_is=0, p_stop=0,
                               mid_indels=((size, kind), (size + 1, kind)))
        packed = genome.packed()
        hk, hm = odb.align(packed), mm2.Mm2Index.from_contigs(genome.contigs).map(db.genes)
        fk = bytes(KaptiveRow.from_result(typer.reduce(genome, hits_to_alignments(db, genome, hk)))).split(b"\t")
        fm = bytes(KaptiveRow.from_result(typer.reduce(genome, hits_to_alignments(db, genome, hm)))).split(b"\t")
        assert fk == fm, (size, kind, genome.id, [c for c in range(len(fk)) if fk[c] != fm[c]])
        sm = {_span(h): h for h in hm}
        for j in odb.joins(packed):
            for k in range(1, int(j["n_pieces"])):
                if j["piece"][k][0] != 1:
                    continue
                g, glen = int(j["gs"]) >> 1, int(db.genes.lengths[int(j["gs"]) >> 1])
                qs, qe = int(j["piece"][k][3]), int(j["piece"][k][4])
                cs = int(packed.ctg_start[int(j["contig"])])
                span = (g, int(j["contig"]), -1 if j["gs"] & 1 else 1, glen - qe if j["gs"] & 1 else qs, glen - qs if j["gs"] & 1 else qe,
                        int(j["piece"][k][5]) - cs, int(j["piece"][k][6]) - cs)
                if span in sm:  # (divergent relatives of the edited gene may end a few bases apart)
                    joined += 1
                    assert int(sm[span]["n_seeds"]) == min(255, int(j["n_anchors"]))
                    # the score is the model's, except where a join's pieces sit in 16-diagonal bands (anchors on one or two
                    # diagonals: kp_piece_width) and the gene is a diverged relative whose best path strays further: seed 7300
                    # has one such, 78 % identical, at 899 against 900
                    off_by = abs(int(sm[span]["score"]) - int(j["piece"][k][9]))
                    assert off_by == 0 or (int(j["width"]) == 16 and off_by <= 2), (size, kind, span, off_by)
                    narrow_off += off_by != 0
    assert joined >= 10 and narrow_off <= 1


def test_events_near_gene_ends_interleaved_events_and_storms_give_the_models_rows(dbs):
    """kp-align v5 (kp_spec.h, CHAINS OF ANCHORS): what round 5's review found outside the sweep -- an insertion / deletion 40,
    70 or 100 bases from a gene's start or end (the stretch beyond it is too short to be a chain of its own: minimap2 chains
    its few anchors, cuts the end off again when it is shorter than twice the gap (mm_fix_bad_ends) and lets the extension
    decide), two events that nearly cancel (the stretches before and after them share a diagonal), storms of events anywhere
    in the locus, and the non-iid background -- gives the rows of the minimap2 model: locus, type, confidence and problems
    always, the whole row in all but a few of these assemblies (tools/concordance.py runs every seed of every class through the
    reference's own Serotyper: profiles/concordance_r6.md)."""
    from tools.concordance import cases

    db, odb, typer = _db(dbs, "kpsc_k", 100)
    wanted = ("endindel_", "interleave_", "storm_", "paralog")
    picked = {}
    for tag, type_with, gen_db, kw, also in cases(1.0, 0):
        if tag.startswith(wanted) and tag not in picked and type_with == ["k"]:
            if tag.startswith("endindel_") and not (tag.startswith(("endindel_40_", "endindel_100_")) and tag.endswith(("del45", "ins200"))):
                continue  # (a third of the 24 end classes here; all of them, every seed, in tools/concordance.py)
            picked[tag] = kw
    assert len(picked) == 8 + 3 + 2 + 1
    same = 0
    for tag, kw in sorted(picked.items()):
        genome = make_assembly(db, **kw)
        packed = genome.packed()
        hk, hm = odb.align(packed), mm2.Mm2Index.from_contigs(genome.contigs).map(db.genes)
        rk = typer.reduce(genome, hits_to_alignments(db, genome, hk))
        rm = typer.reduce(genome, hits_to_alignments(db, genome, hm))
        for field in ("best_locus_name", "phenotype", "typeable"):
            assert getattr(rk, field) == getattr(rm, field), (tag, field)
        fk, fm = bytes(KaptiveRow.from_result(rk)).split(b"\t"), bytes(KaptiveRow.from_result(rm)).split(b"\t")
        assert fk[7] == fm[7], (tag, "Problems", fk[7], fm[7])
        same += fk == fm
    assert same >= len(picked) - 1, same  # (measured: all 14)


def test_occurrence_cut_follows_the_model_at_its_floor(dbs):
    """minimap2's occurrence cut (-f 2e-4, min_mid_occ 10) on a 5 Mbp assembly sits at its floor of 10 unless the assembly is
    full of repeats; kp_spec.h's KP_MID_OCC restates the floor.  A 220-base stretch of a gene planted 12 more times: the
    copies' seeds are dropped by both aligners -- same raw hits, same row -- while 9 more copies (10 occurrences) are all
    reported by both."""
    db, odb, typer = _db(dbs, "kpsc_k", 100)
    for copies, cut in ((12, True), (9, False)):
        genome = make_assembly(db, seed=4100, p_is=0, p_stop=0, repeat_segment=(220, copies))
        packed = genome.packed()
        index = mm2.Mm2Index.from_contigs(genome.contigs)
        assert index.mid_occ == 10
        hk, hm = odb.align(packed), index.map(db.genes)
        sk, sm = {_span(h) for h in hk}, {_span(h) for h in hm}
        assert len(sk ^ sm) <= 4, (copies, len(sk - sm), len(sm - sk))
        plain = len(odb.align(make_assembly(db, seed=4100, p_is=0, p_stop=0).packed()))
        assert (len(hk) < plain + 20) if cut else (len(hk) > plain + 500), (copies, len(hk), plain)  # (every family relative hits every copy)
        if cut:
            fk = bytes(KaptiveRow.from_result(typer.reduce(genome, hits_to_alignments(db, genome, hk))))
            fm = bytes(KaptiveRow.from_result(typer.reduce(genome, hits_to_alignments(db, genome, hm))))
            assert fk == fm


def test_occurrence_cut_follows_the_models_quantile_when_repeats_lift_it(dbs):
    """Round 6 (kp_spec.h, OCCURRENCE CUT): an IS-like element in 40 copies lifts minimap2's mid_occ -- the 2e-4 quantile of the
    assembly's minimizer counts -- far above its floor; a 220-base stretch of a gene in 12 more copies is then NOT cut.  The
    oracle works the assembly's quantile out where a gene has three seeds beyond the floor: same raw hits as the model (round
    5's floor-only rule dropped every copy's seeds: 0 of 3 such rows, profiles/concordance_r5.md)."""
    db, odb, typer = _db(dbs, "kpsc_k", 100)
    genome = make_assembly(db, seed=91_100, p_is=0, p_stop=0, is_copies=(40, 1200), repeat_segment=(220, 12))
    packed = genome.packed()
    index = mm2.Mm2Index.from_contigs(genome.contigs)
    assert index.mid_occ > 20
    hk, hm = odb.align(packed), index.map(db.genes)
    sk, sm = {_span(h) for h in hk}, {_span(h) for h in hm}
    assert len(sk ^ sm) <= 4, (len(sk - sm), len(sm - sk))
    plain = len(odb.align(make_assembly(db, seed=91_100, p_is=0, p_stop=0, is_copies=(40, 1200)).packed()))
    assert len(hk) > plain + 200, (len(hk), plain)  # the copies are reported (measured: 1659 hits against 1220 without them)
    rk = typer.reduce(genome, hits_to_alignments(db, genome, hk))
    rm = typer.reduce(genome, hits_to_alignments(db, genome, hm))
    for field in ("best_locus_name", "phenotype", "typeable"):
        assert getattr(rk, field) == getattr(rm, field), field


# ---- seeds: the oracle's state machine against the model's mm_sketch -----------------------------------------------------------
_CODE = np.full(256, 4, np.uint8)
for _i, _c in enumerate(b"ACGT"):
    _CODE[_c] = _CODE[_c + 32] = _i


def test_oracle_seeds_equal_the_models_sketch():
    rng = np.random.default_rng(11)
    total = 0
    for it in range(1200):
        n = int(rng.integers(1, 400))
        kind = it % 6
        if kind == 0:
            s = rng.choice(list(b"ACGT"), n)
        elif kind == 1:
            s = rng.choice(list(b"ACGTN"), n, p=[0.24, 0.24, 0.24, 0.24, 0.04])
        elif kind == 2:
            s = rng.choice(list(b"AC"), n, p=[0.9, 0.1])  # low complexity: equal 15-mers inside a window
        elif kind == 3:
            unit = rng.choice(list(b"ACGT"), int(rng.integers(1, 12)))  # periodic: ties everywhere
            s = np.tile(unit, n // len(unit) + 1)[:n]
        elif kind == 4:
            s = rng.choice(list(b"ACGT"), n)
            for _ in range(3):
                a = int(rng.integers(0, n))
                s[a : a + int(rng.integers(1, 30))] = ord("N")
        else:
            s = rng.choice(list(b"ACGT"), n)
            a = int(rng.integers(0, n))
            s[a : a + 40] = ord("A")
            if rng.random() < 0.5:
                s[rng.integers(0, n)] = ord("R")
        seq = bytes(bytearray(int(v) for v in s))
        start, z, x = O.seeds(_CODE[np.frombuffer(seq, np.uint8)])
        mx, my = mm2.sketch(seq)
        mine = sorted(zip(start.tolist(), z.tolist(), x.tolist()))
        model = sorted(zip((((my & 0xFFFFFFFF) >> 1).astype(np.int64) - 14).tolist(), (my & 1).tolist(), (mx >> 8).tolist()))
        assert mine == [(int(a), int(b), int(c)) for a, b, c in model], seq[:80]
        total += len(mine)
    assert total > 30_000
