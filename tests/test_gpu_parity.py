"""HIP path vs the CPU oracle, through the C ABI, on the same seeded inputs (run with -m gpu on an MI355X).

Integer work is compared bit for bit, stage by stage: protein DP, anchors, band tasks, hit tables, and finally whole
typing results against the golden vectors recorded from the reference.
"""

import numpy as np
import pytest

from kaptive_amd import _native
from kaptive_amd.core.genome import GenomeAssembly
from kaptive_amd.core.seq import SeqRecord, Sequences
from kaptive_amd.pack import pack_contigs, pack_sequences_flat
from kaptive_amd.serotyping.core import Serotyper
from kaptive_amd.synth import make_assembly, make_db, random_dna
from tests.golden_util import case_names, load_case, load_db
from tests.test_host_golden import check_result_against_golden

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    c = _native.Context(0)
    yield c
    c.close()


def _same_records(a: np.ndarray, b: np.ndarray, what: str):
    assert len(a) == len(b), f"{what}: {len(a)} vs {len(b)}"
    for f in a.dtype.names:
        if f == "pad":
            continue
        bad = np.flatnonzero(a[f] != b[f])
        assert len(bad) == 0, f"{what}: field {f} differs at {bad[:5]}: {a[bad[:3]]} vs {b[bad[:3]]}"


# ---- protein DP ------------------------------------------------------------------------------------------------------
def test_protein_dp_golden(ctx, golden_dir):
    z = np.load(golden_dir / "protein_dp.npz")
    got = ctx.protein_align(z["q_seqs"], z["q_offsets"], z["q_lengths"], z["t_seqs"], z["t_offsets"], z["t_lengths"])
    for i, col in enumerate(("scores", "matches", "mismatches", "gaps", "q_starts", "q_ends", "t_starts", "t_ends")):
        assert np.array_equal(got[:, i], z[col]), col


def test_protein_dp_random_vs_oracle(ctx, oracle):
    rng = np.random.default_rng(5)
    aa = np.frombuffer(b"ARNDCQEGHILKMFPSTWYVBZX*JUOa-\x00\xff", np.uint8)  # (the last seven: J, and bytes outside the alphabet)
    qs, ts = [], []
    for i in range(300):
        n = int(rng.integers(0, 700)) if i % 7 else int(rng.integers(0, 8))
        t = aa[rng.integers(0, 20, size=n)]
        q = t.copy()
        hit = rng.random(n) < rng.uniform(0, 0.5)
        q[hit] = aa[rng.integers(0, len(aa), size=int(hit.sum()))]
        for s in np.flatnonzero(rng.random(len(q)) < 0.01)[::-1]:
            q = np.delete(q, slice(s, s + 3)) if rng.random() < 0.5 else np.insert(q, s, aa[rng.integers(0, 20, size=4)])
        if i % 5 == 0:
            q = q[: int(rng.integers(0, len(q) + 1))]  # truncated: band must widen far beyond 20
        if i % 11 == 0:
            q = q[int(rng.integers(0, len(q) + 1)) :]
        qs.append(q.tobytes())
        ts.append(t.tobytes() + b"*")
    q, t = Sequences.from_bytes(qs), Sequences.from_bytes(ts)
    want = oracle.protein_align(q.seqs, q.offsets, q.lengths, t.seqs, t.offsets, t.lengths)
    got = ctx.protein_align(q.seqs, q.offsets, q.lengths, t.seqs, t.offsets, t.lengths)
    bad = np.flatnonzero((want != got).any(axis=1))
    assert len(bad) == 0, (bad[:5], want[bad[:3]], got[bad[:3]], q.lengths[bad[:3]], t.lengths[bad[:3]])


def test_protein_dp_long_and_wide_vs_oracle(ctx, oracle):
    """Shapes of the row-strip kernel the random test does not reach: windows wider than the LDS row buffer (flat scratch
    path), windows wider than the staged target residues, proteins longer than the register kernel takes with a narrow
    band (many strips, ring buffer), a band far wider than either protein."""
    rng = np.random.default_rng(17)
    aa = np.frombuffer(b"ARNDCQEGHILKMFPSTWYV", np.uint8)
    qs, ts = [], []
    for len_t, len_q, sub in ((1500, 300, 0.1), (3000, 100, 0.05), (2500, 2480, 0.2), (2300, 2300, 0.3), (1000, 20, 0.0),
                              (4200, 3900, 0.15), (5000, 4500, 0.1), (64, 1200, 0.1), (65, 129, 0.0), (2049, 2049, 0.02), (1100, 990, 0.4)):
        t = aa[rng.integers(0, 20, size=len_t)]
        start = int(rng.integers(0, max(1, len_t - len_q))) if len_q < len_t else 0
        q = np.resize(t[start:], len_q).copy() if len_q > len_t else t[start : start + len_q].copy()
        hit = rng.random(len(q)) < sub
        q[hit] = aa[rng.integers(0, 20, size=int(hit.sum()))]
        for s in np.flatnonzero(rng.random(len(q)) < 0.004)[::-1]:
            q = np.delete(q, slice(s, s + 2)) if rng.random() < 0.5 else np.insert(q, s, aa[rng.integers(0, 20, size=3)])
        qs.append(q.tobytes())
        ts.append(t.tobytes() + b"*")
    q, t = Sequences.from_bytes(qs), Sequences.from_bytes(ts)
    want = oracle.protein_align(q.seqs, q.offsets, q.lengths, t.seqs, t.offsets, t.lengths)
    got = ctx.protein_align(q.seqs, q.offsets, q.lengths, t.seqs, t.offsets, t.lengths)
    bad = np.flatnonzero((want != got).any(axis=1))
    assert len(bad) == 0, (bad[:5], want[bad[:3]], got[bad[:3]], q.lengths[bad[:3]], t.lengths[bad[:3]])
    assert (want[:, 0] > 0).sum() >= 8


def test_protein_fragments_in_one_wave_vs_oracle(ctx, oracle):
    """The two whole-wave forms of kp_protein_wide_kernel: bands of 65..512 diagonals in a wave's registers (four or eight
    diagonals per lane) and, beyond that, the whole matrix with two, four or six rows per lane -- fragments of every
    length class from anywhere in a 400..1900-residue protein, diverged, with indels, and low-complexity targets whose
    repeats tie the best score in several cells (the rows-per-lane form does not visit its cells in row-major order)."""
    rng = np.random.default_rng(41)
    aa = np.frombuffer(b"ARNDCQEGHILKMFPSTWYV", np.uint8)
    qs, ts = [], []
    for len_t in (400, 601, 900, 1300, 1900):
        t = aa[rng.integers(0, 20, size=len_t)]
        for len_q in (1, 7, 63, 64, 65, 127, 128, 129, 200, 256, 257, 300, 383, 384, 385):
            if len_q >= len_t - 22:
                continue
            start = int(rng.integers(0, len_t - len_q))
            q = t[start : start + len_q].copy()
            hit = rng.random(len_q) < rng.uniform(0.0, 0.3)
            q[hit] = aa[rng.integers(0, 20, size=int(hit.sum()))]
            for s_ in np.flatnonzero(rng.random(len(q)) < 0.01)[::-1]:
                q = np.delete(q, slice(s_, s_ + 2)) if rng.random() < 0.5 else np.insert(q, s_, aa[rng.integers(0, 20, size=3)])
            qs.append(q.tobytes()); ts.append(t.tobytes() + b"*")
    for len_t, len_q in ((300, 260), (300, 200), (300, 100), (450, 400), (450, 330), (450, 250), (600, 560), (600, 500),
                         (600, 400), (767, 700), (767, 600), (768, 520), (500, 380), (640, 385)):  # bands of 65..512 diagonals
        t = aa[rng.integers(0, 20, size=len_t)]
        start = int(rng.integers(0, len_t - len_q))
        q = t[start : start + len_q].copy()
        hit = rng.random(len_q) < 0.15
        q[hit] = aa[rng.integers(0, 20, size=int(hit.sum()))]
        qs.append(q.tobytes()); ts.append(t.tobytes() + b"*")
    for unit, n_q, n_t in ((b"GS", 90, 700), (b"A", 150, 500), (b"MKL", 60, 300), (b"PT", 190, 420), (b"Q", 300, 1000)):
        qs.append(unit * n_q); ts.append(b"W" + unit * n_t + b"*")   # every placement of the query scores the same
        qs.append(unit * n_q + b"W"); ts.append(unit * n_t + b"*")
    q, t = Sequences.from_bytes(qs), Sequences.from_bytes(ts)
    d = np.abs(q.lengths.astype(np.int64) - t.lengths.astype(np.int64))
    assert (d > 255).sum() > 30 and ((d > 31) & (d <= 255)).sum() > 10  # both forms are exercised
    want = oracle.protein_align(q.seqs, q.offsets, q.lengths, t.seqs, t.offsets, t.lengths)
    got = ctx.protein_align(q.seqs, q.offsets, q.lengths, t.seqs, t.offsets, t.lengths)
    bad = np.flatnonzero((want != got).any(axis=1))
    assert len(bad) == 0, (bad[:5], want[bad[:3]], got[bad[:3]], [qs[i][:20] for i in bad[:3]], q.lengths[bad[:3]], t.lengths[bad[:3]])


def test_protein_exact_prefix_shortcut_vs_oracle(ctx, oracle):
    """Pairs the kernels answer without DP (query = standard residues only and a prefix of the target) and their near
    misses, in both kernels' territory: identical full-length proteins, truncated ones (wide bands), very long ones,
    repeats where the whole query also fits further right (tie broken by the row-major scan), one substitution, an X or
    a B in an otherwise identical query, lower case, a query longer than its target."""
    rng = np.random.default_rng(23)
    aa = np.frombuffer(b"ARNDCQEGHILKMFPSTWYV", np.uint8)
    qs, ts = [], []
    for n in (1, 2, 19, 20, 21, 64, 330, 767, 768, 769, 1500, 4000):
        t = aa[rng.integers(0, 20, size=n)].tobytes()
        qs.append(t); ts.append(t + b"*")                      # identical
        for cut in {1, n // 3, n - 1} - {0}:
            qs.append(t[:cut]); ts.append(t + b"*")            # exact prefix (premature stop / contig edge)
        if n > 4:
            m = bytearray(t); m[n // 2] = ord("W") if m[n // 2] != ord("W") else ord("A")
            qs.append(bytes(m)); ts.append(t + b"*")           # one substitution
            x = bytearray(t); x[n // 4] = ord("X")
            qs.append(bytes(x)); ts.append(bytes(x) + b"*")    # identical but with an X: the DP decides
            b = bytearray(t); b[n // 4] = ord("B")
            qs.append(bytes(b)); ts.append(bytes(b) + b"*")
            qs.append(t.lower()); ts.append(t + b"*")
            qs.append(t + b"AC"); ts.append(t)                 # query longer than target
    for unit, reps in ((b"A", 40), (b"MK", 30), (b"GSG", 200)):
        qs.append(unit * (reps // 2)); ts.append(unit * reps + b"*")  # the query also matches further right
    q, t = Sequences.from_bytes(qs), Sequences.from_bytes(ts)
    want = oracle.protein_align(q.seqs, q.offsets, q.lengths, t.seqs, t.offsets, t.lengths)
    got = ctx.protein_align(q.seqs, q.offsets, q.lengths, t.seqs, t.offsets, t.lengths)
    bad = np.flatnonzero((want != got).any(axis=1))
    assert len(bad) == 0, (bad[:5], want[bad[:3]], got[bad[:3]], [qs[i][:30] for i in bad[:3]], q.lengths[bad[:3]], t.lengths[bad[:3]])
    # the shortcut's closed form on the pairs it applies to (first of every length group): all matches, no gaps
    full = [i for i in range(len(qs)) if ts[i] == qs[i] + b"*" and qs[i].isupper() and not set(qs[i]) & set(b"XB")]
    assert len(full) >= 12 and all(got[i][1] == len(qs[i]) and got[i][2] == got[i][3] == 0 and got[i][5] == len(qs[i]) for i in full)


# ---- aligner stages --------------------------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def small_db():
    return make_db("kpsc_k", seed=7, n_loci=9)


@pytest.fixture(scope="module")
def small_setup(ctx, oracle, small_db):
    codes, off = pack_sequences_flat(small_db.genes)
    ctx.load_genes(codes, off)
    return oracle.OracleDB(codes, off)


def _edge_case_assemblies(db):
    rng = np.random.default_rng(77)
    gene = db.genes[3].seq
    out = []
    # contigs shorter than a k-mer, a gene cut by a contig end, N runs inside and next to a gene, lower case, IUPAC
    recs = [SeqRecord("tiny", b"ACGTACG"), SeqRecord("empty_like", b"A"),
            SeqRecord("half_gene", random_dna(rng, 300, 0.5).tobytes() + gene[: len(gene) // 2]),
            SeqRecord("other_half", gene[len(gene) // 2 :] + random_dna(rng, 41, 0.5).tobytes()),
            SeqRecord("n_inside", random_dna(rng, 77, 0.5).tobytes() + gene[:200] + b"N" * 25 + gene[225:].lower()
                      + b"RYKM" + random_dna(rng, 500, 0.5).tobytes()),
            SeqRecord("exact32", random_dna(rng, 32, 0.5).tobytes()),
            SeqRecord("all_n", b"N" * 100)]  # fmt: skip
    out.append(GenomeAssembly("edge_contigs", Sequences.from_records(recs)))
    out.append(GenomeAssembly("one_base", Sequences.from_records([SeqRecord("c", b"G")])))
    # tandem copies of one gene on both strands, so several bands of one gene land on one contig
    from kaptive_amd.synth import revcomp

    g = np.frombuffer(gene, np.uint8)
    tandem = np.concatenate([random_dna(rng, 100, 0.5), g, random_dna(rng, 37, 0.5), revcomp(g), g[:400],
                             random_dna(rng, 900, 0.5), g[300:]])  # fmt: skip
    out.append(GenomeAssembly("tandem", Sequences.from_records([SeqRecord("t", tandem.tobytes())])))
    return out


def _assemblies(db):
    small = dict(length=90_000, median_contigs=5, min_contig=200)
    asms = [make_assembly(db, seed=s, **small) for s in (11, 13, 15)]
    asms.append(make_assembly(db, seed=19, n_run=40, **small))
    asms.append(make_assembly(db, seed=20, length=90_000, median_contigs=60, min_contig=200, force_split=True))
    asms.append(make_assembly(db, seed=14, locus=-1, **small))
    return asms + _edge_case_assemblies(db)


def test_anchor_task_hit_stages_match_oracle(ctx, small_setup, small_db):
    odb = small_setup
    asms = _assemblies(small_db)
    packed = [a.packed() for a in asms]
    batch = ctx.batch(packed)
    hits, off = batch.align()
    stats = batch.stats()
    total_anchors = 0
    for i, pa in enumerate(packed):
        want = odb.anchors(pa)
        got = batch.anchors(i)
        total_anchors += len(want)
        assert np.array_equal(got, want), f"anchors of {asms[i].id}: {len(got)} vs {len(want)}"
        want_t = np.sort(odb.tasks(pa), order=list(_native.TASK_DTYPE.names))
        got_t = np.sort(batch.tasks(i), order=list(_native.TASK_DTYPE.names))
        _same_records(got_t, want_t, f"tasks of {asms[i].id}")
        _same_records(hits[off[i] : off[i + 1]], odb.align(pa), f"hits of {asms[i].id}")
    assert stats["anchors"] == total_anchors and stats["hits"] == len(hits) and stats["retries"] <= 1
    assert len(hits) > 200  # the comparison above was not vacuous
    batch.close()


def _join_assemblies(db):
    """Genes with insertions and deletions of 33-480 bases (kp_spec.h, kp-align v4, v5): planted by the generator in whole
    assemblies, and hand-made contigs for what it does not reach -- three pieces in one gene, a short piece before a long
    gap (the continuation starts below zero), both strands, an N run next to the junction, a junction at a contig end,
    two copies of the edited gene on one contig, and a tail of unrelated sequence before a chance piece."""
    from kaptive_amd.synth import revcomp

    small = dict(length=90_000, median_contigs=5, min_contig=200, p_is=0, p_stop=0)
    asms = []
    for i, (size, kind) in enumerate(((33, "del"), (40, "ins"), (64, "del"), (100, "ins"), (150, "del"), (300, "ins"), (450, "del"), (480, "ins"))):
        other = "ins" if kind == "del" else "del"
        asms.append(make_assembly(db, seed=500 + i, mid_indels=((size, kind), (size + 1, kind), (size + 2, other)), **small))
    asms.append(make_assembly(db, seed=520, length=90_000, median_contigs=60, min_contig=200, force_split=True, p_is=0, p_stop=0,
                              mid_indels=((60, "del"), (90, "ins"), (200, "del"), (35, "ins"))))
    rng = np.random.default_rng(99)
    genes = [np.frombuffer(db.genes[i].seq, np.uint8) for i in (2, 5, 11, 17, 23)]
    flank = lambda n: random_dna(rng, n, 0.5)  # noqa: E731

    def edit(g, edits):  # (position, "del" | "ins", size), applied from the far end
        for at, kind, size in sorted(edits, reverse=True):
            g = np.delete(g, slice(at, at + size)) if kind == "del" else np.concatenate([g[:at], flank(size), g[at:]])
        return g

    g0, g1, g2, g3, g4 = genes
    recs = [
        SeqRecord("three_pieces", np.concatenate([flank(500), edit(g0, [(len(g0) // 3, "del", 40), (2 * len(g0) // 3, "ins", 90)]), flank(400)]).tobytes()),
        SeqRecord("three_pieces_rc", revcomp(np.concatenate([flank(300), edit(g1, [(len(g1) // 3, "ins", 70), (2 * len(g1) // 3, "del", 36)]), flank(300)])).tobytes()),
        SeqRecord("short_head", np.concatenate([flank(200), edit(g2, [(120, "del", 400)]), flank(200)]).tobytes()),
        SeqRecord("short_tail", np.concatenate([flank(200), edit(g3, [(len(g3) - 110, "ins", 450)]), flank(200)]).tobytes()),
        SeqRecord("two_copies", np.concatenate([flank(100), edit(g4, [(len(g4) // 2, "del", 80)]), flank(150), edit(g4, [(len(g4) // 2, "ins", 120)]), flank(100)]).tobytes()),
        SeqRecord("at_contig_end", np.concatenate([flank(100), edit(g0, [(len(g0) // 2, "ins", 200)])[: len(g0) // 2 + 200 + 60]]).tobytes()),
        SeqRecord("junk_tail", np.concatenate([flank(100), g1[:500], flank(260), g1[700:760], flank(100)]).tobytes()),
        SeqRecord("more_junk", np.concatenate([flank(100), g3[:500], flank(760), g3[1200:], flank(100)]).tobytes()),
    ]
    # a path that the drop test REJECTS: behind a 60-base insertion the gene goes on with 520 bases at 50 % identity (every
    # second base changed: no seeds, -520 on the later piece's diagonal, far worse on the earlier one's) before it is itself
    # again -- the local path crosses the gap and runs through them (what lies before the gap is worth more than they cost),
    # falling ~500 below the level it arrives at: minimap2's z-drop would have split the chain there, and the pieces report their
    # own hits; "not_crossed": with unrelated sequence in their place the best path does not cross and both pieces stand
    gl = max(genes, key=len)
    half = gl[420:940].copy()
    half[::2] = np.frombuffer(b"CGTA", np.uint8)[np.searchsorted(np.frombuffer(b"ACGT", np.uint8), half[::2])]
    dropped = np.concatenate([flank(150), gl[:420], flank(60), half, gl[940:], flank(150)])
    recs.append(SeqRecord("not_crossed", np.concatenate([flank(150), gl[:300], flank(60), flank(560), gl[860:], flank(150)]).tobytes()))
    recs.append(SeqRecord("drop_rejected", dropped.tobytes()))
    with_n = np.concatenate([flank(200), edit(g2, [(len(g2) // 2, "del", 90)]), flank(200)])
    with_n[200 + len(g2) // 2 - 30 : 200 + len(g2) // 2 - 22] = ord("N")
    recs.append(SeqRecord("n_near_junction", with_n.tobytes()))
    asms.append(GenomeAssembly("hand_made_joins", Sequences.from_records(recs)))
    return asms


def test_joins_across_mid_size_indels_match_oracle(ctx, small_setup, small_db):
    """kp-align v5: groups (weak clusters included), minimap2's chaining DP over a group's anchors, pieces, the joined fill
    (every piece local, cross gaps by atomic maxima), the walk-back from the best cell with the drop test and the consumed
    pieces -- join records, band tasks and hit tables equal the oracle's."""
    odb = small_setup
    asms = _join_assemblies(small_db)
    packed = [a.packed() for a in asms]
    batch = ctx.batch(packed)
    hits, off = batch.align()
    n_joined = n_rejected = n_three = 0
    for i, pa in enumerate(packed):
        want_j, got_j = odb.joins(pa), batch.joins(i)
        key = lambda j: np.lexsort((j["lo"][:, 0], j["contig"], j["gs"]))  # noqa: E731
        want_j, got_j = want_j[key(want_j)] if len(want_j) else want_j, got_j[key(got_j)] if len(got_j) else got_j
        assert len(want_j) == len(got_j), (asms[i].id, len(want_j), len(got_j))
        for f in want_j.dtype.names:
            assert np.array_equal(want_j[f], got_j[f]), (asms[i].id, f, want_j[f][:2], got_j[f][:2])
        n_joined += int((want_j["piece"][:, :, 0] == 1).sum())
        n_rejected += int((want_j["piece"][:, :, 0] == 2).sum())
        n_three += int((want_j["n_pieces"] >= 3).sum())
        want_t = np.sort(odb.tasks(pa), order=list(_native.TASK_DTYPE.names))
        got_t = np.sort(batch.tasks(i), order=list(_native.TASK_DTYPE.names))
        _same_records(got_t, want_t, f"tasks of {asms[i].id}")
        _same_records(hits[off[i] : off[i + 1]], odb.align(pa), f"hits of {asms[i].id}")
    assert n_joined >= 25 and n_three >= 2 and n_rejected >= 1, (n_joined, n_rejected, n_three)  # the comparison above was not vacuous
    batch.close()


def test_occurrence_cut_matches_oracle(oracle):
    """minimap2 drops a query seed that occurs more than mid_occ (>= 10) times among the target's minimizers
    (src/kaptive/core/genome.py:177-191 builds that index per assembly); kp_spec.h's KP_MID_OCC restates the floor: a gene
    seed with more than ten anchors in an assembly loses them all (kp_occ_cut_kernel).  A stretch of a gene repeated 9, 10,
    11, 12, 30 and 150 times on both strands, a whole gene in 14 copies, a repeat that lies beyond position 4096 of a 9 kb
    gene and one that straddles it (the second / both counting windows), next to assemblies without any repeat: anchors
    (after the cut), band tasks and hits equal the oracle's, and the cut removed what it should."""
    from kaptive_amd.synth import revcomp

    rng = np.random.default_rng(4242)
    db = make_db("kpsc_k", seed=7, n_loci=4)
    long_gene = random_dna(rng, 9_000, 0.5)
    genes = Sequences.from_records([*[SeqRecord(f"g{i}", db.genes[i].seq) for i in range(len(db.genes))], SeqRecord("long", long_gene.tobytes())])
    codes, off = pack_sequences_flat(genes)
    odb = oracle.OracleDB(codes, off)
    c = _native.Context(0)
    c.load_genes(codes, off)
    pad = lambda n: random_dna(rng, n, 0.5)  # noqa: E731
    g3, g7 = np.frombuffer(db.genes[3].seq, np.uint8), np.frombuffer(db.genes[7].seq, np.uint8)

    def repeats(unit, copies):
        parts = []
        for i in range(copies):
            u = unit.copy()
            u[rng.integers(0, len(u), size=max(1, len(u) // 300))] = ord("A")  # a little divergence: not every seed survives
            parts += [pad(int(rng.integers(40, 400))), u if i % 2 else revcomp(u)]
        return np.concatenate(parts + [pad(100)])

    # (round 6: the cut is minimap2's mid_occ of the assembly -- max(10, the 2e-4 quantile of its minimizers' occurrence counts) --,
    # worked out by the device for the assemblies that hold a gene seed beyond the floor.  It sits AT the floor only when fewer than
    # 2 in 10 000 of an assembly's distinct minimizers are repeats: every assembly gets 2.4 Mbp of unrelated sequence for that;
    # "tiny_x30" has none and its quantile is its most frequent minimizer: nothing is cut there, as in minimap2.)
    big = lambda: SeqRecord("background", pad(2_400_000).tobytes())  # noqa: E731
    asms = []
    for copies in (9, 10, 11, 12, 30, 150):
        recs = [SeqRecord("gene", np.concatenate([pad(500), g3, pad(300)]).tobytes()),
                SeqRecord("copies", repeats(g3[200:420], copies - 1).tobytes()),
                SeqRecord("other", np.concatenate([pad(200), g7, pad(100)]).tobytes()), big()]
        asms.append(GenomeAssembly(f"repeat_x{copies}", Sequences.from_records(recs)))
    # (700 bases of a gene in 14 copies are ~130 repeated minimizers: safely below 2e-4 of the distinct ones on 8 Mbp)
    asms.append(GenomeAssembly("whole_gene_x14", Sequences.from_records([SeqRecord("c", repeats(g7[:700], 14).tobytes()), big(), big(), big(), SeqRecord("more", pad(900_000).tobytes())])))
    asms.append(GenomeAssembly("long_gene_windows", Sequences.from_records([
        SeqRecord("gene", np.concatenate([pad(100), long_gene, pad(100)]).tobytes()),
        SeqRecord("second_window", repeats(long_gene[5000:5300], 15).tobytes()),
        SeqRecord("straddles", repeats(long_gene[3950:4250], 13).tobytes()), big()])))  # fmt: skip
    asms += [make_assembly(db, seed=s, length=60_000, median_contigs=4, min_contig=200) for s in (3, 4)]
    asms.append(GenomeAssembly("tiny_x30", Sequences.from_records([SeqRecord("gene", np.concatenate([pad(500), g3, pad(300)]).tobytes()),
                                                                   SeqRecord("copies", repeats(g3[200:420], 29).tobytes())])))
    # an IS-like element in 40 copies lifts the quantile above the floor (to ~40): a gene stretch in 12 copies is then NOT cut
    element = pad(1200)
    lifted = [SeqRecord("gene", np.concatenate([pad(500), g3, pad(300)]).tobytes()), SeqRecord("copies", repeats(g3[200:420], 11).tobytes()),
              SeqRecord("is", np.concatenate([np.concatenate([pad(int(rng.integers(500, 3000))), element]) for _ in range(40)]).tobytes()),
              SeqRecord("background", pad(900_000).tobytes())]
    asms.append(GenomeAssembly("quantile_lifted", Sequences.from_records(lifted)))
    packed = [a.packed() for a in asms]
    batch = c.batch(packed)
    hits, hoff = batch.align()
    n_anchors = []
    for i, pa in enumerate(packed):
        want = odb.anchors(pa)
        assert np.array_equal(batch.anchors(i), want), asms[i].id
        n_anchors.append(len(want))
        _same_records(np.sort(batch.tasks(i), order=list(_native.TASK_DTYPE.names)),
                      np.sort(odb.tasks(pa), order=list(_native.TASK_DTYPE.names)), f"tasks of {asms[i].id}")
        _same_records(hits[hoff[i] : hoff[i + 1]], odb.align(pa), f"hits of {asms[i].id}")
    # up to ten occurrences every copy is a hit; from eleven on the seeds that all copies share are gone (a copy that lost a
    # seed to its own divergence keeps the others' count at ten or below for that seed), and at thirty nothing is left
    per_asm = [int(((hits[hoff[i] : hoff[i + 1]]["gene"] == 3) & (hits[hoff[i] : hoff[i + 1]]["contig"] == 1)).sum()) for i in range(6)]
    assert per_asm[0] == 8 and per_asm[1] == 9 and per_asm[4:] == [0, 0], per_asm
    assert n_anchors[1] > n_anchors[2] > n_anchors[3] > n_anchors[4] and hoff[7] - hoff[6] == 0  # (fourteen copies of a gene's first 700 bases: every seed they share is cut)
    tiny, lifted_i = len(asms) - 2, len(asms) - 1
    assert int((hits[hoff[tiny] : hoff[tiny + 1]]["contig"] == 1).sum()) >= 25  # thirty copies, none cut: the quantile of a small index is its maximum
    assert int(((hits[hoff[lifted_i] : hoff[lifted_i + 1]]["gene"] == 3) & (hits[hoff[lifted_i] : hoff[lifted_i + 1]]["contig"] == 1)).sum()) >= 9
    batch.close()
    c.close()


def test_paralog_background_matches_oracle(oracle):
    """A background that is not iid (kaptive_amd.synth, background="paralog"): diverged relatives of database genes with
    small indels (weak chains, wide bands), two IS-like families in 5-30 copies and a 7-copy rRNA-like operon next to the
    planted K and O loci, at full size: anchors, band tasks, joins and hits of both databases equal the oracle's."""
    db, dbo = make_db("kpsc_k", seed=100), make_db("kpsc_o", seed=101)
    asms = [make_assembly(db, seed=8800 + i, also=(dbo,), background="paralog") for i in range(3)]
    packed = [a.packed() for a in asms]
    n_wide = 0
    for d in (db, dbo):
        codes, off = pack_sequences_flat(d.genes)
        odb = oracle.OracleDB(codes, off)
        c = _native.Context(0)
        c.load_genes(codes, off)
        batch = c.batch(packed)
        hits, hoff = batch.align()
        for i, pa in enumerate(packed):
            assert np.array_equal(batch.anchors(i), odb.anchors(pa)), asms[i].id
            want_t = np.sort(odb.tasks(pa), order=list(_native.TASK_DTYPE.names))
            _same_records(np.sort(batch.tasks(i), order=list(_native.TASK_DTYPE.names)), want_t, f"tasks of {asms[i].id}")
            n_wide += int((want_t["width"] > 16).sum())
            assert len(batch.joins(i)) == len(odb.joins(pa))
            _same_records(hits[hoff[i] : hoff[i + 1]], odb.align(pa), f"hits of {asms[i].id}")
        batch.close()
        c.close()
    assert n_wide > 300  # the relatives' indels put many tasks into the wider band classes


def test_twelve_thousand_genes_stay_on_the_bucket_sort(oracle):
    """A 540-locus database (about 12 400 genes: 24 800 values of the anchor key's gene/strand field, 97 KB of LDS counters)
    goes through kp_bsort.hip, not the library's radix sort: sorted anchors, tasks and hits equal the oracle's, and the
    library path (`library_sort`) gives the same bytes."""
    db = make_db("ab_k", seed=102, n_loci=540)
    assert len(db.genes) > 12_000
    odb = oracle.OracleDB(*pack_sequences_flat(db.genes))
    asms = [make_assembly(db, seed=3100 + i, length=250_000, median_contigs=40, min_contig=200, force_split=i % 2 == 0) for i in range(3)]
    packed = [a.packed() for a in asms]
    c = _native.Context(0)
    c.load_genes(*pack_sequences_flat(db.genes))
    batch = c.batch(packed)
    hits, off = batch.align()
    for i, pa in enumerate(packed):
        assert np.array_equal(batch.anchors(i), odb.anchors(pa)), asms[i].id
        _same_records(np.sort(batch.tasks(i), order=list(_native.TASK_DTYPE.names)),
                      np.sort(odb.tasks(pa), order=list(_native.TASK_DTYPE.names)), f"tasks of {asms[i].id}")
        _same_records(hits[off[i] : off[i + 1]], odb.align(pa), f"hits of {asms[i].id}")
    assert len(hits) > 500
    c.set_option("library_sort", 1)
    lib = c.batch(packed)
    hits_lib, off_lib = lib.align()
    assert np.array_equal(off, off_lib)
    _same_records(hits, hits_lib, "bucket sort vs library sort, 12 k genes")
    for b in (batch, lib):
        b.close()
    c.close()


def test_genes_with_ambiguous_bases_match_oracle(oracle, small_db):
    """N inside the GENES (code 4 on the query side): the fill kernel reads the genes as ready-made row profiles
    (KpGenes::prof) and takes "this gene holds an N" from the database's per-gene flag, and the traceback then counts
    matches base by base.  Single N, runs of N, an N in the first and in the last row, against clean and N-run assemblies."""
    codes, off = pack_sequences_flat(small_db.genes)
    codes = codes.copy()
    rng = np.random.default_rng(5)
    touched = 0
    for g in range(len(off) - 1):
        o, n = int(off[g]), int(off[g + 1] - off[g])
        if g % 3 == 2 or n < 200:
            continue  # a third of the genes stay clean: their tasks share waves with the others
        codes[o + rng.integers(0, n, size=3)] = 4
        codes[o + 100 : o + 100 + int(rng.integers(1, 9))] = 4
        if g % 2:
            codes[o] = 4
            codes[o + n - 1] = 4
        touched += 1
    assert touched > 20
    c = _native.Context(0)
    c.load_genes(codes, off)
    odb = oracle.OracleDB(codes, off)
    asms = _assemblies(small_db)[:5]
    packed = [a.packed() for a in asms]
    batch = c.batch(packed)
    hits, hoff = batch.align()
    for i, pa in enumerate(packed):
        tasks, got = batch.tasks(i), batch.task_results(i)
        want = odb.sw(pa, tasks)  # the oracle's fill + traceback of exactly these tasks
        kept = want[:, 0] >= 80
        assert np.array_equal(got[~kept][:, 0], want[~kept][:, 0]) and np.array_equal(got[kept], want[kept]), asms[i].id
        _same_records(hits[hoff[i] : hoff[i + 1]], odb.align(pa), f"hits of {asms[i].id}")
    assert len(hits) > 100
    batch.close()
    c.close()


def test_seeds_at_contig_ends_n_runs_and_repeats_match_oracle(oracle):
    """The streaming kernel decides seeds by a window rule, the edge kernel by mm_sketch's state machine next to contig ends
    and N runs (kp_seed_is_interior).  Contigs built from pieces of the genes so that every seed is an anchor, with
    everything that makes the two meet: contigs of 1 to 80 bases, N runs of every length 1..40 at every distance 0..60
    from a contig end and from each other, N at the first and last base, homopolymer and short-period stretches (equal
    15-mers inside a window: ties) in the interior, across the interior/edge boundary and at the very ends, genes cut in
    the middle of a window.  Anchors (and hits) must equal the oracle's, contig by contig."""
    rng = np.random.default_rng(20260929)
    db = make_db("kpsc_k", seed=7, n_loci=4)
    codes, off = pack_sequences_flat(db.genes)
    odb = oracle.OracleDB(codes, off)
    c = _native.Context(0)
    c.load_genes(codes, off)
    acgt = np.frombuffer(b"ACGT", np.uint8)
    gene = lambda: db.genes.seqs[int(db.genes.offsets[g := int(rng.integers(0, len(db.genes)))]):][: int(db.genes.lengths[g])]  # noqa: E731

    def piece(n):
        g = gene()
        a = int(rng.integers(0, max(1, len(g) - n)))
        return g[a : a + n].copy()

    def periodic(n):
        unit = acgt[rng.integers(0, 4, int(rng.integers(1, 9)))]
        return np.tile(unit, n // len(unit) + 1)[:n]

    asms = []
    for a in range(6):
        recs = []
        for k in range(90):
            kind = k % 9
            if kind == 0:  # short contigs, down to one base
                seq = piece(int(rng.integers(1, 81)))
            elif kind == 1:  # an N run near the start / the end / both
                seq = piece(int(rng.integers(60, 400)))
                for _ in range(int(rng.integers(1, 3))):
                    at = int(rng.integers(0, 61)) if rng.random() < 0.5 else len(seq) - 1 - int(rng.integers(0, 61))
                    at = min(max(at, 0), len(seq) - 1)
                    seq[at : at + int(rng.integers(1, 41))] = ord("N")
            elif kind == 2:  # two N runs a few bases apart in the interior
                seq = piece(int(rng.integers(200, 600)))
                at = int(rng.integers(40, len(seq) - 120))
                seq[at : at + int(rng.integers(1, 30))] = ord("N")
                at += int(rng.integers(1, 61))
                seq[at : at + int(rng.integers(1, 30))] = ord("N")
            elif kind == 3:  # ties at the very start or end
                seq = np.concatenate([periodic(int(rng.integers(16, 60))), piece(int(rng.integers(40, 200)))])
                if rng.random() < 0.5:
                    seq = seq[::-1].copy()
            elif kind == 4:  # ties in the interior and across the interior / edge boundary
                seq = piece(int(rng.integers(150, 500)))
                at = int(rng.integers(0, len(seq) - 60))
                n = int(rng.integers(16, 60))
                seq[at : at + n] = periodic(n)
            elif kind == 5:  # a whole contig of one base / one short unit
                seq = periodic(int(rng.integers(15, 200)))
            elif kind == 6:  # N at the first and the last base, IUPAC and lower case in between
                seq = piece(int(rng.integers(30, 300)))
                seq[0] = seq[-1] = ord("N")
                seq[len(seq) // 2] = ord("r")
            elif kind == 7:  # exactly the lengths around k, k + w - 1 and k + 2w
                seq = piece(int(rng.choice([14, 15, 16, 23, 24, 25, 26, 33, 34, 35, 36, 44, 45, 59, 60, 61, 73, 74, 75])))
            else:  # ordinary
                seq = piece(int(rng.integers(300, 2000)))
            recs.append(SeqRecord(f"c{k}", seq.tobytes()))
        asms.append(GenomeAssembly(f"edges{a}", Sequences.from_records(recs)))
    packed = [a.packed() for a in asms]
    batch = c.batch(packed)
    hits, hoff = batch.align()
    n_anchors = 0
    for i, pa in enumerate(packed):
        want = odb.anchors(pa)
        got = batch.anchors(i)
        n_anchors += len(want)
        assert np.array_equal(got, want), f"{asms[i].id}: {len(got)} anchors vs {len(want)}; first difference at " \
            f"{int(np.flatnonzero(got[: min(len(got), len(want))] != want[: min(len(got), len(want))])[0]) if len(got) and len(want) and (got[: min(len(got), len(want))] != want[: min(len(got), len(want))]).any() else min(len(got), len(want))}"
        _same_records(hits[hoff[i] : hoff[i + 1]], odb.align(pa), f"hits of {asms[i].id}")
    assert n_anchors > 20_000
    batch.close()
    c.close()


def test_batches_whose_tasks_are_mostly_wide_match_oracle(oracle):
    """Locus copies with an insertion or deletion every ~60 bases: nearly every cluster spans more than 16 diagonals, so the
    32- to 128-diagonal classes hold most of the batch's tasks and outlast the narrow class in the fill launch -- the blocks
    behind the narrow class's (kp_sw_kernel's helper blocks) and the one task stage per block of kp_chain_kernel do real
    work here, which they do not on the other tests' inputs.  Tasks and hits equal the oracle's."""
    db = make_db("kpsc_k", seed=100)
    odb = oracle.OracleDB(*pack_sequences_flat(db.genes))
    asms = [make_assembly(db, seed=8100 + i, length=120_000, median_contigs=4, min_contig=200, indel_rate=0.016, sub_rate=0.01 * (i % 4), p_is=0.0)
            for i in range(16)]
    packed = [a.packed() for a in asms]
    c = _native.Context(0)
    c.load_genes(*pack_sequences_flat(db.genes))
    batch = c.batch(packed)
    hits, off = batch.align()
    wide = narrow = 0
    for i in range(0, len(packed), 4):
        want_t = np.sort(odb.tasks(packed[i]), order=list(_native.TASK_DTYPE.names))
        _same_records(np.sort(batch.tasks(i), order=list(_native.TASK_DTYPE.names)), want_t, f"tasks of {asms[i].id}")
        wide += int((want_t["width"] > 16).sum())
        narrow += int((want_t["width"] == 16).sum())
    for i, pa in enumerate(packed):
        _same_records(hits[off[i] : off[i + 1]], odb.align(pa), f"hits of {asms[i].id}")
    assert wide > 2 * narrow, (wide, narrow)
    batch.close()
    c.close()


@pytest.mark.parametrize("n_loci", [800, 1500])
def test_more_than_16384_genes_sort_in_wider_buckets(oracle, n_loci):
    """More values of the gene/strand field than LDS holds counters for (32 768): a bucket then spans 2 (18 k genes) or 4
    (34 k genes) neighbouring values and is sorted on the whole key as before -- the hand-written sort takes every
    database up to KP_MAX_GENES; sorted anchors, tasks and hits equal the oracle's and the library path's."""
    db = make_db("ab_k", seed=105, n_loci=n_loci)
    assert len(db.genes) > (16_384 if n_loci == 800 else 32_768)
    codes, goff = pack_sequences_flat(db.genes)
    odb = oracle.OracleDB(codes, goff)
    asms = [make_assembly(db, seed=3300 + i, locus=(n_loci - 1) * i, length=200_000, median_contigs=30, min_contig=200, force_split=True) for i in range(2)]
    packed = [a.packed() for a in asms]
    c = _native.Context(0)
    c.load_genes(codes, goff)
    batch = c.batch(packed)
    hits, off = batch.align()
    for i, pa in enumerate(packed):
        assert np.array_equal(batch.anchors(i), odb.anchors(pa)), asms[i].id
        _same_records(hits[off[i] : off[i + 1]], odb.align(pa), f"hits of {asms[i].id}")
    assert len(hits) > 300 and hits["gene"].max() > len(db.genes) - 40  # (genes of the last locus: the top buckets)
    c.set_option("library_sort", 1)
    lib = c.batch(packed)
    hits_lib, off_lib = lib.align()
    assert np.array_equal(off, off_lib)
    _same_records(hits, hits_lib, "wide buckets vs library sort")
    for b in (batch, lib):
        b.close()
    c.close()


def test_bucket_sort_and_library_sort_give_the_same_anchors(oracle):
    """kp_bsort.hip against rocPRIM's segmented radix sort (`library_sort`) and the oracle, on an assembly built to hit
    every bucket size class of the bucket sort: single anchors, a few (sorting network in one lane), tens (wave ranking),
    hundreds (LDS-staged ranking), and thousands in one gene/strand bucket (a long gene present several times: the
    block-wide pass)."""
    from kaptive_amd.synth import mutate, revcomp

    rng = np.random.default_rng(99)
    long_gene = random_dna(rng, 6000, 0.5)
    genes = [long_gene] + [random_dna(rng, int(n), 0.5) for n in rng.integers(300, 1500, size=50)]
    db_genes = Sequences.from_records([SeqRecord(f"g{i}", g.tobytes()) for i, g in enumerate(genes)])
    codes, off = pack_sequences_flat(db_genes)
    c = _native.Context(0)
    c.load_genes(codes, off)
    odb = oracle.OracleDB(codes, off)
    parts = [random_dna(rng, 500, 0.5)]
    for _ in range(5):  # five copies of the long gene: ~7500 anchors in one bucket
        parts += [mutate(rng, long_gene, 0.002), random_dna(rng, 300, 0.5)]
    parts += [revcomp(long_gene)[:2500], random_dna(rng, 200, 0.5)]
    for i, g in enumerate(genes[1:41]):
        if i % 3 == 0:
            parts += [mutate(rng, g, 0.01 * (i % 7)), random_dna(rng, 150, 0.5)]  # full copies: hundreds of anchors
        elif i % 3 == 1:
            parts += [g[: 40 + 5 * i], random_dna(rng, 100, 0.5)]  # short pieces: a few anchors
        else:
            parts += [mutate(rng, g, 0.12), random_dna(rng, 90, 0.5)]  # diverged: tens of anchors
    for g in genes[41:]:  # 19 bases of a gene that occurs nowhere else: one or two anchors
        parts += [g[100:119], random_dna(rng, 60, 0.5)]
    asm = GenomeAssembly("buckets", Sequences.from_records([SeqRecord("c1", np.concatenate(parts).tobytes()),
                                                            SeqRecord("c2", random_dna(rng, 50_000, 0.5).tobytes())]))
    plain = make_assembly(make_db("kpsc_k", seed=7, n_loci=3), seed=5, length=40_000, median_contigs=3, locus=-1)
    packed = [asm.packed(), plain.packed(), asm.packed()]
    mine = c.batch(packed)
    hits, offs = mine.align()
    c.set_option("library_sort", 1)
    try:
        lib = c.batch(packed)
        hits_lib, offs_lib = lib.align()
    finally:
        c.set_option("library_sort", 0)
    sizes = []
    for i, pa in enumerate(packed):
        want = odb.anchors(pa)
        got = mine.anchors(i)
        assert np.array_equal(got, want), f"bucket sort, assembly {i}: {len(got)} vs {len(want)}"
        assert np.array_equal(lib.anchors(i), want), f"library sort, assembly {i}"
        if len(want):
            sizes.append(np.bincount((want >> np.uint64(46)).astype(np.int64)))
    assert np.array_equal(offs, offs_lib)
    _same_records(hits, hits_lib, "bucket sort vs library sort")
    s = sizes[0]
    assert (s == 1).any() and ((s >= 2) & (s <= 8)).any() and ((s > 8) & (s <= 64)).any() and ((s > 64) & (s <= 512)).any() \
        and (s > 4096).any(), np.unique(s)  # fmt: skip
    for b in (mine, lib):
        b.close()
    c.close()


def test_sw_raw_results_match_oracle(ctx, small_setup, small_db):
    """Every band task's DP result equals the oracle's, row for row: the ones that become hits in all seven fields, the
    ones below the score cut-off (which the device does not trace back) in their score."""
    odb = small_setup
    for seed, extra in ((12, dict(sub_rate=0.08)), (15, dict(p_is=1.0)), (21, dict(sub_rate=0.18, indel_rate=0.004))):
        asm = make_assembly(small_db, seed=seed, length=90_000, median_contigs=5, min_contig=200, **extra)
        pa = asm.packed()
        batch = ctx.batch([pa])
        hits, _ = batch.align()
        tasks = batch.tasks(0)
        got = batch.task_results(0)
        want = odb.sw(pa, tasks)  # the oracle's fill + traceback of exactly these tasks, in this order
        assert len(tasks) == len(got) == len(want) and len(tasks) > 10
        kept = want[:, 0] >= 80
        assert np.array_equal(got[~kept][:, 0], want[~kept][:, 0]), "best-cell scores of the tasks below the cut-off"
        assert np.array_equal(got[kept], want[kept]), "coordinates, matches and columns of the traced tasks"
        assert not got[~kept][:, 1:].any() and (~kept).any() == bool((want[:, 0] < 80).any())
        _same_records(hits, odb.align(pa), "hits")
        assert kept.sum() >= len(hits)
        batch.close()


def test_random_assembly_shapes_match_oracle(ctx, small_setup, small_db):
    """Fuzz over what shapes the chaining / SW / expansion kernels: divergence up to 25 %, indels, many tiny contigs,
    N runs, a second partial locus, no locus at all -- anchors, tasks and hits of one batch, all compared."""
    odb = small_setup
    rng = np.random.default_rng(2024)
    asms = []
    for i in range(28):
        kw = dict(length=int(rng.integers(30_000, 250_000)), median_contigs=int(rng.choice([1, 3, 12, 80, 400])),
                  min_contig=int(rng.choice([16, 200])), sub_rate=float(rng.choice([0.0, 0.01, 0.05, 0.12, 0.25])),
                  p_break=float(rng.choice([0.0, 0.5, 1.0])), p_is=0.3, p_stop=0.3)
        if i % 5 == 0:
            kw["n_run"] = int(rng.integers(1, 300))
        if i % 7 == 0:
            kw["second_locus"] = int(rng.integers(0, 9))
        if i % 9 == 0:
            kw["locus"] = -1
        if i % 4 == 0:
            kw["force_split"] = True
        asms.append(make_assembly(small_db, seed=5000 + i, **kw))
    packed = [a.packed() for a in asms]
    batch = ctx.batch(packed)
    hits, off = batch.align()
    n_hits = 0
    for i, pa in enumerate(packed):
        assert np.array_equal(batch.anchors(i), odb.anchors(pa)), f"anchors of fuzz assembly {i}"
        want_t = np.sort(odb.tasks(pa), order=list(_native.TASK_DTYPE.names))
        got_t = np.sort(batch.tasks(i), order=list(_native.TASK_DTYPE.names))
        _same_records(got_t, want_t, f"tasks of fuzz assembly {i}")
        want = odb.align(pa)
        _same_records(hits[off[i] : off[i + 1]], want, f"hits of fuzz assembly {i}")
        n_hits += len(want)
    assert n_hits > 500
    batch.close()


def test_overflow_retry_gives_same_hits(ctx, small_setup, small_db, monkeypatch):
    odb = small_setup
    asm = make_assembly(small_db, seed=11, length=90_000, median_contigs=5, min_contig=200)
    ctx.set_option("anchor_cap", 1024)
    ctx.set_option("tasks_per_asm", 8)
    batch = ctx.batch([asm.packed()])
    hits, _ = batch.align()
    assert batch.stats()["retries"] >= 1
    ctx.set_option("anchor_cap", 1 << 17)
    ctx.set_option("tasks_per_asm", 4096)
    _same_records(hits, odb.align(asm.packed()), "hits after overflow retry")
    batch.close()


def test_full_size_assemblies_match_oracle(ctx, oracle):
    """Config 2 shape: 5 Mbp assemblies against the 163-locus K database; whole hit tables compared."""
    db = make_db("kpsc_k", seed=100)
    codes, off = pack_sequences_flat(db.genes)
    ctx.load_genes(codes, off)
    odb = oracle.OracleDB(codes, off)
    assert ctx.n_postings == odb.n_postings
    asms = [make_assembly(db, seed=200 + i) for i in range(3)]
    packed = [a.packed() for a in asms]
    batch = ctx.batch(packed)
    hits, hoff = batch.align()
    for i, pa in enumerate(packed):
        _same_records(hits[hoff[i] : hoff[i + 1]], odb.align(pa), f"hits of {asms[i].id}")
    assert len(hits) > 3000
    batch.close()


def test_ab_shape_many_contigs_matches_oracle(ctx, oracle):
    """Config 4 shape (scaled): AT-rich database, assembly shredded into many short contigs."""
    db = make_db("ab_k", seed=102, n_loci=24)
    codes, off = pack_sequences_flat(db.genes)
    ctx.load_genes(codes, off)
    odb = oracle.OracleDB(codes, off)
    asms = [make_assembly(db, seed=300 + i, length=4.0e5, median_contigs=400, min_contig=200, force_split=True)
            for i in range(3)]  # fmt: skip
    packed = [a.packed() for a in asms]
    batch = ctx.batch(packed)
    hits, hoff = batch.align()
    for i, pa in enumerate(packed):
        _same_records(hits[hoff[i] : hoff[i + 1]], odb.align(pa), f"hits of {asms[i].id}")
    batch.close()


def test_empty_batch_and_errors(ctx, small_setup):
    batch = ctx.batch([])
    hits, off = batch.align()
    assert len(hits) == 0 and off.tolist() == [0]
    batch.close()
    with pytest.raises(ValueError):
        ctx.load_genes(np.zeros(70000, np.uint8), np.array([0, 70000], np.int32))  # gene longer than KP_MAX_GENE_LEN


# ---- whole typing results against the reference's goldens ---------------------------------------------------------------
@pytest.mark.parametrize("name", [n for n in case_names() if not n.startswith("random_hits")])
def test_typing_end_to_end_matches_reference(name):
    """FASTA-level input -> HIP aligner -> reduction -> HIP protein DP: every field and the TSV row must equal what
    the reference's Serotyper produced (its aligner stage replaced by the oracle's hit table, see make_golden.py)."""
    key, genome, hits, exp, scalars, kwargs = load_case(name)
    typer = Serotyper(load_db(key), **kwargs)
    check_result_against_golden(typer(genome), exp, scalars)  # a batch of one: reduction on the device
    check_result_against_golden(typer.call_with_host_reduction(genome), exp, scalars)
    typer.engine.close()


def test_type_many_equals_single_calls():
    db = load_db("k")
    typer = Serotyper(db)
    genomes = [load_case(n)[1] for n in ("k_plain1", "k_split", "k_nolocus", "k_is")]
    many = typer.type_many(genomes)
    for g, r in zip(genomes, many):
        single = typer(g)  # a batch of one
        _results_equal(r, typer.call_with_host_reduction(g), g.id)
        assert r.to_dict().keys() == single.to_dict().keys()
        assert r.best_locus_name == single.best_locus_name and r.phenotype == single.phenotype
        assert np.array_equal(r.gene_hits.t_starts, single.gene_hits.t_starts)
        assert r.protein_identities.tobytes() == single.protein_identities.tobytes()
    typer.engine.close()


# ---- batched device reduction ---------------------------------------------------------------------------------------------
def _results_equal(a, b, what):
    da, db_ = a.to_dict(), b.to_dict()
    for k in ("best_locus_idx", "best_locus_name", "phenotype", "typeable", "missing_expected_genes", "problems"):
        assert da[k] == db_[k], (what, k, da[k], db_[k])
    for k in ("best_locus_score", "best_locus_completeness", "length_discrepancy", "percent_identity", "percent_coverage"):
        assert np.float64(da[k]).tobytes() == np.float64(db_[k]).tobytes(), (what, k, da[k], db_[k])
    for k, v in da["gene_hits"].items():
        assert np.array_equal(np.asarray(v), np.asarray(db_["gene_hits"][k])), (what, k)
    assert np.array_equal(da["gene_states"], db_["gene_states"]), what
    assert da["protein_identities"].tobytes() == db_["protein_identities"].tobytes(), what
    for k, v in da["locus_pieces"].items():
        assert np.array_equal(v, db_["locus_pieces"][k]), (what, k)
    for grp in ("locus_seqs", "gene_seqs", "translations"):
        assert da[grp]["seqs"] == db_[grp]["seqs"] and tuple(da[grp]["ids"]) == tuple(db_[grp]["ids"]), (what, grp)


def _adversarial_hits(rng, db, genome, n, by_score=True):
    """Hit tables no aligner would emit (the recipe of the reference goldens' random_hits cases, oracle/make_golden.py):
    heavy overlaps around a few hot spots, scores / matches drawn from a handful of values, mapq 0 / 1 / 60 / 255.
    ``by_score=False``: a gene's hits in any order (minimap2 ranks by dp_max, not by the alignment score: joined hits)."""
    hits = np.zeros(n, _native.HIT_DTYPE)
    genes = np.sort(rng.integers(0, len(db.genes), size=n))
    glen = db.genes.lengths[genes]
    ctg = rng.integers(0, len(genome.contigs), size=n)
    clen = genome.contigs.lengths[ctg]
    span = np.minimum((glen * rng.uniform(0.05, 1.0, size=n)).astype(np.int64) + 1, np.minimum(glen, clen))
    hits["gene"], hits["contig"] = genes, ctg
    hits["q_start"] = (rng.random(n) * (glen - span + 1)).astype(np.int64)
    hits["q_end"] = hits["q_start"] + span
    hot = (rng.integers(0, 6, size=n) * 0.15 * clen).astype(np.int64)
    hits["t_start"] = np.clip(hot + rng.integers(-300, 300, size=n), 0, clen - span)
    hits["t_end"] = hits["t_start"] + span
    hits["strand"] = rng.choice(np.array([1, -1], np.int8), size=n)
    hits["score"] = rng.choice(np.array([80, 120, 120, 500, 500, 900, 2000]), size=n)
    hits["matches"] = rng.choice(np.array([40, 60, 60, 250, 450]), size=n)
    hits["block_len"] = span
    hits["mapq"] = rng.choice(np.array([0, 1, 60, 60, 255], np.uint8), size=n)
    return hits[np.lexsort((-hits["score"] if by_score else rng.random(n), hits["gene"]))]


def test_adversarial_hit_tables_through_the_device_reduction():
    """The reference's reduction orders the cull by (score + 1e9 * priority, matches, uint8-wrapped -mapq)
    (src/kaptive/core/alignment.py:669-675) and takes any hit table.  The golden cases random_hits0-4 -- tables with
    equal scores, equal matches and mapq 0 / 255 / 1 ties, run through the reference's own Serotyper -- and a fuzz of the
    same recipe go through kp_reduce_kernel (bitonic cull order, 64-wide greedy cull, clustering, translation) by way of
    kp_batch_set_hits; results equal the goldens field for field and the host reduction on the fuzz."""
    from kaptive_amd.engine import Engine
    from tests.golden_util import hits_to_alignments

    names = [n for n in case_names() if n.startswith("random_hits")]
    assert len(names) == 5
    for key in ("k", "o"):
        cases = [load_case(n) for n in names if load_case(n)[0] == key]
        db = load_db(key)
        eng = Engine(db)
        typer = Serotyper(db)
        typer._engine = eng
        genomes = [c[1] for c in cases]
        rng = np.random.default_rng(20260930 + len(cases))
        tables = [np.asarray(c[2]) for c in cases]
        for i in range(40):  # the fuzz: the same genomes, fresh tables of 2 to 900 hits
            g = genomes[i % len(genomes)]
            genomes.append(g)
            tables.append(_adversarial_hits(rng, db, g, int(rng.integers(2, 900 if i % 8 else 3000)), by_score=i % 2 == 0))
        batch = eng.ctx.batch([g.packed() for g in genomes])
        batch.align_async()
        batch.wait()
        off = np.concatenate([[0], np.cumsum([len(t) for t in tables])]).astype(np.int64)
        flat = np.concatenate(tables) if len(tables) else np.zeros(0, _native.HIT_DTYPE)
        batch.set_hits(flat.astype(_native.HIT_DTYPE), off)
        got_hits, got_off = batch.hits()
        assert np.array_equal(got_off, off) and got_hits.tobytes() == flat.astype(_native.HIT_DTYPE).tobytes()
        res = eng.type_batch(typer, batch, [g.id for g in genomes], genomes, aligned=True).results()
        for c, r in zip(cases, res):
            check_result_against_golden(r, c[3], c[4])
        for g, t, r in zip(genomes[len(cases):], tables[len(cases):], res[len(cases):]):
            _results_equal(r, typer.reduce(g, hits_to_alignments(db, g, t)), f"fuzz table of {len(t)} hits on {g.id}")
        batch.close()
        eng.close()


@pytest.mark.parametrize("key", ["k", "o"])
def test_batched_device_reduction_matches_golden_cases(key):
    """type_many = alignment + reduction on the device for all cases of one database in one batch; every result must
    equal the reference's golden record (fields and TSV bytes)."""
    names = [n for n in case_names() if not n.startswith("random_hits") and load_case(n)[0] == key
             and not n.startswith("k_divergent_")]  # fmt: skip
    cases = [load_case(n) for n in names]
    typer = Serotyper(load_db(key))
    results = typer.type_many([c[1] for c in cases])
    for name, case, res in zip(names, cases, results):
        check_result_against_golden(res, case[3], case[4])
    typer.engine.close()


def test_batched_device_reduction_matches_host_reduction_on_synthetic_batch(small_db):
    """Device reduction vs the golden-pinned host reduction on assemblies no fixture covers, including edge cases."""
    typer = Serotyper(small_db)
    genomes = _assemblies(small_db) + [make_assembly(small_db, seed=400 + i, length=120_000, median_contigs=8,
                                                      min_contig=200, sub_rate=0.02 * i) for i in range(8)]  # fmt: skip
    many = typer.type_many(genomes)
    for g, r in zip(genomes, many):
        _results_equal(r, typer.call_with_host_reduction(g), g.id)
    typer.engine.close()


def test_batched_reduction_confidence_switches():
    for tag in ("loose", "strict"):
        key, genome, hits, exp, scalars, kwargs = load_case(f"k_divergent_{tag}")
        typer = Serotyper(load_db(key), **kwargs)
        check_result_against_golden(typer.type_many([genome])[0], exp, scalars)
        typer.engine.close()


def test_batched_reduction_full_size():
    db = make_db("kpsc_k", seed=100)
    typer = Serotyper(db)
    genomes = [make_assembly(db, seed=200 + i) for i in range(4)]
    for g, r in zip(genomes, typer.type_many(genomes)):
        _results_equal(r, typer.call_with_host_reduction(g), g.id)
    typer.engine.close()


def test_one_alignment_pass_for_two_databases(small_db):
    """K and O genes in one context (typing groups): every assembly is scanned and aligned once, and each database's
    records, decisions and TSV rows are those of an engine that holds that database alone."""
    from kaptive_amd.engine import Engine

    db_o = make_db("kpsc_o", seed=8)
    genomes = [make_assembly(small_db, seed=900 + i, length=150_000, median_contigs=int(3 + 5 * i), min_contig=200,
                             sub_rate=0.01 * i, also=(db_o,)) for i in range(10)]  # fmt: skip
    genomes.append(make_assembly(small_db, seed=950, length=60_000, locus=-1))  # neither locus
    ids = [g.id for g in genomes]
    packed = [g.packed() for g in genomes]
    both = Engine([small_db, db_o])
    batch = both.ctx.batch(packed)
    batch.align_async()
    typers = [Serotyper(small_db), Serotyper(db_o)]
    together = [both.view(i).type_batch(t, batch, ids, aligned=True) for i, t in enumerate(typers)]
    # results of a group survive work on the other one
    sums_k_again = batch.typing(0)[0]
    assert sums_k_again.tobytes() == together[0].sums.tobytes()
    for i, (db, typer) in enumerate(zip((small_db, db_o), typers)):
        alone = Engine(db)
        b1 = alone.ctx.batch(packed)
        want = alone.type_batch(typer, b1, ids)
        got = together[i]
        assert want.sums.tobytes() == got.sums.tobytes(), f"summaries of database {i}"
        for a in range(len(ids)):  # rows beyond an assembly's own count are whatever the buffers held
            nk, npc = int(want.sums["n_kept"][a]), int(want.sums["n_pieces"][a])
            assert want.kept[a, :nk].tobytes() == got.kept[a, :nk].tobytes(), f"kept hits of database {i}, assembly {a}"
            assert want.pieces[a, :npc].tobytes() == got.pieces[a, :npc].tobytes(), f"pieces of database {i}, assembly {a}"
        assert np.array_equal(want.best_locus, got.best_locus) and want.best_score.tobytes() == got.best_score.tobytes()
        assert want.rows() == got.rows(), f"TSV rows of database {i}"
        assert any(want.typeable) or i == 1
        b1.close()
        alone.close()
    batch.close()
    both.close()


def test_type_batches_slides_a_window_over_more_batches_than_work_slots(small_db):
    """Engine.type_batches with 2 * WORK_SLOTS + 1 batches: no pass is displaced before its records are read, and every
    batch's rows equal those of the same batch typed alone; more than WORK_SLOTS batches aligned up front are refused."""
    from kaptive_amd.engine import Engine

    n = 2 * _native.WORK_SLOTS + 1
    groups = [[make_assembly(small_db, seed=1200 + 10 * b + i, length=60_000, median_contigs=4, min_contig=200,
                             sub_rate=0.01 * i) for i in range(3)] for b in range(n)]  # fmt: skip
    typer = Serotyper(small_db)
    eng = Engine(small_db)
    batches = [eng.ctx.batch([g.packed() for g in grp]) for grp in groups]
    ids = [[g.id for g in grp] for grp in groups]
    got = eng.type_batches(typer, batches, ids)
    assert len(got) == n
    for b, grp in enumerate(groups):
        one = eng.ctx.batch([g.packed() for g in grp])
        want = eng.type_batch(typer, one, ids[b])
        assert want.rows() == got[b].rows(), f"batch {b}"
        assert want.sums.tobytes() == got[b].sums.tobytes(), f"batch {b}"
        one.close()
    with pytest.raises(ValueError, match="WORK_SLOTS"):
        eng.type_batches(typer, batches, ids, aligned=True)
    with pytest.raises(ValueError, match="WORK_SLOTS"):
        eng.reduce_batches(typer, batches)
    for b in batches:
        b.close()
    eng.close()


def test_reduction_buffer_overflow_retry(small_db, monkeypatch):
    monkeypatch.setenv("KAPTIVE_AMD_KEPT_CAP", "4")
    monkeypatch.setenv("KAPTIVE_AMD_PIECE_CAP", "1")
    monkeypatch.setenv("KAPTIVE_AMD_PROT_CAP", "256")
    monkeypatch.setenv("KAPTIVE_AMD_HIT_CAP", "16")
    typer = Serotyper(small_db)
    genomes = _assemblies(small_db)[:4]
    many = typer.type_many(genomes)
    monkeypatch.undo()
    for g, r in zip(genomes, many):
        _results_equal(r, typer.call_with_host_reduction(g), g.id)
    typer.engine.close()


def test_hundreds_of_kept_hits_cluster_outside_lds(small_db):
    """More kept hits than the reduction kernel's LDS copy holds (292 records): every gene of the database planted twice,
    far apart, so that nothing overlaps and ~360 hits survive the cull -- the kept-hit buffer grows past its default 256
    and the clustering runs on the records in global memory.  Against the host statement of the reduction."""
    rng = np.random.default_rng(61)
    parts, recs = [], []
    for copy in range(2):
        for i in range(len(small_db.genes)):
            parts.append(random_dna(rng, 150, 0.5).tobytes() + small_db.genes[i].seq)
            if len(parts) == 12:
                recs.append(SeqRecord(f"c{len(recs)}", b"".join(parts) + random_dna(rng, 150, 0.5).tobytes()))
                parts = []
    if parts:
        recs.append(SeqRecord(f"c{len(recs)}", b"".join(parts) + random_dna(rng, 150, 0.5).tobytes()))
    genome = GenomeAssembly("every_gene_twice", Sequences.from_records(recs))
    typer = Serotyper(small_db)
    got = typer.type_many([genome, _assemblies(small_db)[0]])
    want = typer.call_with_host_reduction(genome)
    _results_equal(got[0], want, genome.id)
    n_hits = len(np.asarray(next(iter(want.to_dict()["gene_hits"].values()))))
    assert n_hits > 292, n_hits  # (the records did not fit the kernel's LDS copy)
    typer.engine.close()


def test_cli_types_fasta_files_like_the_reference(tmp_path):
    """`python -m kaptive_amd assembly db.npz *.fasta -o out.tsv -j out.jsonl`: rows equal the reference's golden rows."""
    from kaptive_amd import KAPTIVE_COMPAT_VERSION
    from kaptive_amd.cli import main
    from kaptive_amd.serotyping.io import KaptiveRow

    names = ["k_plain1", "k_split", "k_nolocus", "k_is", "k_nrun"]
    paths, want = [], []
    for n in names:
        key, genome, hits, exp, scalars, _ = load_case(n)
        p = tmp_path / f"{genome.id}.fasta"
        p.write_bytes(genome.contigs.to_fasta())
        paths.append(str(p))
        want.append(bytes(exp["kaptive_row"]).replace(scalars["kaptive_version"].encode(), KAPTIVE_COMPAT_VERSION.encode(), 1))
    db_path = load_db("k").save(tmp_path / "db_k.npz")
    out, js = tmp_path / "out.tsv", tmp_path / "out.jsonl"
    rc = main(["assembly", str(db_path), *paths, "-o", str(out), "-j", str(js), "-g", str(tmp_path / "genes"),
               "--batch-size", "2", "-t", "2"])  # fmt: skip
    assert rc == 0
    rows = out.read_bytes().splitlines(keepends=True)
    assert rows[0] == KaptiveRow.header() and rows[1:] == want
    assert len(js.read_bytes().splitlines()) == len(names)
    assert (tmp_path / "genes" / "k_plain1_kaptive_results.ffn").stat().st_size > 1000
    assert main(["assembly", str(tmp_path / "missing.npz"), paths[0], "-o", str(out)]) == 1


def test_cli_tsv_fast_path_equals_the_per_result_path_and_device_processes(tmp_path):
    """`kaptive assembly -o` alone streams `BatchTyping.tsv()` (no per-assembly objects, no sequence text kept); with any
    other output the rows come from SerotypingResult objects.  Both must give the same bytes; so must two device
    processes (`--devices 0,0`: two contexts on the one GPU of the box, chunks dealt round-robin, rows gathered in input
    order) and a batch size that leaves a ragged last chunk."""
    from kaptive_amd.cli import main

    db = make_db("kpsc_k", seed=7, n_loci=9)
    db_path = db.save(tmp_path / "db.npz")
    paths = []
    for i in range(11):
        g = make_assembly(db, seed=900 + i, name=f"asm{i:02d}", length=60_000 + 3_000 * i, median_contigs=3 + i % 4,
                          n_run=10 * (i % 3), locus=-1 if i == 4 else None)
        p = tmp_path / f"asm{i:02d}.fasta{'.gz' if i % 5 == 0 else ''}"
        data = g.contigs.to_fasta()
        if i % 5 == 0:
            import gzip

            data = gzip.compress(data)
        p.write_bytes(data)
        paths.append(str(p))
    fast, slow, two = tmp_path / "fast.tsv", tmp_path / "slow.tsv", tmp_path / "two.tsv"
    assert main(["assembly", str(db_path), *paths, "-o", str(fast), "--batch-size", "4", "-t", "3"]) == 0
    assert main(["assembly", str(db_path), *paths, "-o", str(slow), "-j", str(tmp_path / "r.jsonl"), "--batch-size", "3"]) == 0
    import subprocess
    import sys

    r = subprocess.run([sys.executable, "-m", "kaptive_amd", "assembly", str(db_path), *paths, "-o", str(two), "--batch-size", "2",
                        "--devices", "0,0", "-t", "2"], capture_output=True, text=True, timeout=600,
                       cwd=str(__import__("pathlib").Path(__file__).resolve().parent.parent))  # (as a user runs it: the device processes re-import __main__)
    assert r.returncode == 0, r.stderr[-2000:]
    rows = fast.read_bytes().splitlines(keepends=True)
    assert len(rows) == 1 + len(paths) and [r.split(b"\t")[3] for r in rows[1:]] == [f"asm{i:02d}".encode() for i in range(11)]
    assert fast.read_bytes() == slow.read_bytes() == two.read_bytes()
    assert len((tmp_path / "r.jsonl").read_bytes().splitlines()) == len(paths)
    assert main(["assembly", str(db_path), paths[0], str(tmp_path / "nope.fasta"), "-o", str(fast)]) == 1
    assert main(["assembly", str(db_path), *paths, "-o", str(tmp_path / "all.tsv"), "--devices", "all"]) == 0  # every GPU the box has
    assert (tmp_path / "all.tsv").read_bytes() == fast.read_bytes() and _native.device_count() >= 1
    # --pha4ge rides on the batch's columns too (BatchTyping.pha4ge); with -j beside it the rows come from result objects
    ph_fast, ph_slow = tmp_path / "fast.pha4ge", tmp_path / "slow.pha4ge"
    assert main(["assembly", str(db_path), *paths, "-o", str(tmp_path / "f2.tsv"), "--pha4ge", str(ph_fast), "--batch-size", "4"]) == 0
    assert main(["assembly", str(db_path), *paths, "-o", str(tmp_path / "s2.tsv"), "--pha4ge", str(ph_slow), "-j", str(tmp_path / "r2.jsonl"),
                 "--batch-size", "5"]) == 0  # fmt: skip
    from kaptive_amd.serotyping.io import Pha4geRow

    assert ph_fast.read_bytes() == ph_slow.read_bytes() and ph_fast.read_bytes().startswith(Pha4geRow.header())
    assert len(ph_fast.read_bytes().splitlines()) == 1 + len(paths) and (tmp_path / "f2.tsv").read_bytes() == fast.read_bytes()
    # the same pipeline for a library caller: Serotyper.tsv_from_files, a typer that goes on typing afterwards
    from kaptive_amd.serotyping.io import KaptiveRow

    typer = Serotyper(db)
    chunks = list(typer.tsv_from_files(paths, batch_size=4, threads=2))
    assert len(chunks) == 3 and KaptiveRow.header() + b"".join(chunks) == fast.read_bytes()
    again = typer.type_many(paths[:2])
    assert [bytes(KaptiveRow.from_result(r)) for r in again] == rows[1:3]
    # -j: the native JSON lines of the batch (kp_format_json) are the lines the Python serialiser writes for the result objects,
    # and `-j` with `-g` beside it (result objects for the fasta files) writes the same lines
    from kaptive_amd.cli import result_to_json

    everyone = typer.type_many(paths)
    assert (tmp_path / "r.jsonl").read_bytes() == b"".join(result_to_json(r) for r in everyone)
    assert main(["assembly", str(db_path), *paths, "-o", str(tmp_path / "s3.tsv"), "-j", str(tmp_path / "r3.jsonl"), "-g", str(tmp_path / "genes3"),
                 "--batch-size", "4"]) == 0  # fmt: skip
    assert (tmp_path / "r3.jsonl").read_bytes() == (tmp_path / "r.jsonl").read_bytes()
    # -l / -g / -p: a file per assembly and kind, their records made per batch (kp_format_fasta): the bytes of the result
    # objects' locus_seqs / gene_seqs / translations .to_fasta()
    from kaptive_amd.cli import FILE_SUFFIX

    assert main(["assembly", str(db_path), *paths, "-o", str(tmp_path / "s4.tsv"), "-l", str(tmp_path / "loci4"), "-g", str(tmp_path / "genes4"),
                 "-p", str(tmp_path / "prot4"), "--batch-size", "3"]) == 0  # fmt: skip
    assert (tmp_path / "s4.tsv").read_bytes() == (tmp_path / "s3.tsv").read_bytes()
    for r in everyone:
        for d, ext, seqs in (("loci4", "fna", r.locus_seqs), ("genes4", "ffn", r.gene_seqs), ("prot4", "faa", r.translations)):
            assert (tmp_path / d / f"{r.genome}_{FILE_SUFFIX}.{ext}").read_bytes() == seqs.to_fasta(), (r.genome, ext)
    assert (tmp_path / "genes3" / f"{everyone[0].genome}_{FILE_SUFFIX}.ffn").read_bytes() == everyone[0].gene_seqs.to_fasta()


# ---- BASELINE.json configs at their real shape --------------------------------------------------------------------------
def _oracle_typer(db, oracle):
    """Host reduction (the golden-pinned statement) fed by the oracle's aligner and protein DP: the expected rows."""
    from kaptive_amd.core.pairwise import PairwiseAlignments
    from tests.golden_util import hits_to_alignments

    odb = oracle.OracleDB(*pack_sequences_flat(db.genes))
    typer = Serotyper(
        db,
        aligner=lambda g: hits_to_alignments(db, g, odb.align(g.packed())),
        protein_aligner=lambda q, t: PairwiseAlignments.from_table(
            oracle.protein_align(q.seqs, q.offsets, q.lengths, t.seqs, t.offsets, t.lengths)),
    )  # fmt: skip
    return odb, typer


def _rows_of(results):
    from kaptive_amd.serotyping.io import KaptiveRow

    return [bytes(KaptiveRow.from_result(r)) for r in results]


def test_config3_k_and_o_at_full_size(oracle):
    """Config 3 shape: 5 Mbp assemblies that carry a K and an O locus, typed against both databases -- by one context
    per database sharing one device copy of the batch (as bench.py does) and by one shared alignment pass.  Hit tables
    equal the oracle's for both databases; TSV rows equal the host reduction's."""
    from kaptive_amd.engine import Engine

    db_k, db_o = make_db("kpsc_k", seed=100), make_db("kpsc_o", seed=101)
    genomes = [make_assembly(db_k, seed=7000 + i, also=(db_o,)) for i in range(4)]
    ids = [g.id for g in genomes]
    packed = [g.packed() for g in genomes]
    want_rows = []
    engines = [Engine(db_k), Engine(db_o)]
    first = engines[0].ctx.batch(packed)
    batches = [first, engines[1].ctx.batch(packed, device_words=first.device_words, after=first)]
    for b in batches:
        b.align_async()
    for db, eng, b in zip((db_k, db_o), engines, batches):
        odb, cpu = _oracle_typer(db, oracle)
        b.wait()
        hits, off = b.hits()
        for i, pa in enumerate(packed):
            _same_records(hits[off[i] : off[i + 1]], odb.align(pa), f"{db.metadata.keyword} hits of {ids[i]}")
        typer = Serotyper(db)
        typer._engine = eng
        got = eng.type_batch(typer, b, ids, aligned=True)
        want = _rows_of([cpu(g) for g in genomes])
        assert got.rows() == want, f"rows of {db.metadata.keyword}"
        assert got.tsv() == b"".join(want)
        want_rows.append(want)
    assert sum(b"Typeable" in r for r in want_rows[0]) >= 3 and sum(b"Typeable" in r for r in want_rows[1]) >= 3
    for b in reversed(batches):
        b.close()
    for e in engines:
        e.close()
    # the same through one alignment pass over the genes of both databases
    both = Engine([db_k, db_o])
    batch = both.ctx.batch(packed)
    batch.align_async()
    for i, db in enumerate((db_k, db_o)):
        typer = Serotyper(db)
        assert both.view(i).type_batch(typer, batch, ids, aligned=True).rows() == want_rows[i], f"shared pass, database {i}"
    batch.close()
    both.close()


def test_config4_acinetobacter_at_full_size(oracle):
    """Config 4 shape: the 240-locus A. baumannii-shaped K database, 4 Mbp assemblies at GC 0.39 in ~1 500 contigs,
    every locus split across contigs."""
    from kaptive_amd.engine import Engine

    db = make_db("ab_k", seed=102)
    assert len(db.loci) == 240
    genomes = [make_assembly(db, seed=8000 + i, length=4.0e6, median_contigs=1500, min_contig=200, force_split=True)
               for i in range(3)]  # fmt: skip
    assert min(len(g.contigs) for g in genomes) > 700
    ids = [g.id for g in genomes]
    packed = [g.packed() for g in genomes]
    odb, cpu = _oracle_typer(db, oracle)
    eng = Engine(db)
    batch = eng.ctx.batch(packed)
    hits, off = batch.align()
    for i, pa in enumerate(packed):
        _same_records(hits[off[i] : off[i + 1]], odb.align(pa), f"hits of {ids[i]}")
    typer = Serotyper(db)
    typer._engine = eng
    got = eng.type_batch(typer, batch, ids, aligned=True)
    want = [cpu(g) for g in genomes]
    assert got.rows() == _rows_of(want)
    assert all(len(r.locus_pieces) >= 2 for r in want)  # fragmented loci: the '?' problem is exercised
    batch.close()
    eng.close()


def test_parity_sweep_at_full_size(oracle):
    """Many full-size assemblies per configuration (KAPTIVE_AMD_SWEEP of them, default 128: BASELINE configs 2/3 -- 5 Mbp, K
    and O databases -- and config 4 -- 240 loci, 4 Mbp in ~1500 contigs), with divergence from 0 to 12 %, indels, N runs,
    second loci, tandem copies and insertions / deletions of 33-480 bases inside genes (joined hits, kp-align v5), and as
    many small assemblies whose locus copy carries 2-24 insertions / deletions of 1-520 bases anywhere ("storm"), and as many full-size ones on the
    background that is not iid ("paralog"): the device's hit tables equal the oracle's record for record and its report rows equal the
    host reduction's byte for byte, for every assembly and database.  The oracle runs in spawned workers; the summary of a large run is kept under profiles/."""
    import json
    import multiprocessing as mp
    import os

    from kaptive_amd.engine import Engine
    from tests import sweep_util as S

    n = int(os.environ.get("KAPTIVE_AMD_SWEEP", "128"))
    per_batch = int(os.environ.get("KAPTIVE_AMD_SWEEP_BATCH", "64"))  # (1024 x 5 Mbp: batch-wide base positions beyond 2^32)
    summary = {}
    with mp.get_context("spawn").Pool(min(16, max(2, (os.cpu_count() or 2) // 2))) as pool:
        for config in os.environ.get("KAPTIVE_AMD_SWEEP_CONFIGS", ",".join(S.CONFIGS)).split(","):
            jobs = [(config, i) for i in range(n)]
            pending = pool.map_async(S.oracle_hits, jobs, chunksize=1)
            made = [S.make(config, i) for i in range(n)]
            genomes, dbs = [m[0] for m in made], [d for d in made[0][1:] if d is not None]
            packed = [g.packed() for g in genomes]
            engines = [Engine(db) for db in dbs]
            typers = []
            for db, e in zip(dbs, engines):
                t = Serotyper(db)
                t._engine = e
                typers.append(t)
            compared = rows_equal = 0
            for lo in range(0, n, per_batch):
                part = packed[lo : lo + per_batch]
                first = engines[0].ctx.batch(part)
                batches = [first] + [e.ctx.batch(part, device_words=first.device_words, after=first) for e in engines[1:]]
                for b in batches:
                    b.align_async()
                got = []
                for b in batches:
                    b.wait()
                    got.append(b.hits())
                if lo == 0:
                    want = pending.get(timeout=3600)
                for k, (hits, off) in enumerate(got):
                    for i in range(len(part)):
                        _same_records(hits[off[i] : off[i + 1]], want[lo + i][0][k], f"{config} database {k}, assembly {lo + i} {S.assembly_kwargs(config, lo + i)}")
                        compared += int(off[i + 1] - off[i])
                    # ... and the rows of the device's reduction equal the host reduction's on the oracle's hits
                    ids = [g.id for g in genomes[lo : lo + per_batch]]
                    rows = engines[k].type_batch(typers[k], batches[k], ids, aligned=True).rows()
                    for i, row in enumerate(rows):
                        assert row == want[lo + i][1][k], f"{config} database {k}, row of assembly {lo + i} {S.assembly_kwargs(config, lo + i)}"
                    rows_equal += len(rows)
                for b in reversed(batches):
                    b.close()
            for e in engines:
                e.close()
            summary[config] = {"assemblies": n, "assemblies_per_batch": min(n, per_batch), "databases": len(dbs),
                               "hit_records_equal": compared, "report_rows_equal": rows_equal, "differing": 0}
    if out := os.environ.get("KAPTIVE_AMD_SWEEP_OUT"):
        with open(out, "w") as f:
            json.dump(summary, f, indent=1)
    assert all(v["hit_records_equal"] > 100 * n for v in summary.values())


def test_many_hit_stress_forces_every_fallback(oracle):
    """A full-size assembly carrying five K loci: more hits than the LDS sort stage holds (SORT_LDS = 4096), with hit,
    kept, piece and protein buffers started far too small, so every grow-and-rerun path runs at full size."""
    from kaptive_amd.engine import Engine
    from kaptive_amd.synth import mutate

    db = make_db("kpsc_k", seed=100)
    rng = np.random.default_rng(99)
    base = make_assembly(db, seed=9100, locus=3, p_break=0.0, p_is=0.0)
    seqs = [np.frombuffer(r.seq, np.uint8).copy() for r in base.contigs]
    big = int(np.argmax([len(s) for s in seqs]))
    extra = []
    for li in (10, 50, 90, 130):  # further loci appended as contigs of their own
        o, n = int(db.loci.offsets[li]), int(db.loci.lengths[li])
        extra.append(SeqRecord(f"extra_{li}", mutate(rng, db.loci.seqs[o : o + n], 0.01).tobytes()))
    recs = [SeqRecord(f"c{i}", s.tobytes()) for i, s in enumerate(seqs)] + extra
    assert len(seqs[big]) > 1000
    genome = GenomeAssembly("five_loci", Sequences.from_records(recs))
    plain = make_assembly(db, seed=9101)
    packed = [genome.packed(), plain.packed()]
    odb, cpu = _oracle_typer(db, oracle)
    eng = Engine(db)
    for name, v in (("hit_cap", 512), ("kept_cap", 8), ("piece_cap", 1), ("prot_cap", 2048), ("tasks_per_asm", 256)):
        eng.ctx.set_option(name, v)
    batch = eng.ctx.batch(packed)
    hits, off = batch.align()
    assert off[1] > 4096, f"{off[1]} hits: the stress case no longer exceeds SORT_LDS"
    for i, pa in enumerate(packed):
        _same_records(hits[off[i] : off[i + 1]], odb.align(pa), f"hits of assembly {i}")
    typer = Serotyper(db)
    typer._engine = eng
    got = eng.type_batch(typer, batch, [genome.id, plain.id], aligned=True)
    assert batch.stats()["retries"] >= 3
    assert got.rows() == _rows_of([cpu(genome), cpu(plain)])
    batch.close()
    eng.close()


# ---- context-owned work sets, streaming uploads --------------------------------------------------------------------------
def _reload_small(ctx, oracle, small_db):
    codes, off = pack_sequences_flat(small_db.genes)
    ctx.load_genes(codes, off)  # earlier tests of the module left other databases in the shared context
    return oracle.OracleDB(codes, off)


def test_work_slots_rotate_and_displaced_results_raise(ctx, oracle, small_db):
    odb = _reload_small(ctx, oracle, small_db)
    asms = [make_assembly(small_db, seed=600 + i, length=60_000, median_contigs=4, min_contig=200)
            for i in range(_native.WORK_SLOTS + 1)]
    batches = [ctx.batch([a.packed()]) for a in asms]
    for b in batches:
        b.align_async()  # one pass more than the context has work sets: the first batch's results are displaced
    for b, a in zip(batches[1:], asms[1:]):
        b.wait()
        hits, _ = b.hits()
        _same_records(hits, odb.align(a.packed()), a.id)
    with pytest.raises(_native.NativeError, match="no resident alignment results"):
        batches[0].wait()
    hits, _ = batches[0].align()  # aligned again it takes the next work set
    _same_records(hits, odb.align(asms[0].packed()), asms[0].id)
    for b in batches:
        b.close()


def test_async_upload_from_pinned_memory_and_shared_device_words(ctx, oracle, small_db):
    odb = _reload_small(ctx, oracle, small_db)
    asms = [make_assembly(small_db, seed=700 + i, length=80_000, median_contigs=6, min_contig=200) for i in range(4)]
    packed = [a.packed() for a in asms]
    pin = _native.PinnedBuffer(sum(len(p.words) for p in packed), np.uint32)
    pin.array[:] = np.concatenate([p.words for p in packed])
    up = ctx.batch(packed, pinned_words=pin.array)
    hits, off = up.align()  # the pass waits for the copy on the device
    other = _native.Context(0)
    codes, goff = pack_sequences_flat(small_db.genes)
    other.load_genes(codes, goff)
    twin = other.batch(packed, device_words=up.device_words, after=up)
    hits2, off2 = twin.align()
    assert np.array_equal(off, off2)
    _same_records(hits, hits2, "second context on the same device words")
    for i, pa in enumerate(packed):
        _same_records(hits[off[i] : off[i + 1]], odb.align(pa), asms[i].id)
    # the same words from a block that was reserved as plain huge-page memory, filled, and only then page-locked
    # (kp_host_reserve / kp_host_lock: what the command line's readers do before a context exists); large enough for the
    # mapped kind of kp_host_alloc as well
    lazy = _native.PinnedBuffer(3 << 20, np.uint32, lazy=True)
    assert not lazy.locked and lazy.array.ctypes.data % (2 << 20) == 0
    n = len(pin.array)
    lazy.array[:n] = pin.array
    before = _native.pinned_bytes()
    lazy.lock()
    lazy.lock()  # (idempotent)
    assert _native.pinned_bytes() - before >= 12 << 20
    big = _native.PinnedBuffer(3 << 20, np.uint32)  # eager: reserve + lock in one call
    big.array[:n] = pin.array
    for buf in (lazy, big):
        again = ctx.batch(packed, pinned_words=buf.array[:n])
        hits3, off3 = again.align()
        assert np.array_equal(off, off3)
        _same_records(hits, hits3, "words from a mapped block")
        again.close()
        held = _native.pinned_bytes()
        buf.close()
        assert held - _native.pinned_bytes() >= 12 << 20
    twin.close()
    other.close()
    up.close()
    pin.close()


def test_longest_genes_reach_the_packed_score_range(oracle):
    """Genes at KP_FILL16_MAX_GENE_LEN: the fill kernel's biased 16-bit scores (2 * length + 12) come within a few hundred of the
    half-precision infinity pattern 0x7C00 its three-way maxima (v_pk_maximum3_f16 on integer bit patterns) must stay below.  Exact copies, copies with substitutions, with an insertion and a deletion (wide bands), with an N run,
    on both strands.  (Longer genes: test_genes_beyond_the_packed_score_range.)"""
    from kaptive_amd.synth import revcomp

    rng = np.random.default_rng(4242)
    max_len = _native.FILL16_MAX_GENE_LEN
    g1, g2, g3 = (random_dna(rng, n, 0.5) for n in (max_len, max_len - 1, 15000))
    genes = Sequences.from_records([SeqRecord("g1", g1.tobytes()), SeqRecord("g2", g2.tobytes()), SeqRecord("g3", g3.tobytes())])
    codes, off = pack_sequences_flat(genes)
    ctx = _native.Context(0)
    ctx.load_genes(codes, off)
    odb = oracle.OracleDB(codes, off)
    sub = g2.copy()
    at = rng.choice(len(sub), 900, replace=False)
    sub[at] = np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, len(at))]
    indel = np.concatenate([g3[:5000], random_dna(rng, 9, 0.5), g3[5000:11000], g3[11006:]])
    with_n = g1.copy()
    with_n[7000:7030] = ord("N")
    pad = lambda n: random_dna(rng, n, 0.5)  # noqa: E731
    recs = [SeqRecord("exact", np.concatenate([pad(300), g1, pad(200)]).tobytes()),
            SeqRecord("rc_sub", np.concatenate([pad(100), revcomp(sub), pad(50)]).tobytes()),
            SeqRecord("indel", np.concatenate([pad(64), indel, pad(64)]).tobytes()),
            SeqRecord("n_run", np.concatenate([pad(10), with_n, pad(10)]).tobytes()),
            SeqRecord("tail_only", g1[9000:].tobytes())]  # fmt: skip
    asm = GenomeAssembly("long_genes", Sequences.from_records(recs))
    pa = asm.packed()
    batch = ctx.batch([pa])
    hits, _ = batch.align()
    want = odb.align(pa)
    _same_records(hits, want, "hits of the longest genes")
    assert hits["score"].max() == 2 * max_len and len(hits) >= 5
    batch.close()
    ctx.close()


def test_genes_beyond_the_packed_score_range(oracle):
    """The reference compiles any CDS (src/kaptive/db/core.py:402-404, 478).  Genes longer than KP_FILL16_MAX_GENE_LEN do
    not fit the packed 16-bit fill kernel: their tasks are filled with 32-bit scores (kp_sw_long_kernel), everything else
    in the same pass the usual way.  A 20 kb and a 41 kb gene -- exact, with substitutions, with an insertion and a
    deletion, with an N run, reverse strand, running off a contig end -- beside ordinary genes: hits equal the oracle's.
    Only genes beyond KP_MAX_GENE_LEN (16-bit query positions) are rejected at load."""
    from kaptive_amd.synth import revcomp

    rng = np.random.default_rng(777)
    big, huge = random_dna(rng, 20_000, 0.5), random_dna(rng, 41_003, 0.5)
    small = [random_dna(rng, n, 0.5) for n in (900, 1500, 15_800)]
    genes = Sequences.from_records([SeqRecord("big", big.tobytes()), SeqRecord("s0", small[0].tobytes()), SeqRecord("huge", huge.tobytes()),
                                    SeqRecord("s1", small[1].tobytes()), SeqRecord("s2", small[2].tobytes())])  # fmt: skip
    codes, off = pack_sequences_flat(genes)
    ctx = _native.Context(0)
    ctx.load_genes(codes, off)
    odb = oracle.OracleDB(codes, off)
    sub = big.copy()
    at = rng.choice(len(sub), 1200, replace=False)
    sub[at] = np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, len(at))]
    indel = np.concatenate([huge[:9000], random_dna(rng, 7, 0.5), huge[9000:30000], huge[30011:]])
    with_n = big.copy()
    with_n[12_000:12_025] = ord("N")
    pad = lambda n: random_dna(rng, n, 0.5)  # noqa: E731
    recs = [SeqRecord("exact", np.concatenate([pad(300), big, pad(200), small[0], pad(50)]).tobytes()),
            SeqRecord("rc_sub", np.concatenate([pad(100), revcomp(sub), pad(50), small[2], pad(20)]).tobytes()),
            SeqRecord("indel", np.concatenate([pad(64), indel, pad(64)]).tobytes()),
            SeqRecord("n_run", np.concatenate([pad(10), with_n, pad(10), revcomp(small[1])]).tobytes()),
            SeqRecord("tail_only", huge[25_000:].tobytes()),
            SeqRecord("head_only", np.concatenate([pad(5), revcomp(huge[:17_000])]).tobytes()),
            # ... and with insertions / deletions beyond a band's reach: joined alignments (kp-align v5) whose pieces are tasks
            # of the 32-bit fill, three pieces over 20 000 rows and two over 41 000
            SeqRecord("big_joined", np.concatenate([pad(40), big[:8000], big[8200:15_000], pad(90), big[15_000:], pad(40)]).tobytes()),
            SeqRecord("huge_joined_rc", revcomp(np.concatenate([pad(40), huge[:22_000], huge[22_450:], pad(40)])).tobytes())]  # fmt: skip
    asm = GenomeAssembly("longer_genes", Sequences.from_records(recs))
    pa = asm.packed()
    batch = ctx.batch([pa, pa])
    hits, hoff = batch.align()
    want = odb.align(pa)
    for i in range(2):
        _same_records(hits[hoff[i] : hoff[i + 1]], want, "hits of genes beyond the packed range")
    joins = odb.joins(pa)
    # (round 6: the chain of a group is made of its anchors -- up to KP_JOIN_ANCHOR_MAX = 4096 of them: the 20 kb gene's three
    # pieces are joined, the 41 kb gene's group holds ~7 400 anchors and stays two band-task hits, on the device as in the oracle)
    assert sorted(joins["n_pieces"][(joins["piece"][:, :, 0] == 1).any(axis=1)].tolist()) == [3], joins["n_pieces"]
    for f in joins.dtype.names:
        assert np.array_equal(np.sort(joins, order=["gs", "contig"])[f], np.sort(batch.joins(0), order=["gs", "contig"])[f]), f
    assert (want["score"] == 2 * 20_000).any() and want["score"].max() > 80_000 and (want["gene"] == 2).sum() >= 3 and (want["gene"] == 4).any()
    batch.close()
    too_long = Sequences.from_records([SeqRecord("g", random_dna(rng, _native.MAX_GENE_LEN + 1, 0.5).tobytes())])
    c2, o2 = pack_sequences_flat(too_long)
    with pytest.raises(ValueError):
        ctx.load_genes(c2, o2)
    ctx.close()
