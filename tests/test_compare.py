"""compare.LocusComparator against the reference's own outputs.

``tests/golden/compare.npz`` was written by running the reference's RandstrobeIndex, PairwiseAligner.align_seeds and
LocusComparator (oracle/make_golden.py::gen_compare) on five synthetic loci, one of them in two pieces.  Without a GPU:
the native randstrobe records and seeds, the oracle's seeded protein DP, the coordinate normalisation, and the whole
comparator with the oracle standing in for the device aligner.  With a GPU (-m gpu): the comparator as shipped."""

import numpy as np
import pytest

from kaptive_amd.compare import LocusComparator, LocusData
from kaptive_amd.core.interval import Intervals
from kaptive_amd.core.kmers import RandstrobeIndex, Seeds
from kaptive_amd.core.pairwise import PairwiseAlignments
from kaptive_amd.core.seq import Sequences
from kaptive_amd.serotyping.models import LocusPieces
from tests.golden_util import GOLDEN

COLS = ("scores", "matches", "mismatches", "gaps", "q_starts", "q_ends", "t_starts", "t_ends")


@pytest.fixture(scope="module")
def gold():
    return np.load(GOLDEN / "compare.npz")


def _inputs(z) -> list[LocusData]:
    out = []
    n_loci = len(z["locus_names"])
    for li in range(n_loci):
        prot = Sequences(tuple(str(x) for x in z[f"l{li}.ids"]), z[f"l{li}.seqs"], z[f"l{li}.offsets"], z[f"l{li}.lengths"])
        kw = {}
        if f"l{li}.gene_states" in z.files:
            kw = dict(pieces=LocusPieces(z["l2.piece_ctg"], z["l2.piece_starts"], z["l2.piece_ends"], z["l2.piece_strands"]),
                      gene_ctg_indices=z[f"l{li}.gene_ctg_indices"], gene_states=z[f"l{li}.gene_states"],
                      gene_descriptions=z[f"l{li}.gene_descriptions"])  # fmt: skip
        out.append(LocusData(proteins=prot, name=str(z["locus_names"][li]),
                             backbone=Intervals(z[f"l{li}.starts"], z[f"l{li}.ends"], z[f"l{li}.strands"]), **kw))  # fmt: skip
    return out


def _check(res, z):
    e = res.edges
    assert len(e) == len(z["edges.query_indices"]) > 20
    for col in ("query_locus_indices", "target_locus_indices", "query_indices", "target_indices", "global_query_indices",
                "global_target_indices"):  # fmt: skip
        assert np.array_equal(getattr(e, col), z[f"edges.{col}"]), col
    got = np.stack([getattr(e.alignments, c) for c in COLS], axis=1)
    bad = np.flatnonzero((got != z["edges.alignments"]).any(axis=1))
    assert len(bad) == 0, (bad[:5], got[bad[:3]], z["edges.alignments"][bad[:3]])
    assert res.locus_names == tuple(str(x) for x in z["locus_names"])
    assert np.array_equal(res.locus_lengths, z["locus_lengths"]) and np.array_equal(res.locus_offsets, z["locus_offsets"])
    assert [str(x) for x in res.gene_names] == [str(x) for x in z["gene_names"]]
    assert [str(x) for x in res.gene_descriptions] == [str(x) for x in z["gene_descriptions"]]
    assert np.array_equal(res.gene_states, z["gene_states"]) and res.gene_states.dtype == np.int8
    gi = res.gene_intervals
    for f in ("starts", "ends", "strands", "original_indices"):
        assert np.array_equal(getattr(gi, f), z[f"gi.{f}"]), f


def test_randstrobe_records_and_seeds_equal_reference(gold):
    loci = _inputs(gold)
    for li, inp in enumerate(loci):
        for sort, key in ((False, "records"), (True, "records_sorted")):
            got = RandstrobeIndex.build(inp.proteins, k=10, s=5, sort_by_hash=sort).records
            want = gold[f"l{li}.{key}"]
            assert len(got) == len(want) > 100
            for f in ("hash", "seq_idx", "pos1", "pos2"):
                assert np.array_equal(got[f], want[f]), (li, key, f)
    n_seeds = 0
    for i in range(len(loci)):
        for j in range(i + 1, len(loci)):
            t = RandstrobeIndex.build(loci[j].proteins, k=10, s=5, sort_by_hash=True)
            q = RandstrobeIndex.build(loci[i].proteins, k=10, s=5, sort_by_hash=False)
            sd = t.top_hits(q, min_score=1)
            want = gold[f"seeds.{i}.{j}"]
            got = np.stack([sd.query_indices.astype(np.int64), sd.target_indices.astype(np.int64), sd.scores.astype(np.int64),
                            sd.offsets.astype(np.int64)])  # fmt: skip
            assert np.array_equal(got, want), (i, j)
            assert np.array_equal(t.top_hits(loci[i].proteins).offsets, sd.offsets)  # raw sequences as queries
            n_seeds += len(sd)
    assert n_seeds == len(gold["edges.query_indices"])
    with pytest.raises(ValueError):
        RandstrobeIndex.build(loci[0].proteins, k=5, s=5)
    assert len(RandstrobeIndex.build(Sequences.empty())) == 0 and len(Seeds.empty()) == 0


class _OracleAligner:
    """The oracle's seeded DP behind PairwiseAligner's interface (tests only)."""

    def __init__(self, oracle, k=20):
        self.oracle, self.k = oracle, k

    def align_seeds(self, queries, targets, seeds):
        q, t = seeds.extract_sequences(queries, targets)
        return PairwiseAlignments.from_table(
            self.oracle.protein_align_seeded(q.seqs, q.offsets, q.lengths, t.seqs, t.offsets, t.lengths, seeds.offsets, self.k))


def test_comparator_with_the_oracle_aligner_equals_reference(gold, oracle):
    cmp = LocusComparator()
    cmp.aligner = _OracleAligner(oracle)
    _check(cmp(_inputs(gold)), gold)
    one = cmp(_inputs(gold)[:1])  # a single locus: no edges, coordinates still normalised
    assert len(one.edges) == 0 and one.gene_intervals.starts.min() == 0
    assert len(cmp([]).edges) == 0


@pytest.mark.gpu
def test_comparator_on_the_gpu_equals_reference(gold):
    _check(LocusComparator()(_inputs(gold)), gold)


@pytest.mark.gpu
def test_seeded_protein_kernel_matches_oracle_on_wide_and_long_pairs(oracle):
    """Seeded mode outside the comparator's comfort zone: offsets far from 0, k up to 40 (register kernel) and proteins
    longer than the register kernel stages (row-strip kernel), empty sequences."""
    from kaptive_amd import _native

    rng = np.random.default_rng(31)
    aa = np.frombuffer(b"ARNDCQEGHILKMFPSTWYV", np.uint8)
    qs, ts, offs = [], [], []
    for n, shift, sub in ((300, 0, 0.1), (300, 57, 0.1), (300, -80, 0.2), (900, 120, 0.1), (1400, -300, 0.15), (40, 5, 0.0),
                          (0, 0, 0.0), (250, 400, 0.1), (600, 3, 0.3)):  # fmt: skip
        core = aa[rng.integers(0, 20, size=n)]
        q = core.copy()
        hit = rng.random(n) < sub
        q[hit] = aa[rng.integers(0, 20, size=int(hit.sum()))]
        for s in np.flatnonzero(rng.random(len(q)) < 0.01)[::-1]:
            q = np.delete(q, slice(s, s + 2)) if rng.random() < 0.5 else np.insert(q, s, aa[rng.integers(0, 20, size=3)])
        pad = aa[rng.integers(0, 20, size=abs(shift))]
        # offset = query position - target position of the homologous residues
        if shift >= 0:
            q, t = np.concatenate([pad, q]), core
        else:
            q, t = q, np.concatenate([pad, core])
        qs.append(q.tobytes()); ts.append(t.tobytes() + b"*"); offs.append(shift)
    q, t = Sequences.from_bytes(qs), Sequences.from_bytes(ts)
    ctx = _native.Context(0)
    for k in (20, 8, 40):
        want = oracle.protein_align_seeded(q.seqs, q.offsets, q.lengths, t.seqs, t.offsets, t.lengths, offs, k)
        got = ctx.protein_align_seeded(q.seqs, q.offsets, q.lengths, t.seqs, t.offsets, t.lengths, np.array(offs, np.int32), k)
        bad = np.flatnonzero((want != got).any(axis=1))
        assert len(bad) == 0, (k, bad, want[bad[:3]], got[bad[:3]])
        assert (want[:, 0] > 50).sum() >= 6
    ctx.close()
