"""The batched reduction's core functions (kaptive_amd/csrc/kp_reduce_core.h, the code the HIP kernels run) compiled
with g++ by a test-only harness and checked against the golden vectors recorded from the reference.  The protein DP in
between comes from the oracle here; the GPU tests repeat the whole thing with the HIP kernels."""

import numpy as np
import pytest

from kaptive_amd.serotyping import batch as B
from kaptive_amd.serotyping.core import Serotyper
from tests import harness_util as H
from tests.golden_util import GOLDEN, case_names, load_case, load_db
from tests.test_host_golden import _same, check_result_against_golden


@pytest.mark.parametrize("name", case_names())
def test_core_reduction_matches_reference(name, oracle):
    key, genome, hits, exp, scalars, kwargs = load_case(name)
    db = load_db(key)
    typer = Serotyper(db, aligner=lambda g: None, **kwargs)
    hdb, prm = H.HarnessDb(db), H.params(db, typer)

    # hit finalisation: any permutation (plus duplicated rows) of a recorded aligner table comes back as recorded.
    # (the random_hits tables are not aligner output: their mapq/order are arbitrary, so they skip this step)
    if not name.startswith("random_hits"):
        rng = np.random.default_rng(len(hits))
        perm = rng.permutation(len(hits) + len(hits) // 7)
        shuffled = np.concatenate([hits, hits[: len(hits) // 7]])[perm]
        # (between the aligner and the mapq computation a record carries its chain score in the mapq / pad bytes)
        chain = np.load(GOLDEN / f"typing_{name}.npz")["hit_chain_scores"]
        chain = np.concatenate([chain, chain[: len(hits) // 7]])[perm]
        bonus, chain = chain >> 16, np.minimum(chain & 0xFFFF, 65535)  # (a joined hit's order-score bonus rides above bit 16)
        shuffled["mapq"], shuffled["pad"] = chain & 255, chain >> 8
        shuffled["score"] |= bonus << 20
        again = H.finalise_hits(shuffled)
        assert again.tobytes() == np.ascontiguousarray(hits).tobytes()

    scores, counts = H.locus_scores(hits, hdb, typer.min_gene_coverage)
    best, final, completeness = B.choose_best_loci(scores[None, :], counts[None, :], typer._expected_genes_per_locus)
    _same(final[0], exp["last_scores"], "penalised locus scores")
    _same(completeness[0], exp["last_completeness"], "locus completeness")
    assert int(best[0]) == scalars["best_locus_idx"]

    pa = genome.packed()
    kept, pieces, summary, prot = H.reduce(hits, hdb, prm, best[0], pa)
    # proteins translated from the packed stream; protein DP by the oracle; then states
    q_off, q_len = kept["prot_off"], kept["prot_len"]
    t_off, t_len = db.translations.offsets[kept["gene"]], db.translations.lengths[kept["gene"]]
    dp = oracle.protein_align(prot, q_off, q_len, db.translations.seqs, t_off, t_len)
    kept = H.states(kept, hdb, prm, genome.contigs.lengths, dp)
    res = B.assemble(typer, genome.id, summary, kept, pieces, scores[best[0]], genome=genome)
    check_result_against_golden(res, exp, scalars)
    # device-translated proteins equal the host translation of the extracted genes
    keep = (kept["flags"] & B.F_SPURIOUS) == 0
    got = [bytes(prot[o : o + n]) for o, n in zip(kept["prot_off"][keep], kept["prot_len"][keep])]
    assert got == [r.seq for r in res.translations]
    # without the genome the rows are still exact (sequences are only needed for FASTA outputs)
    from kaptive_amd import KAPTIVE_COMPAT_VERSION
    from kaptive_amd.serotyping.io import KaptiveRow

    bare = B.assemble(typer, genome.id, summary, kept, pieces, scores[best[0]], genome=None)
    row = bytes(KaptiveRow.from_result(bare)).replace(KAPTIVE_COMPAT_VERSION.encode(), scalars["kaptive_version"].encode())
    assert row == bytes(exp["kaptive_row"])


def _batch_from_cases(names, oracle):
    """Stack several golden cases of one database into the arrays the device returns for a batch."""
    key0 = load_case(names[0])[0]
    db = load_db(key0)
    typer = Serotyper(db, aligner=lambda g: None)
    hdb, prm = H.HarnessDb(db), H.params(db, typer)
    rows, genomes, exps = [], [], []
    for name in names:
        key, genome, hits, exp, scalars, kwargs = load_case(name)
        assert key == key0 and not kwargs
        scores, counts = H.locus_scores(hits, hdb, typer.min_gene_coverage)
        best, _, _ = B.choose_best_loci(scores[None, :], counts[None, :], typer._expected_genes_per_locus)
        kept, pieces, summary, prot = H.reduce(hits, hdb, prm, best[0], genome.packed())
        dp = oracle.protein_align(prot, kept["prot_off"], kept["prot_len"], db.translations.seqs,
                                  db.translations.offsets[kept["gene"]], db.translations.lengths[kept["gene"]])
        kept = H.states(kept, hdb, prm, genome.contigs.lengths, dp, summary)
        rows.append((summary, kept, pieces, scores, best[0]))
        genomes.append(genome)
        exps.append((exp, scalars))
    n, kc, pc = len(rows), max(len(r[1]) for r in rows) + 3, max(len(r[2]) for r in rows) + 2
    sums = np.zeros(n, B.SUMMARY_DTYPE)
    kept = np.zeros((n, kc), B.KEPT_DTYPE)
    kept["pident"] = 55.5  # garbage beyond n_kept must not leak into any column
    pieces = np.zeros((n, pc), B.PIECE_DTYPE)
    scores = np.zeros((n, len(db.loci)))
    best = np.zeros(n, np.int32)
    for i, (s, k, p, sc, b) in enumerate(rows):
        sums[i], scores[i], best[i] = s, sc, b
        kept[i, : len(k)] = k
        pieces[i, : len(p)] = p
    return typer, genomes, exps, B.BatchTyping(typer, [g.id for g in genomes], sums, kept, pieces, scores, best, genomes)


@pytest.mark.parametrize("key", ["k", "o"])
def test_batch_columns_match_reference(key, oracle):
    names = [n for n in case_names() if not n.startswith("k_divergent_") and load_case(n)[0] == key]
    typer, genomes, exps, bt = _batch_from_cases(names, oracle)
    from kaptive_amd.serotyping.models import SerotypingProblem

    for i, (exp, scalars) in enumerate(exps):
        res = bt.result(i)
        check_result_against_golden(res, exp, scalars)
        # the vectorised columns agree with the per-assembly objects bit for bit
        assert bt.typeable[i] == res.typeable and bt.phenotype[i] == res.phenotype, names[i]
        assert SerotypingProblem(int(bt.problems[i])) == res.problems, names[i]
        for col, val in (("percent_identity", res.percent_identity), ("percent_coverage", res.percent_coverage),
                         ("completeness", res.best_locus_completeness), ("length_discrepancy", res.length_discrepancy),
                         ("best_score", res.best_locus_score)):  # fmt: skip
            assert np.float64(getattr(bt, col)[i]).tobytes() == np.float64(val).tobytes(), (names[i], col)
        assert bt.n_hits[i] == len(res.gene_hits) and bt.n_pieces[i] == len(res.locus_pieces)
    # the native batch formatter (kp_format_rows) gives the reference's TSV bytes for every assembly, in one call
    from kaptive_amd import KAPTIVE_COMPAT_VERSION

    want = b"".join(bytes(exp["kaptive_row"]).replace(scalars["kaptive_version"].encode(), KAPTIVE_COMPAT_VERSION.encode(), 1)
                    for exp, scalars in exps)
    assert bt.tsv() == want
    assert bt.tsv() == b"".join(bt.rows())


def test_float32_sum_has_numpys_association():
    """kp_np_sum_f32 (used for the batched mean identity) must reproduce np.add.reduce / np.mean on float32 arrays of
    every length class of numpy's pairwise summation (< 8, <= 128, split)."""
    rng = np.random.default_rng(0)
    for n in list(range(0, 150)) + [255, 256, 257, 1000, 2047]:
        vals = (rng.random(n) * 100).astype(np.float32)
        got = H.np_sum_f32(vals)
        assert got.tobytes() == np.float32(np.add.reduce(vals) if n else 0).tobytes(), n
        if n:
            assert np.float32(np.float64(got) / n).tobytes() == np.float32(np.mean(vals)).tobytes(), n


def test_primary_hit_is_the_top_scoring_one_whatever_the_emission_order(oracle):
    """Emission order ranks a joined hit by its order score (kp_spec.h), so a gene's hits are not in descending order of
    their alignment score; the primary hit of an expected gene is still its top-scoring kept hit (core.py:236-245).  The
    k_plain1 table with every gene's hits in random order: the device core's records equal the host reduction's."""
    from tests.golden_util import hits_to_alignments
    from tests.test_gpu_parity import _adversarial_hits, _results_equal

    key, genome, hits, exp, scalars, kwargs = load_case("k_split")
    db = load_db(key)
    typer = Serotyper(db, aligner=lambda g: None, protein_aligner=lambda q, t: __import__("kaptive_amd.core.pairwise", fromlist=["x"]).PairwiseAlignments.from_table(
        oracle.protein_align(q.seqs, q.offsets, q.lengths, t.seqs, t.offsets, t.lengths)), **kwargs)
    hdb, prm = H.HarnessDb(db), H.params(db, typer)
    rng = np.random.default_rng(5)
    pa = genome.packed()
    for it in range(6):
        table = hits[np.lexsort((rng.random(len(hits)), hits["gene"]))] if it < 3 else _adversarial_hits(rng, db, genome, 300, by_score=False)
        scores, counts = H.locus_scores(table, hdb, typer.min_gene_coverage)
        best, _, _ = B.choose_best_loci(scores[None, :], counts[None, :], typer._expected_genes_per_locus)
        kept, pieces, summary, prot = H.reduce(table, hdb, prm, best[0], pa)
        t_off, t_len = db.translations.offsets[kept["gene"]], db.translations.lengths[kept["gene"]]
        dp = oracle.protein_align(prot, kept["prot_off"], kept["prot_len"], db.translations.seqs, t_off, t_len)
        kept = H.states(kept, hdb, prm, genome.contigs.lengths, dp)
        res = B.assemble(typer, genome.id, summary, kept, pieces, scores[best[0]], genome=genome)
        _results_equal(res, typer.reduce(genome, hits_to_alignments(db, genome, table)), f"table {it}")
