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


@pytest.mark.parametrize("key", ["k", "o", "kfull", "abfull"])
def test_native_json_lines_equal_the_python_serialiser(key, oracle):
    """``BatchTyping.jsonl()`` (kp_format_json: one native call per batch) writes, for every golden case, the bytes that
    ``dumps_line(result(i).to_dict())`` writes -- the restatement of orjson's conventions the reference's ``-j`` output is held
    to (kaptive_amd/serotyping/jsonl.py; src/kaptive/serotyping/cli.py:67-76) --, and the lines read back through
    ``SerotypingResult.from_dict`` into the same rows."""
    import json

    from kaptive_amd.serotyping.io import KaptiveRow
    from kaptive_amd.serotyping.jsonl import dumps_line
    from kaptive_amd.serotyping.models import SerotypingResult

    names = [n for n in case_names() if not n.startswith("k_divergent_") and load_case(n)[0] == key]
    typer, genomes, exps, bt = _batch_from_cases(names, oracle)
    lines = bt.jsonl().splitlines(keepends=True)
    assert len(lines) == len(names)
    for i, name in enumerate(names):
        res = bt.result(i)
        want = dumps_line(res.to_dict())
        assert lines[i] == want, (name, next((j, lines[i][max(0, j - 60): j + 60], want[max(0, j - 60): j + 60]) for j in range(min(len(want), len(lines[i]))) if lines[i][j] != want[j]))
        back = SerotypingResult.from_dict(json.loads(lines[i]))
        assert bytes(KaptiveRow.from_result(back)) == bytes(KaptiveRow.from_result(res))


@pytest.mark.parametrize("key", ["k", "o", "kfull", "abfull"])
def test_native_fasta_records_equal_the_result_objects(key, oracle):
    """``BatchTyping.fasta()`` (kp_format_fasta: one native call per batch and kind) gives, per assembly, the bytes of
    ``result(i).locus_seqs / gene_seqs / translations .to_fasta()`` -- what ``-l / -g / -p`` write per assembly (reference:
    src/kaptive/serotyping/cli.py:78-114) -- for every golden case."""
    names = [n for n in case_names() if not n.startswith("k_divergent_") and load_case(n)[0] == key]
    typer, genomes, exps, bt = _batch_from_cases(names, oracle)
    got = bt.fasta()
    assert set(got) == {"loci", "genes", "proteins"} and all(len(v) == len(names) for v in got.values())
    n_records = 0
    for i, name in enumerate(names):
        res = bt.result(i)
        for kind, attr in (("loci", "locus_seqs"), ("genes", "gene_seqs"), ("proteins", "translations")):
            want = getattr(res, attr).to_fasta()
            assert got[kind][i] == want, (name, kind, got[kind][i][:120], want[:120])
            n_records += want.count(b">")
    assert n_records > 20 * len(names)
    assert bt.fasta(("genes",)).keys() == {"genes"}


def test_native_json_number_layout():
    """The number layouts of kp_format_json against jsonl.py's: float64 and float32 values across the switch points of Ryu's
    format (16 / 13 digits to the right, 5 / 6 zeros to the left), NaN, infinities, zeros, through a one-hit batch."""
    from kaptive_amd.serotyping.jsonl import format_f32, format_f64

    key, genome, hits, exp, scalars, kwargs = load_case("k_plain1")
    db = load_db(key)
    typer = Serotyper(db, aligner=lambda g: None)
    rng = np.random.default_rng(3)
    vals = [0.0, -0.0, 1.0, 100.0, 0.1, 1e15, 1e16, 1e17, 123456789012345680.0, 1e-5, 1e-6, 1e-7, 9.999999e-5, float("nan"), float("inf"),
            5e-324, 1.7976931348623157e308, 99.32123565673828, 28.33096164055609, 1e21, 1e22, 12345.678]
    vals += list(np.exp(rng.uniform(-40, 40, 200))) + [float(np.float32(v)) for v in np.exp(rng.uniform(-30, 30, 100))]
    sums = np.zeros(len(vals), B.SUMMARY_DTYPE)
    sums["n_kept"] = 1
    kept = np.zeros((len(vals), 1), B.KEPT_DTYPE)
    with np.errstate(over="ignore"):
        f32 = np.array(vals, np.float64).astype(np.float32)
    kept["coverage"][:, 0] = f32
    kept["pident"][:, 0] = f32[::-1]
    kept["t_end"] = 9
    kept["strand"] = 1
    pieces = np.zeros((len(vals), 1), B.PIECE_DTYPE)
    scores = np.zeros((len(vals), len(db.loci)))
    scores[:, 0] = vals
    bt = B.BatchTyping(typer, [f"g{i}" for i in range(len(vals))], sums, kept, pieces, scores, np.zeros(len(vals), np.int32), [genome] * len(vals))
    for i, line in enumerate(bt.jsonl().splitlines()):
        text = line.decode()
        assert f'"best_locus_score":{format_f64(vals[i])},' in text, (vals[i], text[:400])
        assert f'"coverages":[{format_f32(f32[i])}]' in text, (vals[i], text)
        assert f'"protein_identities":[{format_f32(f32[::-1][i])}]' in text, (vals[::-1][i], text)


def test_native_formatters_refuse_records_that_point_outside_their_tables():
    """kp_format_json / kp_format_fasta index the database's and the assembly's tables with values taken from the records:
    an untyped assembly (best locus -1), a gene or contig index past its table, an interval past the contig text or more
    kept hits than the stride are KP_EINVAL -- a ValueError here --, not a read past a buffer."""
    key, genome, hits, exp, scalars, kwargs = load_case("k_plain1")
    db = load_db(key)
    typer = Serotyper(db, aligner=lambda g: None)

    def batch(**edit):
        sums = np.zeros(1, B.SUMMARY_DTYPE)
        sums["n_kept"] = 1
        kept = np.zeros((1, 1), B.KEPT_DTYPE)
        kept["t_end"] = 9
        kept["strand"] = 1
        pieces = np.zeros((1, 1), B.PIECE_DTYPE)
        bt = B.BatchTyping(typer, ["g"], sums, kept, pieces, np.zeros((1, len(db.loci))), np.zeros(1, np.int32), [genome])
        bt.phenotype  # noqa: B018  (the host columns are computed from the valid records; the damage comes afterwards)
        for name, v in edit.items():
            if name == "best":
                bt.best_locus[:] = v
            elif name in sums.dtype.names:
                bt.sums[name] = v
            elif name.startswith("piece_"):
                bt.pieces[name[6:]] = v
            else:
                bt.kept[name] = v
        return bt

    assert batch().jsonl().startswith(b'{"kaptive_version"')
    n_text = len(genome.contigs.seqs)
    bad = [dict(best=-1), dict(best=len(db.loci)), dict(gene=len(db.genes.ids)), dict(gene=-1), dict(contig=len(genome.contigs.offsets)),
           dict(contig=-1), dict(t_end=n_text + 1), dict(t_start=-1), dict(t_start=12), dict(n_kept=2), dict(n_pieces=2),
           dict(n_pieces=1, piece_contig=-1), dict(n_pieces=1, piece_end=n_text + 5)]  # fmt: skip
    for edit in bad:
        with pytest.raises(ValueError):
            batch(**edit).jsonl()
        if "best" not in edit:
            with pytest.raises(ValueError):
                batch(**edit).fasta()
