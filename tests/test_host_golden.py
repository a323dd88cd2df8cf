"""Host-side mirror of the reference interface (kaptive_amd.core / .serotyping) against the golden vectors.

The typing cases replay a recorded hit table through ``Serotyper.reduce`` and compare every field of the result, and
the TSV rows byte for byte, with what the reference's own ``Serotyper.__call__`` returned for the same hits
(src/kaptive/serotyping/core.py:124-486).  The protein stage is supplied by the oracle here (no GPU in this suite);
tests/test_gpu_parity.py repeats the comparison with the HIP kernels in place.
"""

import numpy as np
import pytest

from kaptive_amd import KAPTIVE_COMPAT_VERSION
from kaptive_amd.core.interval import Intervals
from kaptive_amd.core.pairwise import PairwiseAlignments
from kaptive_amd.core.seq import Sequences
from kaptive_amd.serotyping.core import Serotyper
from kaptive_amd.serotyping.io import KaptiveRow, Pha4geRow
from tests.golden_util import case_names, hits_to_alignments, load_case, load_db


def test_intervals_match_reference(golden_dir):
    z = np.load(golden_dir / "intervals.npz")
    for i in range(int(z["n_cases"])):
        s, e, g, order = (z[f"c{i}_{k}"] for k in ("starts", "ends", "groups", "order"))
        iv = Intervals(s, e, np.ones(len(s), np.int8))
        assert np.array_equal(iv.cull_overlaps(order=order, max_overlap_fraction=0.1, group_by=g), z[f"c{i}_kept"])
        assert np.array_equal(iv.cluster_spatial(tolerance=int(z[f"c{i}_tol"]), group_by=g), z[f"c{i}_clusters"])


def test_sequences_match_reference(golden_dir):
    z = np.load(golden_dir / "seqs.npz")
    seqs = Sequences(tuple(str(i) for i in range(len(z["offsets"]))), z["seqs"], z["offsets"], z["lengths"])
    ex = seqs.extract(z["ex_idx"], z["ex_starts"], z["ex_ends"], z["ex_strands"])
    assert np.array_equal(ex.seqs, z["ex_seqs"]) and np.array_equal(ex.offsets, z["ex_offsets"])
    assert ex.ids[0] == "0_0_6_-1"
    for to_stop in (0, 1):
        tr = seqs.translate(frames=z["frames"], to_stop=bool(to_stop))
        assert np.array_equal(tr.seqs, z[f"tr{to_stop}_seqs"]) and np.array_equal(tr.lengths, z[f"tr{to_stop}_lengths"])
        assert np.array_equal(tr.offsets, z[f"tr{to_stop}_offsets"])
    tr = seqs.translate()
    assert np.array_equal(tr.seqs, z["tr_default_seqs"])


def _same(got, exp, what):
    got, exp = np.asarray(got), np.asarray(exp)
    assert got.shape == exp.shape, (what, got.shape, exp.shape)
    if exp.dtype.kind == "f":  # floats are compared by bit pattern (NaN == NaN)
        assert got.dtype == exp.dtype, (what, got.dtype, exp.dtype)
        assert got.tobytes() == exp.tobytes(), what
    elif exp.dtype.kind == "U":
        assert [str(x) for x in got.ravel()] == [str(x) for x in exp.ravel()], what
    else:
        assert np.array_equal(got, exp), what


def check_result_against_golden(res, exp, scalars):
    d = res.to_dict()
    for k in ("best_locus_idx", "best_locus_name", "phenotype", "typeable", "genome", "database_name",
              "database_version", "database_organism", "database_taxon"):  # fmt: skip
        assert d[k] == scalars[k], k
    assert list(d["missing_expected_genes"]) == scalars["missing_expected_genes"]
    assert int(d["problems"]) == scalars["problems"]
    for k in ("best_locus_score", "best_locus_completeness", "length_discrepancy", "percent_identity",
              "percent_coverage"):  # fmt: skip
        _same(np.float64(d[k]), exp[k], k)
    _same(d["gene_states"], exp["gene_states"], "gene_states")
    _same(d["protein_identities"], exp["protein_identities"], "protein_identities")
    for k, v in d["gene_hits"].items():
        _same(np.array(v, dtype="U") if isinstance(v, list) else v, exp[f"gene_hits.{k}"], f"gene_hits.{k}")
        if not isinstance(v, list):
            assert np.asarray(v).dtype == exp[f"gene_hits.{k}"].dtype, k
    for k, v in d["locus_pieces"].items():
        _same(v, exp[f"locus_pieces.{k}"], f"locus_pieces.{k}")
    for group in ("locus_seqs", "gene_seqs", "translations"):
        s = d[group]
        _same(np.array(list(s["ids"]), dtype="U"), exp[f"{group}.ids"], f"{group}.ids")
        _same(np.frombuffer(s["seqs"].encode(), np.uint8), exp[f"{group}.seqs"], f"{group}.seqs")
        _same(s["lengths"], exp[f"{group}.lengths"], f"{group}.lengths")
        _same(s["offsets"], exp[f"{group}.offsets"], f"{group}.offsets")
    # TSV rows, byte for byte, after swapping in the version string the reference printed
    for row_cls, key in ((KaptiveRow, "kaptive_row"), (Pha4geRow, "pha4ge_row")):
        ours = bytes(row_cls.from_result(res)).replace(KAPTIVE_COMPAT_VERSION.encode(), scalars["kaptive_version"].encode())
        assert ours == bytes(exp[key]), key


@pytest.mark.parametrize("name", case_names())
def test_typing_case_matches_reference(name, oracle):
    key, genome, hits, exp, scalars, kwargs = load_case(name)
    db = load_db(key)

    def oracle_proteins(q, t):
        return PairwiseAlignments.from_table(oracle.protein_align(q.seqs, q.offsets, q.lengths, t.seqs, t.offsets, t.lengths))

    alns = hits_to_alignments(db, genome, hits)
    typer = Serotyper(db, aligner=lambda g: alns, protein_aligner=oracle_proteins, **kwargs)
    res = typer(genome)
    _same(typer._last_scores, exp["last_scores"], "final locus scores")
    _same(typer._last_completeness, exp["last_completeness"], "locus completeness")
    check_result_against_golden(res, exp, scalars)


def test_header_bytes():
    assert KaptiveRow.header().startswith(b"Kaptive version\tDatabase name\tDatabase version\tAssembly\tBest match locus")
    assert b"Expected genes in locus, details" in KaptiveRow.header()
    assert Pha4geRow.header().startswith(b"sample\tgenotyping_method\t")
