"""The batched reduction's core functions (kaptive_amd/csrc/kp_reduce_core.h, the code the HIP kernels run) compiled
with g++ by a test-only harness and checked against the golden vectors recorded from the reference.  The protein DP in
between comes from the oracle here; the GPU tests repeat the whole thing with the HIP kernels."""

import numpy as np
import pytest

from kaptive_amd.serotyping import batch as B
from kaptive_amd.serotyping.core import Serotyper
from tests import harness_util as H
from tests.golden_util import case_names, load_case, load_db
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
        shuffled = np.concatenate([hits, hits[: len(hits) // 7]])[rng.permutation(len(hits) + len(hits) // 7)]
        shuffled["mapq"] = 0
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
