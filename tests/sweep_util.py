"""Workers of the full-size parity sweep (tests/test_gpu_parity.py::test_parity_sweep_at_full_size): the CPU oracle's hit
tables for seeded synthetic assemblies, computed in spawned processes (the test's own process holds a HIP context, which
does not survive a fork).  TEST INFRASTRUCTURE: imports oracle/."""

from __future__ import annotations

import numpy as np

_STATE: dict = {}

# (database of the assembly's main locus, seed of that database, databases typed against)
CONFIGS = {
    "kpsc": dict(main=("kpsc_k", 100), also=("kpsc_o", 101), asm={}),
    "ab_k": dict(main=("ab_k", 102), also=None, asm=dict(length=4.0e6, median_contigs=1500, min_contig=200, force_split=True)),
    # small assemblies whose locus copy is riddled with insertions / deletions anywhere (several per gene, gene ends,
    # spacers): chains of three and more pieces, weak end pieces, events closer together than a band is wide
    "storm": dict(main=("kpsc_k", 100), also=("kpsc_o", 101), asm=dict(length=3.0e5, median_contigs=6, min_contig=200, p_is=0.0)),
}
# ... and full-size assemblies whose background is not iid (diverged relatives of database genes, IS-like repeats, an operon
# in seven copies: wide bands, occurrence cuts, many weak tasks -- bench.py --background paralog)
CONFIGS["paralog"] = dict(main=("kpsc_k", 100), also=("kpsc_o", 101), asm=dict(background="paralog"))
STORM_SIZES = ((33, 500), (20, 60), (100, 300), (400, 520), (1, 40), (30, 36), (490, 510))


def assembly_kwargs(config: str, i: int) -> dict:
    """Deterministic variety: divergence from 0 to 12 %, indels, N runs, a second locus, a tandem gene copy, mid-size indels."""
    kw = dict(CONFIGS[config]["asm"])
    if config == "storm":
        kw["sub_rate"] = [0.0, 0.01, 0.03, 0.06, 0.1][i % 5]
        kw["indel_storm"] = (2 + (7 * i) % 23, *STORM_SIZES[i % len(STORM_SIZES)])
        if i % 6 == 5:
            kw["tandem_gene"] = 1
        if i % 9 == 7:
            kw["median_contigs"], kw["force_split"] = 150, True
        return kw
    kw["sub_rate"] = [None, 0.0, 0.01, 0.03, 0.06, 0.09, 0.12, None][i % 8]
    if i % 5 == 1:
        kw["indel_rate"] = 2e-3
    if i % 6 == 2:
        kw["n_run"] = 3
    if i % 7 == 3 and config == "kpsc":
        kw["second_locus"] = (i * 13) % 100
    if i % 9 == 4:
        kw["tandem_gene"] = 1
    if i % 4 == 3:  # insertions / deletions of 33-480 bases inside genes: joined alignments (kp_spec.h, kp-align v4)
        size = 33 + (41 * i) % 448
        kind, other = ("del", "ins") if i % 8 == 3 else ("ins", "del")
        kw["mid_indels"] = ((size, kind), (size + 1, kind), (33 + (size * 7) % 448, other))
    if i % 10 == 7:  # (round 6) events 40-100 bases from a gene's start or end and a pair that nearly cancels: weak end pieces,
        off = (40, 70, 100)[(i // 10) % 3] * (-1 if (i // 30) % 2 else 1)  # pieces that share a diagonal (kp_spec.h, kp-align v5)
        kw["placed_indels"] = ((("del", 45 + i % 60, off),), (("ins", 61 + i % 140, off),), (("del", 40, 200), ("ins", 50, 640)))
    return kw


def make(config: str, i: int):
    from kaptive_amd.synth import make_assembly, make_db

    if ("dbs", config) not in _STATE:
        c = CONFIGS[config]
        _STATE["dbs", config] = (make_db(*[c["main"][0]], seed=c["main"][1]),
                                 make_db(c["also"][0], seed=c["also"][1]) if c["also"] else None)
    main, also = _STATE["dbs", config]
    return make_assembly(main, seed=31_000 + 97 * i, also=(also,) if also is not None else (), **assembly_kwargs(config, i)), main, also


def oracle_hits(job):
    """(config, i) -> ([hit table per database] from the oracle's aligner, [KaptiveRow bytes per database] from the host
    reduction -- the statement the reference goldens pin -- fed by those hits and the oracle's protein DP)."""
    from kaptive_amd.core.pairwise import PairwiseAlignments
    from kaptive_amd.pack import pack_sequences_flat
    from kaptive_amd.serotyping.core import Serotyper
    from kaptive_amd.serotyping.io import KaptiveRow
    from oracle import oracle as O
    from tests.golden_util import hits_to_alignments

    config, i = job
    g, main, also = make(config, i)
    hits, rows = [], []
    for k, db in enumerate((main, also)):
        if db is None:
            continue
        if ("odb", config, k) not in _STATE:
            _STATE["odb", config, k] = O.OracleDB(*pack_sequences_flat(db.genes))
        h = np.array(_STATE["odb", config, k].align(g.packed()))
        hits.append(h)
        typer = Serotyper(
            db, aligner=lambda genome, db=db, h=h: hits_to_alignments(db, genome, h),
            protein_aligner=lambda q, t: PairwiseAlignments.from_table(
                O.protein_align(q.seqs, q.offsets, q.lengths, t.seqs, t.offsets, t.lengths)),
        )  # fmt: skip
        rows.append(bytes(KaptiveRow.from_result(typer(g))))
    return hits, rows
