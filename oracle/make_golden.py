"""Generates tests/golden/* by running the REFERENCE itself (shimmed import, see oracle/ref_env.py).

Run in the build container only:  python -m oracle.make_golden
Inputs come from kaptive_amd.synth (fixed seeds below) and, for the typing cases, from the oracle aligner
(oracle/kp_oracle.c) whose hit tables are replayed into the reference's Serotyper through the rammappy stand-in.
Every expected value in the fixtures is an output of reference code:

  protein_dp.npz   kaptive.core.pairwise.PairwiseAligner            (src/kaptive/core/pairwise.py:255-584)
  intervals.npz    kaptive.core.interval.Intervals.cull_overlaps / cluster_spatial (interval.py:435-493)
  seqs.npz         kaptive.core.seq.Sequences.extract / translate    (src/kaptive/core/seq.py:327-408)
  typing_*.npz     kaptive.serotyping.core.Serotyper.__call__ + KaptiveRow/Pha4geRow (core.py:124-486, io.py)
"""

from __future__ import annotations

import json
import sys
import time
from pathlib import Path

import numpy as np

from oracle import ref_env

ref_env.activate()

import rammappy  # noqa: E402  (the replay stand-in)
from kaptive.core.genome import GenomeAssembly as RefGenome  # noqa: E402
from kaptive.core.interval import Intervals as RefIntervals  # noqa: E402
from kaptive.core.pairwise import PairwiseAligner as RefPairwiseAligner  # noqa: E402
from kaptive.core.seq import SeqRecord as RefSeqRecord  # noqa: E402
from kaptive.core.seq import Sequences as RefSequences  # noqa: E402
from kaptive.db.core import Database as RefDatabase  # noqa: E402
from kaptive.db.models import DatabaseMetadata as RefMeta  # noqa: E402
from kaptive.db.models import Phenotypes as RefPhenotypes  # noqa: E402
from kaptive.serotyping.core import Serotyper as RefSerotyper  # noqa: E402
from kaptive.serotyping.io import KaptiveRow as RefKaptiveRow  # noqa: E402
from kaptive.serotyping.io import Pha4geRow as RefPha4geRow  # noqa: E402

from kaptive_amd.pack import pack_sequences_flat  # noqa: E402
from kaptive_amd.synth import make_assembly, make_db, mutate, random_dna  # noqa: E402
from oracle import oracle as O  # noqa: E402

OUT = Path(__file__).resolve().parent.parent / "tests" / "golden"
AA = np.frombuffer(b"ARNDCQEGHILKMFPSTWYV", np.uint8)


def ref_sequences(ids, seqs, offsets, lengths) -> RefSequences:
    return RefSequences(tuple(ids), np.asarray(seqs), np.asarray(offsets, np.int32), np.asarray(lengths, np.int32))


def to_ref_db(db) -> RefDatabase:
    ph = db.phenotypes
    return RefDatabase(
        metadata=RefMeta.from_dict(db.metadata.to_dict()),
        loci=ref_sequences(db.loci.ids, db.loci.seqs, db.loci.offsets, db.loci.lengths),
        serotypes=db.serotypes,
        locus_gene_offsets=db.locus_gene_offsets,
        locus_gene_lengths=db.locus_gene_lengths,
        gene_intervals=RefIntervals(db.gene_intervals.starts, db.gene_intervals.ends, db.gene_intervals.strands),
        genes=ref_sequences(db.genes.ids, db.genes.seqs, db.genes.offsets, db.genes.lengths),
        translations=ref_sequences(db.translations.ids, db.translations.seqs, db.translations.offsets,
                                   db.translations.lengths),
        extra_genes=db.extra_genes,
        gene_locus_indices=db.gene_locus_indices,
        cluster_keys=db.cluster_keys,
        gene_cluster_ids=db.gene_cluster_ids,
        description_keys=db.description_keys,
        gene_description_ids=db.gene_description_ids,
        gene_positions=db.gene_positions,
        phenotypes=RefPhenotypes(ph.ids, ph.locus_masks, ph.extra_masks, ph.inactive_masks, ph.extra_counts,
                                 ph.priorities, ph.as_suffix),
        loci_sketches=None,
    )  # fmt: skip


# ---------------------------------------------------------------------------------------------------------------------
def gen_protein_dp(rng: np.random.Generator) -> None:
    pairs: list[tuple[bytes, bytes]] = [
        (b"MKTLLILAVSAAAGW", b"MKTLLILAVAAAAGW*"), (b"MSKGEELFTG", b"MSKGEELFTG*"),
        (b"MKTLLILAVSAAGWPQRSTVWYFMKTLLILAV", b"MKTLLILAVSAAAGWPQRSTVWYFMKTLLILAV*"), (b"MKV", b"MKTLLILAV*"),
        (b"", b"MKT*"), (b"MKT", b""), (b"", b""), (b"MKTXLIL", b"MKTLLIL*"), (b"mktllil", b"MKTLLIL*"),
        (b"WWWW", b"PPPP*"), (b"M", b"M"), (b"BJZX*", b"BJZX*"), (b"MKT-LIL", b"MKT-LIL*"),
    ]  # fmt: skip

    def prot(n):
        return AA[rng.integers(0, 20, size=n)]

    def mutated(p, sub, indel):
        p = p.copy()
        hit = rng.random(len(p)) < sub
        p[hit] = AA[rng.integers(0, 20, size=int(hit.sum()))]
        for s in np.flatnonzero(rng.random(len(p)) < indel)[::-1]:
            k = int(rng.integers(1, 6))
            p = np.delete(p, slice(s, s + k)) if rng.random() < 0.5 else np.insert(p, s, prot(k))
        return p

    for _ in range(40):
        t = prot(int(rng.integers(20, 180)))
        q = mutated(t, float(rng.uniform(0, 0.4)), float(rng.uniform(0, 0.03)))
        mode = rng.integers(0, 6)
        if mode == 0:  # truncated query (premature stop): band must widen
            q = q[: int(rng.integers(1, max(2, len(q) // 2)))]
        elif mode == 1:  # query starts late
            q = q[int(rng.integers(1, max(2, len(q) // 2))) :]
        elif mode == 2:  # unknown residues
            q[rng.integers(0, len(q), size=3)] = ord("X")
        pairs.append((q.tobytes(), t.tobytes() + b"*"))
    for _ in range(6):  # unrelated pairs
        pairs.append((prot(int(rng.integers(5, 90))).tobytes(), prot(int(rng.integers(5, 90))).tobytes() + b"*"))
    # a long-gap case inside the default band, and one just outside it
    base = prot(120)
    pairs.append((np.delete(base, slice(40, 58)).tobytes(), base.tobytes() + b"*"))
    pairs.append((np.delete(base, slice(40, 64)).tobytes(), base.tobytes() + b"*"))

    q = RefSequences.from_bytes([p[0] for p in pairs])
    t = RefSequences.from_bytes([p[1] for p in pairs])
    t0 = time.time()
    r = RefPairwiseAligner()(q, t)
    print(f"protein_dp: {len(pairs)} pairs through the reference in {time.time() - t0:.1f}s")
    np.savez_compressed(
        OUT / "protein_dp.npz",
        q_seqs=q.seqs, q_offsets=q.offsets, q_lengths=q.lengths, t_seqs=t.seqs, t_offsets=t.offsets,
        t_lengths=t.lengths, scores=r.scores, matches=r.matches, mismatches=r.mismatches, gaps=r.gaps,
        q_starts=r.q_starts, q_ends=r.q_ends, t_starts=r.t_starts, t_ends=r.t_ends, pidents=r.pidents,
    )  # fmt: skip


def gen_intervals(rng: np.random.Generator) -> None:
    cases = {}
    # the survey's known answer first
    fixed = dict(starts=[0, 50, 95, 300, 0], ends=[100, 150, 200, 400, 100], groups=[0, 0, 0, 0, 1],
                 order=[0, 1, 2, 3, 4], tol=99)  # fmt: skip
    specs = [fixed]
    for n in (1, 2, 7, 40, 200, 600):
        starts = rng.integers(0, 5000, size=n)
        lens = rng.integers(-3, 900, size=n)  # a few empty / negative spans
        specs.append(dict(starts=starts, ends=starts + lens, groups=rng.integers(0, 3, size=n),
                          order=rng.permutation(n), tol=int(rng.integers(0, 400))))  # fmt: skip
    for i, sp in enumerate(specs):
        s, e = np.asarray(sp["starts"], np.int32), np.asarray(sp["ends"], np.int32)
        g, order = np.asarray(sp["groups"], np.int32), np.asarray(sp["order"], np.int32)
        iv = RefIntervals(s, e, np.ones(len(s), np.int8))
        cases[f"c{i}_starts"], cases[f"c{i}_ends"], cases[f"c{i}_groups"], cases[f"c{i}_order"] = s, e, g, order
        cases[f"c{i}_tol"] = np.int64(sp["tol"])
        cases[f"c{i}_kept"] = iv.cull_overlaps(order=order, max_overlap_fraction=0.1, group_by=g)
        cases[f"c{i}_clusters"] = iv.cluster_spatial(tolerance=sp["tol"], group_by=g)
    cases["n_cases"] = np.int64(len(specs))
    np.savez_compressed(OUT / "intervals.npz", **cases)
    print(f"intervals: {len(specs)} cases")


def gen_seqs(rng: np.random.Generator) -> None:
    recs = [b"AACCGGTTNacgt", b"CATGAAANNNTTTtaaGGG", b"AT", b"ATGTGA", b"", b"ATGAAATAGCCC"]
    for _ in range(10):
        s = random_dna(rng, int(rng.integers(1, 400)), 0.5)
        for pos in rng.integers(0, len(s), size=int(rng.integers(0, 4))):
            s[pos] = rng.choice(np.frombuffer(b"NnRYacgtUu-", np.uint8))
        recs.append(s.tobytes())
    seqs = RefSequences.from_bytes(recs)
    n = 60
    idx = rng.integers(0, len(recs), size=n).astype(np.int32)
    a = (rng.random(n) * (seqs.lengths[idx] + 1)).astype(np.int32)
    b = (rng.random(n) * (seqs.lengths[idx] + 1)).astype(np.int32)
    starts, ends = np.minimum(a, b), np.maximum(a, b)
    strands = rng.choice(np.array([1, -1, 0], np.int8), size=n)
    # known answers from SURVEY.md Appendix C come first
    idx[:2], starts[:2], ends[:2], strands[:2] = [0, 0], [0, 2], [6, 13], [-1, -1]
    ex = seqs.extract(idx, starts, ends, strands)
    frames = rng.integers(0, 3, size=len(seqs)).astype(np.int8)
    frames[1:4] = [1, 0, 0]
    out = dict(seqs=seqs.seqs, offsets=seqs.offsets, lengths=seqs.lengths, ex_idx=idx, ex_starts=starts,
               ex_ends=ends, ex_strands=strands, ex_seqs=ex.seqs, ex_offsets=ex.offsets, ex_lengths=ex.lengths,
               frames=frames)  # fmt: skip
    for to_stop in (0, 1):
        tr = seqs.translate(frames=frames, to_stop=bool(to_stop))
        out[f"tr{to_stop}_seqs"], out[f"tr{to_stop}_offsets"], out[f"tr{to_stop}_lengths"] = (
            tr.seqs, tr.offsets, tr.lengths)  # fmt: skip
    tr = seqs.translate()
    out["tr_default_seqs"], out["tr_default_lengths"] = tr.seqs, tr.lengths
    np.savez_compressed(OUT / "seqs.npz", **out)
    print(f"seqs: {len(recs)} records, {n} extractions")


# ---------------------------------------------------------------------------------------------------------------------
def run_reference_typing(ref_db, serotyper, genome, hits) -> dict:
    """Replay `hits` (HIT_DTYPE, emission order) through the reference Serotyper for `genome`."""
    ref_genome = RefGenome.from_records(genome.id, [RefSeqRecord(r.id, r.seq) for r in genome.contigs])
    by_gene: dict[int, list] = {}
    names = genome.contigs.ids
    lens = genome.contigs.lengths
    for h in hits:
        c = int(h["contig"])
        by_gene.setdefault(int(h["gene"]), []).append(
            rammappy.make_hit(names[c].encode(), lens[c], h["q_start"], h["q_end"], h["t_start"], h["t_end"],
                              int(h["strand"]), h["score"], h["matches"], h["block_len"], h["mapq"])
        )  # fmt: skip
    rammappy.PENDING_HITS = by_gene
    res = serotyper(ref_genome)
    rammappy.PENDING_HITS = {}
    d = res.to_dict()
    exp: dict = {}
    scalars = {}
    for k, v in d.items():
        if k in ("locus_pieces", "gene_hits"):
            for kk, vv in v.items():
                exp[f"{k}.{kk}"] = np.asarray(vv) if not isinstance(vv, list) else np.array(vv, dtype="U")
        elif k in ("locus_seqs", "gene_seqs", "translations"):
            exp[f"{k}.ids"] = np.array(list(v["ids"]), dtype="U")
            exp[f"{k}.seqs"] = np.frombuffer(v["seqs"].encode("ascii"), np.uint8)
            exp[f"{k}.offsets"], exp[f"{k}.lengths"] = v["offsets"], v["lengths"]
        elif isinstance(v, np.ndarray):
            exp[k] = v
        elif k == "missing_expected_genes":
            scalars[k] = list(v)
        elif k == "problems":
            scalars[k] = int(v)
        else:
            scalars[k] = v.item() if isinstance(v, np.generic) else v
    # NaN is not JSON: carry length_discrepancy as an array
    exp["length_discrepancy"] = np.float64(scalars.pop("length_discrepancy"))
    exp["best_locus_score"] = np.float64(scalars.pop("best_locus_score"))
    exp["percent_identity"] = np.float64(scalars.pop("percent_identity"))
    exp["percent_coverage"] = np.float64(scalars.pop("percent_coverage"))
    exp["best_locus_completeness"] = np.float64(scalars.pop("best_locus_completeness"))
    exp["last_scores"] = serotyper._last_scores
    exp["last_completeness"] = serotyper._last_completeness
    exp["kaptive_row"] = np.frombuffer(bytes(RefKaptiveRow.from_result(res)), np.uint8)
    exp["pha4ge_row"] = np.frombuffer(bytes(RefPha4geRow.from_result(res)), np.uint8)
    exp["scalars_json"] = np.frombuffer(json.dumps(scalars).encode(), np.uint8)
    return exp


def random_hit_table(rng, db, genome, n) -> np.ndarray:
    """Adversarial hit tables that no aligner would emit: heavy overlaps, score/match/mapq ties, mapq 0/1/255."""
    hits = np.zeros(n, O.HIT_DTYPE)
    genes = np.sort(rng.integers(0, len(db.genes), size=n))
    glen = db.genes.lengths[genes]
    ctg = rng.integers(0, len(genome.contigs), size=n)
    clen = genome.contigs.lengths[ctg]
    span = np.minimum((glen * rng.uniform(0.05, 1.0, size=n)).astype(np.int64) + 1, np.minimum(glen, clen))
    hits["gene"], hits["contig"] = genes, ctg
    hits["q_start"] = (rng.random(n) * (glen - span + 1)).astype(np.int64)
    hits["q_end"] = hits["q_start"] + span
    # targets drawn from a handful of hot spots so that culling and clustering have work to do
    hot = (rng.integers(0, 6, size=n) * 0.15 * clen).astype(np.int64)
    hits["t_start"] = np.clip(hot + rng.integers(-300, 300, size=n), 0, clen - span)
    hits["t_end"] = hits["t_start"] + span
    hits["strand"] = rng.choice(np.array([1, -1], np.int8), size=n)
    hits["score"] = rng.choice(np.array([80, 120, 120, 500, 500, 900, 2000]), size=n)
    hits["matches"] = rng.choice(np.array([40, 60, 60, 250, 450]), size=n)
    hits["block_len"] = span
    hits["mapq"] = rng.choice(np.array([0, 1, 60, 60, 255], np.uint8), size=n)
    order = np.lexsort((-hits["score"], hits["gene"]))
    return hits[order]


def gen_typing() -> None:
    dbs = {
        "k": make_db("kpsc_k", seed=7, n_loci=9),
        "o": make_db("kpsc_o", seed=8),
    }
    for key, db in dbs.items():
        db.save(OUT / f"db_{key}.npz")
    small = dict(length=90_000, median_contigs=5, min_contig=200)
    cases = [
        ("k_plain1", "k", dict(seed=11, p_break=0, p_is=0, p_stop=0, **small)),
        ("k_plain2", "k", dict(seed=12, **small)),
        ("k_split", "k", dict(seed=13, force_split=True, p_is=0, **small)),
        ("k_nolocus", "k", dict(seed=14, locus=-1, **small)),
        ("k_is", "k", dict(seed=15, p_is=1.0, p_break=0, **small)),
        ("k_stop", "k", dict(seed=16, p_stop=1.0, p_break=0, p_is=0, **small)),
        ("k_divergent", "k", dict(seed=17, sub_rate=0.13, p_break=0, p_is=0, p_stop=0, **small)),
        ("k_second", "k", dict(seed=18, second_locus=3, locus=5, p_is=0, **small)),
        ("k_nrun", "k", dict(seed=19, n_run=40, p_break=0, p_is=0, p_stop=0, **small)),
        ("k_shredded", "k", dict(seed=20, length=90_000, median_contigs=60, min_contig=200, force_split=True)),
        # (round 5, kp-align v4) insertions and deletions of 33-450 bases inside genes: joined hits
        ("k_midindel_del", "k", dict(seed=21, p_break=0, p_is=0, p_stop=0, mid_indels=((60, "del"), (151, "del"), (300, "ins")), **small)),
        ("k_midindel_ins", "k", dict(seed=22, p_is=0, p_stop=0, mid_indels=((45, "ins"), (98, "ins"), (450, "del"), (36, "del")), **small)),
        # (round 6, kp-align v5) events 40-100 bases from a gene's start / end, pairs of events that nearly cancel, a storm of
        # events anywhere in the locus: chains with weak end pieces, pieces that share a diagonal, chains of three and more pieces
        ("k_endindel_start", "k", dict(seed=23, p_break=0, p_is=0, p_stop=0, sub_rate=0.01,
                                       placed_indels=((("del", 45, 40),), (("ins", 61, 70),), (("del", 150, 100),), (("ins", 200, 40),)), **small)),
        ("k_endindel_end", "k", dict(seed=24, p_break=0, p_is=0, p_stop=0, sub_rate=0.01,
                                     placed_indels=((("del", 45, -40),), (("ins", 61, -40),), (("del", 150, -70),), (("ins", 200, -100),)), **small)),
        ("k_interleave", "k", dict(seed=25, p_break=0, p_is=0, p_stop=0, sub_rate=0.01,
                                   placed_indels=((("del", 40, 200), ("ins", 50, 640)), (("ins", 60, 200), ("del", 45, 450)), (("del", 33, 200), ("ins", 36, 383))), **small)),
        ("k_storm", "k", dict(seed=26, p_is=0, p_stop=0, indel_storm=(6, 33, 300), **small)),
        ("o_plain", "o", dict(seed=31, p_extra=0.0, **small)),
        ("o_extra1", "o", dict(seed=32, p_extra=3.0, **small)),
        ("o_extra2", "o", dict(seed=33, p_extra=3.0, locus=0, **small)),
        ("o_extra3", "o", dict(seed=34, p_extra=3.0, locus=1, **small)),
    ]
    ref = {k: to_ref_db(db) for k, db in dbs.items()}
    typers = {k: RefSerotyper(r) for k, r in ref.items()}
    odbs = {k: O.OracleDB(*pack_sequences_flat(db.genes)) for k, db in dbs.items()}
    index = []
    for name, key, kw in cases:
        t0 = time.time()
        genome = make_assembly(dbs[key], name=name, **kw)
        hits, chain = odbs[key].align(genome.packed(), with_chain=True)
        exp = run_reference_typing(ref[key], typers[key], genome, hits)
        save_case(name, key, genome, hits, exp, chain)
        index.append(name)
        print(f"typing {name}: {len(hits)} hits -> {bytes(exp['kaptive_row'])[:90]!r} ({time.time() - t0:.1f}s)")
    rng = np.random.default_rng(99)
    for i, (key, n) in enumerate((("k", 0), ("k", 1), ("k", 60), ("k", 400), ("o", 150))):
        name = f"random_hits{i}"
        genome = make_assembly(dbs[key], seed=50 + i, name=name, length=30_000, median_contigs=4, locus=-1)
        hits = random_hit_table(rng, dbs[key], genome, n)
        t0 = time.time()
        exp = run_reference_typing(ref[key], typers[key], genome, hits)
        save_case(name, key, genome, hits, exp)
        index.append(name)
        print(f"typing {name}: {len(hits)} hits -> {bytes(exp['kaptive_row'])[:90]!r} ({time.time() - t0:.1f}s)")
    # variants of the confidence switches on one case
    genome = make_assembly(dbs["k"], name="k_divergent", **dict(cases[6][2]))
    hits, chain = odbs["k"].align(genome.packed(), with_chain=True)
    for tag, kwargs in (("loose", dict(max_other_genes=5, min_completeness=0.1, allow_below_threshold=True)),
                        ("strict", dict(max_other_genes=0, min_completeness=0.99, partial_edge_tolerance=50))):  # fmt: skip
        typer = RefSerotyper(ref["k"], **kwargs)
        exp = run_reference_typing(ref["k"], typer, genome, hits)
        exp["typer_kwargs_json"] = np.frombuffer(json.dumps(kwargs).encode(), np.uint8)
        save_case(f"k_divergent_{tag}", "k", genome, hits, exp, chain)
        index.append(f"k_divergent_{tag}")
    # Full-size cases (BASELINE.json configs 2 and 4 at their real shape): the inputs are DESCRIBED, not stored -- database
    # and assembly are regenerated from the seeds by kaptive_amd.synth (tests/golden_util.py) -- and everything expected is
    # what the reference's Serotyper returned for them here.
    for name, key, db_args, kw in FULL_SIZE_CASES:
        t0 = time.time()
        db = make_db(*db_args[:1], seed=db_args[1])
        genome = make_assembly(db, name=name, **kw)
        ref_db = to_ref_db(db)
        hits, chain = O.OracleDB(*pack_sequences_flat(db.genes)).align(genome.packed(), with_chain=True)
        exp = run_reference_typing(ref_db, RefSerotyper(ref_db), genome, hits)
        save_case(name, key, genome, hits, exp, chain, synth=dict(db=list(db_args), assembly=kw))
        index.append(name)
        print(f"typing {name}: {len(hits)} hits -> {bytes(exp['kaptive_row'])[:90]!r} ({time.time() - t0:.1f}s)")
    (OUT / "typing_index.json").write_text(json.dumps(index, indent=1) + "\n")


def gen_typing_full_missing() -> None:
    index = json.loads((OUT / "typing_index.json").read_text())
    for name, key, db_args, kw in FULL_SIZE_CASES:
        if (OUT / f"typing_{name}.npz").exists() and name in index:
            continue
        t0 = time.time()
        db = make_db(*db_args[:1], seed=db_args[1])
        genome = make_assembly(db, name=name, **kw)
        ref_db = to_ref_db(db)
        hits, chain = O.OracleDB(*pack_sequences_flat(db.genes)).align(genome.packed(), with_chain=True)
        exp = run_reference_typing(ref_db, RefSerotyper(ref_db), genome, hits)
        save_case(name, key, genome, hits, exp, chain, synth=dict(db=list(db_args), assembly=kw))
        if name not in index:
            index.append(name)
        print(f"typing {name}: {len(hits)} hits -> {bytes(exp['kaptive_row'])[:90]!r} ({time.time() - t0:.1f}s)")
    (OUT / "typing_index.json").write_text(json.dumps(index, indent=1) + "\n")


FULL_SIZE_CASES = [
    ("k_fullsize", "kfull", ("kpsc_k", 100), dict(seed=20_001)),                                                   # config 2: 5 Mbp, ~120 contigs
    ("ab_fullsize", "abfull", ("ab_k", 102), dict(seed=40_001, length=4.0e6, median_contigs=1500, min_contig=200, force_split=True)),  # config 4
    # (round 4, later) the shapes the parity sweep varies, each once through the reference's own reduction at full size
    ("k_full_div9", "kfull", ("kpsc_k", 100), dict(seed=20_002, sub_rate=0.09)),
    ("k_full_indel", "kfull", ("kpsc_k", 100), dict(seed=20_003, indel_rate=2e-3)),
    ("k_full_nrun_second", "kfull", ("kpsc_k", 100), dict(seed=20_004, n_run=3, second_locus=17)),
    ("k_full_tandem", "kfull", ("kpsc_k", 100), dict(seed=20_005, tandem_gene=1)),
    ("o_fullsize", "ofull", ("kpsc_o", 101), dict(seed=20_006)),                                                   # config 3's O half
    ("ab_full_div6", "abfull", ("ab_k", 102), dict(seed=40_002, length=4.0e6, median_contigs=1500, min_contig=200, force_split=True, sub_rate=0.06)),
    # (round 5, kp-align v4) mid-size insertions and deletions inside genes at full size
    ("k_full_midindel", "kfull", ("kpsc_k", 100), dict(seed=20_007, mid_indels=((48, "del"), (100, "ins"), (301, "del"), (450, "ins")))),
    ("ab_full_midindel", "abfull", ("ab_k", 102), dict(seed=40_003, length=4.0e6, median_contigs=1500, min_contig=200, force_split=True,
                                                      mid_indels=((33, "ins"), (64, "del"), (150, "ins"), (300, "del")))),
]


def save_case(name, key, genome, hits, exp, chain=None, synth=None) -> None:
    """``chain``: the chain score behind every hit (aligner tables only) -- an input of the hit's mapq that the record does
    not keep; the hit-finalisation test needs it to recompute the recorded mapq."""
    inputs = dict(contig_ids=np.array(list(genome.contigs.ids), dtype="U"), contig_seqs=genome.contigs.seqs,
                  contig_lengths=genome.contigs.lengths)
    if synth is not None:  # regenerated from seeds by the loader; a digest of the sequence text guards the generator
        import hashlib

        inputs = dict(synth_json=np.frombuffer(json.dumps(synth).encode(), np.uint8),
                      contig_sha1=np.array(hashlib.sha1(genome.contigs.seqs.tobytes()).hexdigest()))
    np.savez_compressed(
        OUT / f"typing_{name}.npz",
        db_key=np.array(key), genome_id=np.array(genome.id), hits=hits, **inputs,
        hit_chain_scores=np.zeros(len(hits), np.int32) if chain is None else np.asarray(chain, np.int32),
        **{f"exp.{k}": v for k, v in exp.items()},
    )  # fmt: skip


def gen_genbank() -> None:
    """tests/golden/db_genbank_expected.npz: what the REFERENCE's Database.from_genbank (src/kaptive/db/core.py:289-507)
    compiles from the hand-written tests/golden/handwritten_db.gbk + .toml.  The gb-io wheel is absent here, so the
    flat file reaches the reference through oracle/refshim/gb_io (this repository's reader behind gb-io's record
    surface): every derived array -- ids, positions, vocabularies, intervals, gene and protein sequences, phenotype
    masks -- is the reference's own output."""
    db = RefDatabase.from_genbank(OUT / "handwritten_db.gbk")
    ph = db.phenotypes
    np.savez_compressed(
        OUT / "db_genbank_expected.npz",
        locus_ids=np.array(db.loci.ids), serotypes=np.array(db.serotypes), gene_ids=np.array(db.genes.ids),
        loci_seqs=db.loci.seqs, loci_offsets=db.loci.offsets, loci_lengths=db.loci.lengths,
        gene_seqs=db.genes.seqs, gene_offsets=db.genes.offsets, gene_lengths=db.genes.lengths,
        prot_seqs=db.translations.seqs, prot_offsets=db.translations.offsets, prot_lengths=db.translations.lengths,
        gene_starts=db.gene_intervals.starts, gene_ends=db.gene_intervals.ends, gene_strands=db.gene_intervals.strands,
        gene_positions=db.gene_positions, extra_genes=db.extra_genes, gene_locus_indices=db.gene_locus_indices,
        locus_gene_offsets=db.locus_gene_offsets, locus_gene_lengths=db.locus_gene_lengths,
        gene_cluster_ids=db.gene_cluster_ids, gene_description_ids=db.gene_description_ids,
        cluster_keys=np.array(db.cluster_keys), description_keys=np.array(db.description_keys),
        max_locus_length=np.int64(db.max_locus_length), id_threshold=np.float64(db.metadata.id_threshold),
        pheno_ids=np.array(ph.ids), pheno_locus_masks=ph.locus_masks, pheno_extra_masks=ph.extra_masks,
        pheno_inactive_masks=ph.inactive_masks, pheno_extra_counts=ph.extra_counts, pheno_priorities=ph.priorities,
        pheno_as_suffix=ph.as_suffix,
    )  # fmt: skip
    print(f"genbank: {len(db.loci)} loci, {len(db.genes)} genes, {len(ph)} phenotype rules")


def gen_compare() -> None:
    """tests/golden/compare.npz: the REFERENCE's compare.LocusComparator (src/kaptive/compare.py:195-396) -- randstrobe
    records and top-hit seeds (src/kaptive/core/kmers.py), seeded protein alignments (pairwise.py align_seeds) and the
    normalised gene coordinates -- on five loci of a synthetic K database (proteins truncated to 200
    residues so that the pure-Python fallback of the numba kernels finishes in minutes).  Locus 2 is given as two pieces
    on two contigs, the second on the reverse strand, with byte-string descriptions and gene states."""
    from kaptive.compare import LocusComparator as RefComparator
    from kaptive.compare import LocusData as RefLocusData
    from kaptive.core.kmers import RandstrobeIndex as RefRandstrobeIndex
    from kaptive.serotyping.models import LocusPieces as RefLocusPieces

    n_loci = 5
    db = make_db("kpsc_k", seed=11, n_loci=n_loci)
    inputs, raw = [], {}
    for li in range(n_loci):
        g0, n = int(db.locus_gene_offsets[li]), int(db.locus_gene_lengths[li])
        ids, chunks = [], []
        for g in range(g0, g0 + n):
            o, ln = int(db.translations.offsets[g]), int(db.translations.lengths[g])
            chunks.append(db.translations.seqs[o : o + min(ln, 200)])
            ids.append(db.genes.ids[g])
        lengths = np.array([len(c) for c in chunks], np.int32)
        offsets = np.zeros(n, np.int32)
        np.cumsum(lengths[:-1], out=offsets[1:])
        seqs = np.concatenate(chunks)
        starts = db.gene_intervals.starts[g0 : g0 + n].astype(np.int32) + 1000 * li
        ends = db.gene_intervals.ends[g0 : g0 + n].astype(np.int32) + 1000 * li
        strands = db.gene_intervals.strands[g0 : g0 + n].astype(np.int8)
        raw[f"l{li}.ids"], raw[f"l{li}.seqs"], raw[f"l{li}.offsets"], raw[f"l{li}.lengths"] = np.array(ids), seqs, offsets, lengths
        raw[f"l{li}.starts"], raw[f"l{li}.ends"], raw[f"l{li}.strands"] = starts, ends, strands
        kw = {}
        if li == 2:  # fragmented: genes 0..4 on contig 3, the rest on contig 7 (reverse strand)
            cut = int(ends[4]) + 10
            kw["pieces"] = RefLocusPieces(np.array([3, 7], np.uint32), np.array([int(starts[0]) - 5, cut], np.int32),
                                          np.array([cut, int(ends[-1]) + 20], np.int32), np.array([1, -1], np.int8))
            kw["gene_ctg_indices"] = np.array([3] * 5 + [7] * (n - 5), np.uint32)
            kw["gene_states"] = (np.arange(n) % 4).astype(np.int8)
            kw["gene_descriptions"] = np.array([f"product {x}".encode() for x in range(n)], dtype="S64")
            for key in ("gene_ctg_indices", "gene_states", "gene_descriptions"):
                raw[f"l{li}.{key}"] = kw[key]
            raw["l2.piece_ctg"], raw["l2.piece_starts"], raw["l2.piece_ends"], raw["l2.piece_strands"] = (
                kw["pieces"].ctg_indices, kw["pieces"].starts, kw["pieces"].ends, kw["pieces"].strands)
        inputs.append(RefLocusData(proteins=ref_sequences(ids, seqs, offsets, lengths), name=db.loci.ids[li],
                                   backbone=RefIntervals(starts, ends, strands), **kw))
    t0 = time.time()
    for li, inp in enumerate(inputs):
        raw[f"l{li}.records"] = RefRandstrobeIndex.build(inp.proteins, k=10, s=5, sort_by_hash=False).records
        raw[f"l{li}.records_sorted"] = RefRandstrobeIndex.build(inp.proteins, k=10, s=5, sort_by_hash=True).records
    for i in range(n_loci):
        for j in range(i + 1, n_loci):
            t = RefRandstrobeIndex.build(inputs[j].proteins, k=10, s=5, sort_by_hash=True)
            q = RefRandstrobeIndex.build(inputs[i].proteins, k=10, s=5, sort_by_hash=False)
            sd = t.top_hits(q, min_score=1)
            raw[f"seeds.{i}.{j}"] = np.stack([sd.query_indices.astype(np.int64), sd.target_indices.astype(np.int64),
                                              sd.scores.astype(np.int64), sd.offsets.astype(np.int64)])
    res = RefComparator()(inputs)
    e = res.edges
    for col in ("query_locus_indices", "target_locus_indices", "query_indices", "target_indices", "global_query_indices",
                "global_target_indices"):
        raw[f"edges.{col}"] = getattr(e, col)
    raw["edges.alignments"] = np.stack([getattr(e.alignments, c) for c in
                                        ("scores", "matches", "mismatches", "gaps", "q_starts", "q_ends", "t_starts", "t_ends")], axis=1)
    raw["locus_names"], raw["locus_lengths"], raw["locus_offsets"] = np.array(res.locus_names), res.locus_lengths, res.locus_offsets
    raw["gene_names"] = np.array([str(x) for x in res.gene_names])
    raw["gene_descriptions"] = np.array([str(x) for x in res.gene_descriptions])
    raw["gene_states"] = res.gene_states
    gi = res.gene_intervals
    raw["gi.starts"], raw["gi.ends"], raw["gi.strands"], raw["gi.original_indices"] = gi.starts, gi.ends, gi.strands, gi.original_indices
    np.savez_compressed(OUT / "compare.npz", **raw)
    print(f"compare: {len(e)} edges between {n_loci} loci ({time.time() - t0:.0f} s)")


def main() -> None:
    OUT.mkdir(parents=True, exist_ok=True)
    what = set(sys.argv[1:]) or {"protein", "intervals", "seqs", "typing", "genbank", "compare"}
    if "compare" in what:
        gen_compare()
    if "genbank" in what:
        gen_genbank()
    if "protein" in what:
        gen_protein_dp(np.random.default_rng(1))
    if "intervals" in what:
        gen_intervals(np.random.default_rng(2))
    if "seqs" in what:
        gen_seqs(np.random.default_rng(3))
    if "typing" in what:
        gen_typing()
    if "typing_full" in what:  # only the full-size cases that are not there yet (the index gains their names)
        gen_typing_full_missing()


if __name__ == "__main__":
    main()
