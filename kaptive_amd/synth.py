"""Synthetic databases and assemblies shaped like the benchmark configs.

No curated database or genome exists in the build container, so tests and bench use these
generators; every draw comes from one ``numpy.random.default_rng(seed)``.

* ``make_db("kpsc_k")``  163 loci x 16-30 CDS (600-1800 bp), 8 gene families shared by all loci at 85-99 % identity.
* ``make_db("kpsc_o")``  13 loci x 6-10 CDS + 10 extra genes + 12 ``extra_genes`` phenotype rules.
* ``make_db("ab_k")``    240 loci x 18-28 CDS.
* ``make_assembly``      iid background (GC set per organism) with one mutated locus copy, cut into contigs.
"""

from __future__ import annotations

import numpy as np

from kaptive_amd.core.genome import GenomeAssembly
from kaptive_amd.core.seq import COMP_MAP, SeqRecord, Sequences
from kaptive_amd.db import Database

_ACGT = np.frombuffer(b"ACGT", np.uint8)
_STOPS = {b"TAA", b"TAG", b"TGA"}
_SHARED = ("galF", "cpsACP", "wzi", "wza", "wzb", "wzc", "gnd", "ugd")

DB_SHAPES = {
    "kpsc_k": dict(n_loci=163, genes=(16, 30), shared=_SHARED, gc=0.57, n_extra=0, n_rules=0, prefix="KL",
                   type_prefix="K", organism="Klebsiella pneumoniae species complex", taxon=573, antigen="K"),
    "kpsc_o": dict(n_loci=13, genes=(6, 10), shared=("wzm", "wzt"), gc=0.57, n_extra=10, n_rules=12, prefix="OL",
                   type_prefix="O", organism="Klebsiella pneumoniae species complex", taxon=573, antigen="O"),
    "ab_k": dict(n_loci=240, genes=(18, 28), shared=("fkpA", "wzc", "wzb", "wza", "gna", "galU", "ugd", "gpi"),
                 gc=0.39, n_extra=0, n_rules=0, prefix="KL", type_prefix="K", organism="Acinetobacter baumannii",
                 taxon=470, antigen="K"),
}  # fmt: skip


def _base_probs(gc: float) -> np.ndarray:
    return np.array([(1 - gc) / 2, gc / 2, gc / 2, (1 - gc) / 2])


def random_dna(rng: np.random.Generator, n: int, gc: float) -> np.ndarray:
    return _ACGT[rng.choice(4, size=n, p=_base_probs(gc))]


def _kill_internal_stops(orf: np.ndarray) -> None:
    """In place: any in-frame stop before the last codon gets its first base set to C."""
    body = orf[: len(orf) - 3].reshape(-1, 3)
    t = body[:, 0] == ord("T")
    stop = t & (
        ((body[:, 1] == ord("A")) & ((body[:, 2] == ord("A")) | (body[:, 2] == ord("G"))))
        | ((body[:, 1] == ord("G")) & (body[:, 2] == ord("A")))
    )
    body[stop, 0] = ord("C")


def random_orf(rng: np.random.Generator, n_bp: int, gc: float) -> np.ndarray:
    n_bp -= n_bp % 3
    orf = random_dna(rng, n_bp, gc)
    orf[:3] = np.frombuffer(b"ATG", np.uint8)
    orf[-3:] = np.frombuffer((b"TAA", b"TAG", b"TGA")[rng.integers(3)], np.uint8)
    _kill_internal_stops(orf)
    return orf


def mutate(rng: np.random.Generator, seq: np.ndarray, sub_rate: float, indel_rate: float = 0.0) -> np.ndarray:
    out = seq.copy()
    hit = np.flatnonzero(rng.random(len(out)) < sub_rate)
    # a substitution always changes the base: add 1..3 in ACGT index space
    idx = np.searchsorted(_ACGT, out[hit])
    out[hit] = _ACGT[(idx + rng.integers(1, 4, size=len(hit))) % 4]
    if indel_rate > 0:
        sites = np.flatnonzero(rng.random(len(out)) < indel_rate)
        for s in sites[::-1]:
            k = int(rng.integers(1, 4))
            if rng.random() < 0.5:
                out = np.delete(out, slice(s, s + k))
            else:
                out = np.insert(out, s, random_dna(rng, k, 0.5))
    return out


def revcomp(seq: np.ndarray) -> np.ndarray:
    return COMP_MAP[seq[::-1]]


def make_db(kind: str = "kpsc_k", seed: int = 100, n_loci: int | None = None, id_threshold: float = 82.5) -> Database:
    shape = DB_SHAPES[kind]
    rng = np.random.default_rng(seed)
    gc = shape["gc"]
    n_loci = n_loci or shape["n_loci"]
    families = {name: random_orf(rng, int(rng.integers(600, 1801)), gc) for name in shape["shared"]}
    loci = []
    for li in range(n_loci):
        name = f"{shape['prefix']}{li + 1}"
        n_genes = int(rng.integers(shape["genes"][0], shape["genes"][1] + 1))
        # shared families take the outer positions (first half up front, rest at the end), unique genes in between
        fam = list(shape["shared"])
        lead, tail = fam[: (len(fam) + 1) // 2], fam[(len(fam) + 1) // 2 :]
        n_unique = max(n_genes - len(fam), 2)
        order = lead + [None] * n_unique + tail
        parts, genes = [random_dna(rng, int(rng.integers(50, 200)), gc)], []
        pos = len(parts[0])
        for gi, fam_name in enumerate(order):
            if fam_name is None:
                orf = random_orf(rng, int(rng.integers(600, 1801)), gc)
                gname = f"wc{chr(97 + li % 26)}{chr(97 + gi % 26)}{li // 26}"
                product = f"hypothetical protein {li}-{gi}"
            else:
                orf = mutate(rng, families[fam_name], float(rng.uniform(0.01, 0.15)))
                orf[:3] = families[fam_name][:3]
                orf[-3:] = families[fam_name][-3:]
                _kill_internal_stops(orf)
                gname, product = fam_name, f"{fam_name} family protein"
            strand = 1 if (fam_name is not None or rng.random() < 0.8) else -1
            parts.append(orf if strand == 1 else revcomp(orf))
            genes.append(dict(start=pos, end=pos + len(orf), strand=strand, gene=gname, product=product))
            pos += len(orf)
            spacer = random_dna(rng, int(rng.integers(30, 151)), gc)
            parts.append(spacer)
            pos += len(spacer)
        loci.append(dict(name=name, type=f"{shape['type_prefix']}{li + 1}", extra=False,
                         seq=np.concatenate(parts).tobytes(), genes=genes))  # fmt: skip
    logic = {}
    if shape["n_extra"]:
        parts, genes, pos = [], [], 0
        for ei in range(shape["n_extra"]):
            orf = random_orf(rng, int(rng.integers(600, 1501)), gc)
            parts += [orf, random_dna(rng, 60, gc)]
            genes.append(dict(start=pos, end=pos + len(orf), strand=1, gene=f"opx{ei}", product=f"modifier {ei}"))
            pos += len(orf) + 60
        loci.append(dict(name="Extra_genes", type="", extra=True, seq=np.concatenate(parts).tobytes(), genes=genes))
        for ri in range(shape["n_rules"]):
            li = ri % n_loci
            logic[f"{shape['type_prefix']}{li + 1}v{ri}"] = dict(
                loci=[f"{shape['prefix']}{li + 1}"], extra_genes=[f"opx{ri % shape['n_extra']}"], priority=50 + ri
            )
    meta = dict(
        name=f"synthetic {kind}", keyword=kind, genbank=f"{kind}.gbk", organism=shape["organism"],
        taxon=shape["taxon"], antigen=shape["antigen"], pathway="Wzx/Wzy", version=f"synth-{seed}",
        id_threshold=id_threshold, doi=[], owner="kaptive_amd", repo="synthetic", branch="main", contact={},
        phenotype_logic=logic,
    )  # fmt: skip
    return Database.from_parts(meta, loci)


def make_assembly(
    db: Database,
    seed: int,
    length: float = 5.0e6,
    gc: float | None = None,
    median_contigs: int = 120,
    locus: int | None = None,
    min_contig: int = 200,
    p_break: float = 0.3,
    p_is: float = 0.05,
    p_stop: float = 0.05,
    p_extra: float = 0.5,
    force_split: bool = False,
    name: str | None = None,
    sub_rate: float | None = None,
    second_locus: int | None = None,
    n_run: int = 0,
    also: tuple = (),
    tandem_gene: int = 0,
    indel_rate: float = 2e-5,
    mid_indels: tuple = (),
    indel_storm: tuple = (),
    placed_indels: tuple = (),
    repeat_segment: tuple = (),
    is_copies: tuple = (),
    background: str = "iid",
) -> GenomeAssembly:
    """One synthetic assembly holding a mutated copy of one database locus.
    ``locus`` < 0 plants no locus at all.  ``mid_indels``: (size, "del" | "ins") pairs, each planted inside a gene of
    the locus copy of its own (a deletion of ``size`` bases or an insertion of ``size`` random ones somewhere in the
    gene's middle half) -- the 30-500 base events minimap2 chains across (bw = 500); drawn from a generator of their
    own, so that the other draws of a seed do not move.  ``indel_storm`` = (events, smallest, largest): that many insertions /
    deletions of smallest..largest bases ANYWHERE in the locus copy (several per gene, gene ends, spacers: chains of
    three and more pieces, weak end pieces, events closer together than a band is wide).  ``placed_indels``: one entry per gene
    to edit, each a tuple of events ``(kind, size, offset)`` at an EXPLICIT place -- ``offset`` >= 0 counts from the gene's first
    base in locus coordinates, ``offset`` < 0 from its end (a deletion then ends, an insertion sits, ``-offset`` bases before
    the end) -- so that events next to a gene's ends and pairs of events that nearly cancel can be planted (``mid_indels`` only
    reaches the middle half of a gene); genes long enough for their entry are drawn from a generator of their own.  ``repeat_segment`` = (length, copies): a stretch of one gene of
    the locus copy planted ``copies`` more times around the genome, each copy mutated a little (seeds that occur more
    than ten times: minimap2's occurrence cut); ``is_copies`` = (copies, length): one random IS-like element planted that
    many times (not in any database: it only moves the quantile minimap2 derives its cut from).  ``background`` = "paralog":
    the iid background also carries what real genomes hold beside the locus -- 40 diverged relatives (70-92 % identity,
    small indels) of database genes, half of them of the conserved families, 5-30 copies of two IS-like elements and seven
    copies of an rRNA-like 5 kb operon -- so that chance seeds, weak chains and repeats exist outside the planted locus."""
    rng = np.random.default_rng(seed)
    gc = DB_SHAPES.get(db.metadata.keyword, {}).get("gc", 0.5) if gc is None else gc
    total = int(length * rng.uniform(0.95, 1.05))
    typed = np.flatnonzero(~_locus_is_extra(db))
    li = int(rng.choice(typed)) if locus is None else locus
    genome = random_dna(rng, total, gc)
    cuts_inside: list[int] = []
    if li >= 0:
        o, n = int(db.loci.offsets[li]), int(db.loci.lengths[li])
        rate = float(rng.uniform(0.0, 0.03)) if sub_rate is None else sub_rate
        copy = mutate(rng, db.loci.seqs[o : o + n], rate)
        g0, g1 = int(db.locus_gene_offsets[li]), int(db.locus_gene_offsets[li] + db.locus_gene_lengths[li])
        for gi in range(g0, g1):  # substitutions that would create an in-frame stop are purged, as selection does
            s, e = int(db.gene_intervals.starts[gi]), int(db.gene_intervals.ends[gi])
            if db.gene_intervals.strands[gi] > 0:
                _kill_internal_stops(copy[s:e])
            else:
                orf = revcomp(copy[s:e])
                _kill_internal_stops(orf)
                copy[s:e] = revcomp(orf)
        if rng.random() < p_stop:  # premature stop in one gene (planted on the unmutated coordinates; close enough)
            gi = int(rng.integers(g0, g1))
            s, e = int(db.gene_intervals.starts[gi]), int(db.gene_intervals.ends[gi])
            at = s + 3 * int(rng.integers(5, max(6, (e - s) // 6)))
            if at + 3 < len(copy):
                stop = np.frombuffer(b"TAA", np.uint8)
                copy[at : at + 3] = stop if db.gene_intervals.strands[gi] > 0 else revcomp(stop)
        copy = mutate(rng, copy, 0.0, indel_rate=indel_rate)
        if mid_indels:
            rng2 = np.random.default_rng([seed, 0x1DE1])
            genes = rng2.choice(np.arange(g0, g1), size=min(len(mid_indels), g1 - g0), replace=False)
            edits = []
            for gi, (size, kind) in zip(genes, mid_indels):
                s, e = int(db.gene_intervals.starts[gi]), int(db.gene_intervals.ends[gi])
                at = s + int(rng2.integers((e - s) // 4, max((e - s) // 4 + 1, 3 * (e - s) // 4 - (size if kind == "del" else 0))))
                edits.append((at, int(size), kind, random_dna(rng2, int(size), gc)))
            for at, size, kind, ins in sorted(edits, key=lambda t: -t[0]):  # from the far end: earlier coordinates stay put
                copy = np.delete(copy, slice(at, at + size)) if kind == "del" else np.concatenate([copy[:at], ins, copy[at:]])
        if placed_indels:
            rng5 = np.random.default_rng([seed, 0xE2D])
            glen = db.gene_intervals.ends[g0:g1].astype(np.int64) - db.gene_intervals.starts[g0:g1]
            taken: set = set()
            edits = []
            for entry in placed_indels:
                need = 60 + max(abs(int(off)) + (int(size) if kind == "del" else 0) for kind, size, off in entry)
                ok = [g0 + i for i in range(g1 - g0) if glen[i] >= need and g0 + i not in taken]
                if not ok:
                    continue
                gi = int(rng5.choice(ok))
                taken.add(gi)
                s, e = int(db.gene_intervals.starts[gi]), int(db.gene_intervals.ends[gi])
                for kind, size, off in entry:
                    size, off = int(size), int(off)
                    at = s + off if off >= 0 else (e + off - size if kind == "del" else e + off)
                    edits.append((at, size, kind, random_dna(rng5, size, gc)))
            for at, size, kind, ins in sorted(edits, key=lambda t: -t[0]):
                copy = np.delete(copy, slice(at, at + size)) if kind == "del" else np.concatenate([copy[:at], ins, copy[at:]])
        if indel_storm:
            rng4 = np.random.default_rng([seed, 0x5702])
            n_ev, lo_size, hi_size = indel_storm
            for at in sorted((int(x) for x in rng4.integers(200, len(copy) - 200 - hi_size, size=n_ev)), reverse=True):
                size = int(rng4.integers(lo_size, hi_size + 1))
                if rng4.random() < 0.5:
                    copy = np.delete(copy, slice(at, at + size))
                else:
                    copy = np.concatenate([copy[:at], random_dna(rng4, size, gc), copy[at:]])
        if tandem_gene:  # `tandem_gene` extra copies of one gene, head to tail, each mutated on its own (multi-copy stress)
            gi = int(rng.integers(g0, g1))
            s, e = int(db.gene_intervals.starts[gi]), min(int(db.gene_intervals.ends[gi]), len(copy))
            unit = copy[s:e]
            reps = [mutate(rng, unit, 0.01) for _ in range(tandem_gene)]
            copy = np.concatenate([copy[:e], *reps, copy[e:]])
        if rng.random() < p_is:  # IS-like 1.2 kb insertion inside the locus
            at = int(rng.integers(len(copy) // 4, 3 * len(copy) // 4))
            copy = np.concatenate([copy[:at], random_dna(rng, 1200, 0.5), copy[at:]])
        if n_run:  # a scaffold gap inside the locus
            at = int(rng.integers(len(copy) // 4, 3 * len(copy) // 4))
            copy[at : at + n_run] = ord("N")
        if rng.random() < 0.5:
            copy = revcomp(copy)
        where = int(rng.integers(total // 10, total - total // 10 - len(copy)))
        genome[where : where + len(copy)] = copy
        if repeat_segment or is_copies or background == "paralog":
            rng3 = np.random.default_rng([seed, 0x0CC])
            free_lo, free_hi = total // 10, total - total // 8  # (the last twelfth belongs to `also`; the locus is avoided below)

            def plant(unit):
                for _ in range(50):
                    at = int(rng3.integers(free_lo, free_hi - len(unit)))
                    if at + len(unit) < where - 3000 or at > where + len(copy) + 3000:
                        genome[at : at + len(unit)] = unit if rng3.random() < 0.5 else revcomp(unit)
                        return

            if repeat_segment:
                seg_len, copies = repeat_segment
                gi = int(rng3.integers(g0, g1))
                s0, e0 = int(db.gene_intervals.starts[gi]), int(db.gene_intervals.ends[gi])
                a0 = o + s0 + int(rng3.integers(0, max(1, e0 - s0 - seg_len)))
                unit = db.loci.seqs[a0 : a0 + seg_len]
                for _ in range(copies):
                    plant(mutate(rng3, unit, 0.005))
            if is_copies:
                copies, is_len = is_copies
                element = random_dna(rng3, is_len, 0.5)
                for _ in range(copies):
                    plant(mutate(rng3, element, 0.002))
            if background == "paralog":
                rng4 = np.random.default_rng([seed, 0xBA6])
                shared = set(DB_SHAPES.get(db.metadata.keyword, {}).get("shared", ()))
                ids = [str(i) for i in db.genes.ids]
                fam = np.array([i for i, name in enumerate(ids) if name.rsplit("_", 1)[-1] in shared], np.int64)
                picks = np.r_[rng4.choice(fam, size=20) if len(fam) else [], rng4.integers(0, len(ids), size=20)].astype(np.int64)
                for gi in picks:  # diverged relatives of database genes
                    go, gn = int(db.genes.offsets[gi]), int(db.genes.lengths[gi])
                    plant(mutate(rng4, db.genes.seqs[go : go + gn], float(rng4.uniform(0.08, 0.30)), indel_rate=0.004))
                for _ in range(2):  # IS-like elements
                    element = random_dna(rng4, int(rng4.integers(800, 1600)), 0.5)
                    for _ in range(int(rng4.integers(5, 31))):
                        plant(mutate(rng4, element, 0.002))
                operon = random_dna(rng4, 5000, 0.5)
                for _ in range(7):
                    plant(mutate(rng4, operon, 0.001))
        if second_locus is not None:  # part of another locus elsewhere in the genome
            o2, n2 = int(db.loci.offsets[second_locus]), int(db.loci.lengths[second_locus])
            other = mutate(rng, db.loci.seqs[o2 : o2 + n2 // 2], 0.01)
            at = where - len(other) - 2000
            if at < 0:
                at = where + len(copy) + 2000
            genome[at : at + len(other)] = other[: max(0, total - at)]
        if force_split or rng.random() < p_break:
            cuts_inside.append(where + int(rng.integers(len(copy) // 5, 4 * len(copy) // 5)))
    for k, other in enumerate(also):  # one locus of each further database (e.g. the O locus next to the K locus)
        typed2 = np.flatnonzero(~_locus_is_extra(other))
        lj = int(rng.choice(typed2))
        o2, n2 = int(other.loci.offsets[lj]), int(other.loci.lengths[lj])
        copy2 = mutate(rng, other.loci.seqs[o2 : o2 + n2], float(rng.uniform(0.0, 0.03)))
        g0, g1 = int(other.locus_gene_offsets[lj]), int(other.locus_gene_offsets[lj] + other.locus_gene_lengths[lj])
        for gi in range(g0, g1):
            s, e = int(other.gene_intervals.starts[gi]), int(other.gene_intervals.ends[gi])
            if other.gene_intervals.strands[gi] > 0:
                _kill_internal_stops(copy2[s:e])
            else:
                orf = revcomp(copy2[s:e])
                _kill_internal_stops(orf)
                copy2[s:e] = revcomp(orf)
        if rng.random() < 0.5:
            copy2 = revcomp(copy2)
        at = total - total // 12 + k * 60_000  # the last twelfth of the genome is never used by the main locus
        if at + len(copy2) < total:
            genome[at : at + len(copy2)] = copy2
    extra_rows = np.flatnonzero(db.extra_genes)
    for gi in extra_rows:  # unlinked modifier genes planted elsewhere
        if rng.random() < p_extra / max(len(extra_rows), 1) * 3:
            o, n = int(db.genes.offsets[gi]), int(db.genes.lengths[gi])
            g = mutate(rng, db.genes.seqs[o : o + n], 0.01)
            at = int(rng.integers(0, total // 12))
            genome[at : at + len(g)] = g if rng.random() < 0.5 else revcomp(g)
    # contig lengths: lognormal around total/median_contigs, floored at min_contig
    n_ctg = max(1, int(rng.lognormal(np.log(median_contigs), 0.3)))
    w = rng.lognormal(0.0, 1.0, size=n_ctg)
    cuts = np.unique(np.r_[(np.cumsum(w)[:-1] / w.sum() * total).astype(np.int64), np.array(cuts_inside, np.int64)])
    cuts = cuts[(cuts > 0) & (cuts < total)]
    bounds = np.r_[0, cuts, total]
    keep = np.flatnonzero(np.diff(bounds) >= min_contig)
    recs = [SeqRecord(f"contig_{k + 1}", genome[bounds[i] : bounds[i + 1]].tobytes()) for k, i in enumerate(keep)]
    if rng.random() < 0.3 and len(recs) > 1:  # some contigs come out reverse-complemented, as assemblers do
        for k in rng.choice(len(recs), size=max(1, len(recs) // 3), replace=False):
            r = recs[k]
            recs[k] = SeqRecord(r.id, revcomp(np.frombuffer(r.seq, np.uint8)).tobytes())
    return GenomeAssembly(name or f"asm_{seed}", Sequences.from_records(recs))


def _locus_is_extra(db: Database) -> np.ndarray:
    return db.extra_genes[db.locus_gene_offsets.astype(np.int64)]
