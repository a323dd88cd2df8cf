"""Host side of the batched GPU reduction: record layouts, the float steps that stay in numpy, and result assembly.

The device returns, per assembly, a summary, the kept hits (culled, clustered, translated, protein-aligned, with state
and flags) and the locus pieces (csrc/kp_reduce_core.h).  Three things are finished here, in numpy, because the
reference's own numpy calls are the only way to reproduce their bits exactly on the same machine:

* ``choose_best_loci``  -- ``locus_scores * (counts / expected) ** 3`` and ``argmax`` (core.py:200-206; float32 ``**``
  goes through the platform's ``powf``);
* piece order            -- ``np.argsort`` of the mean expected positions (core.py:281; unstable kind);
* mean identity          -- ``np.mean`` of a float32 array (core.py:395-396; pairwise float32 summation).

Everything else in ``assemble`` is indexing and the shared ``Serotyper.finish`` tail.
"""

from __future__ import annotations

import numpy as np

from kaptive_amd.core.genome import GenomeAssembly
from kaptive_amd.core.seq import Sequences
from kaptive_amd.serotyping.models import LocusPieces, SerotypingResult

F_EXPECTED, F_INSIDE, F_EXTRA, F_PARTIAL, F_SPURIOUS, F_PRIMARY = 1, 2, 4, 8, 16, 32
MAX_LOCUS_GENES = 256

KEPT_DTYPE = np.dtype(
    [("gene", "<i4"), ("contig", "<i4"), ("q_start", "<i4"), ("q_end", "<i4"), ("t_start", "<i4"), ("t_end", "<i4"),
     ("score", "<i4"), ("prot_off", "<i4"), ("prot_len", "<i4"), ("cluster", "<i4"), ("dp", "<i4", 8),
     ("pident", "<f4"), ("coverage", "<f4"), ("strand", "i1"), ("state", "i1"), ("flags", "u1"), ("pad", "u1")]
)  # fmt: skip
PIECE_DTYPE = np.dtype([("contig", "<i4"), ("start", "<i4"), ("end", "<i4"), ("strand", "<i4"), ("mean_pos", "<f8")])
SUMMARY_DTYPE = np.dtype(
    [("n_hits", "<i4"), ("n_kept", "<i4"), ("n_final", "<i4"), ("n_pieces", "<i4"), ("best_locus", "<i4"),
     ("n_expected", "<i4"), ("n_missing", "<i4"), ("overflow", "<i4"), ("missing_mask", "<u8", MAX_LOCUS_GENES // 64),
     ("ident_sum", "<f4"), ("n_normal", "<i4")]
)  # fmt: skip
assert KEPT_DTYPE.itemsize == 84 and PIECE_DTYPE.itemsize == 24 and SUMMARY_DTYPE.itemsize == 72


def choose_best_loci(locus_scores: np.ndarray, locus_counts: np.ndarray, expected_per_locus: np.ndarray):
    """Batched form of core.py:196-206.  ``locus_scores`` f64 [n_asm, n_loci], ``locus_counts`` int [n_asm, n_loci];
    returns (best locus per assembly, penalised scores, completeness)."""
    completeness = locus_counts.astype(np.float32) / expected_per_locus  # float32 / float32
    final = locus_scores * (completeness**3)
    return np.argmax(final, axis=1).astype(np.int32), final, completeness


def assemble(typer, genome_id: str, summary: np.void, kept: np.ndarray, pieces: np.ndarray, best_score: float,
             genome: GenomeAssembly | None = None) -> SerotypingResult:  # fmt: skip
    """One assembly's device records -> SerotypingResult.  With ``genome`` the three sequence collections are extracted
    on the host exactly as the single-genome path does; without it they carry ids and zero-length sequences (enough
    for the TSV rows, which only read ids)."""
    db = typer._db
    best = int(summary["best_locus"])
    k = kept[(kept["flags"] & F_SPURIOUS) == 0]
    g = k["gene"]
    flags = k["flags"]
    hits = typer.gene_hits_table(
        g, k["q_start"], k["q_end"], k["contig"].astype(np.uint32), k["t_start"], k["t_end"], k["strand"],
        (flags & F_EXPECTED) != 0, (flags & F_INSIDE) != 0, (flags & F_EXTRA) != 0, k["coverage"].copy(),
    )  # fmt: skip
    order = np.argsort(np.ascontiguousarray(pieces["mean_pos"]))
    locus_pieces = LocusPieces(
        pieces["contig"].astype(np.uint32)[order], pieces["start"].astype(np.int32)[order],
        pieces["end"].astype(np.int32)[order], pieces["strand"].astype(np.int8)[order],
    )  # fmt: skip
    n_exp, n_missing = int(summary["n_expected"]), int(summary["n_missing"])
    completeness = 1.0 - (n_missing / n_exp) if n_exp > 0 else 1.0
    g0 = int(db.locus_gene_offsets[best])
    mask = summary["missing_mask"]
    missing = tuple(
        db.genes.ids[g0 + j] for j in range(min(int(db.locus_gene_lengths[best]), MAX_LOCUS_GENES))
        if (int(mask[j >> 6]) >> (j & 63)) & 1
    )  # fmt: skip
    gene_ids = tuple(typer._gene_ids_obj[g])
    if genome is not None:
        locus_seqs = (
            genome.contigs.extract(locus_pieces.ctg_indices, locus_pieces.starts, locus_pieces.ends, locus_pieces.strands)
            if len(locus_pieces) else Sequences.empty()
        )  # fmt: skip
        gene_seqs = genome.contigs.extract_intervals(hits.t_indices, hits.t_intervals, new_ids=gene_ids)
        prot_seqs = gene_seqs.translate(frames=hits.frames, to_stop=True)
    else:
        zeros = np.zeros(len(k), np.int32)
        locus_seqs = Sequences.empty()
        gene_seqs = Sequences(gene_ids, np.empty(0, np.uint8), zeros, zeros)
        prot_seqs = Sequences(gene_ids, np.empty(0, np.uint8), zeros, zeros)
    return typer.finish(
        genome_id, best, best_score, completeness, hits, k["state"].copy(), k["pident"].copy(), locus_pieces, missing,
        locus_seqs, gene_seqs, prot_seqs,
    )  # fmt: skip


class BatchTyping:
    """Typing results of a whole batch as columns (one row per assembly), finished with array operations.

    Everything ``Serotyper.finish`` decides per genome (core.py:343-349, 395-459; models.py:538-558) is computed here
    for all assemblies at once; ``result(i)`` builds the full ``SerotypingResult`` of one assembly on demand and
    ``rows()`` formats the TSV lines.  Scalars per assembly:

    best_locus, best_score, completeness, percent_identity, percent_coverage, length_discrepancy, typeable, problems,
    phenotype, n_pieces, n_hits (hits kept in the result).
    """

    def __init__(self, typer, ids, sums, kept, pieces, scores, best, genomes=None) -> None:
        self.typer, self.ids, self.genomes = typer, list(ids), genomes
        self.sums, self.kept, self.pieces = sums, kept, pieces
        db = typer._db
        n = len(sums)
        self.best_locus = np.asarray(best, np.int32)
        self.best_score = scores[np.arange(n), self.best_locus] if n else np.zeros(0)
        n_kept, n_pieces = sums["n_kept"], sums["n_pieces"]
        valid = np.arange(kept.shape[1])[None, :] < n_kept[:, None]
        flags, state = kept["flags"], kept["state"]
        alive = valid & ((flags & F_SPURIOUS) == 0)
        inside, expected, extra = (flags & F_INSIDE) != 0, (flags & F_EXPECTED) != 0, (flags & F_EXTRA) != 0
        self.alive = alive
        self.n_hits = alive.sum(axis=1)
        self.n_pieces = n_pieces
        # completeness of the reconstructed locus (core.py:299-301)
        n_exp, n_missing = sums["n_expected"], sums["n_missing"]
        with np.errstate(divide="ignore", invalid="ignore"):
            self.completeness = np.where(n_exp > 0, 1.0 - (n_missing / np.maximum(n_exp, 1)), 1.0)
        # locus coverage and length discrepancy (core.py:343-349)
        pvalid = np.arange(pieces.shape[1])[None, :] < n_pieces[:, None]
        assem_len = np.where(pvalid, pieces["end"].astype(np.int64) - pieces["start"], 0).sum(axis=1)
        ref_len = db.loci.lengths[self.best_locus].astype(np.int64)
        with np.errstate(divide="ignore", invalid="ignore"):
            cov = np.minimum(100.0, (assem_len / ref_len) * 100.0)
        self.percent_coverage = np.where(ref_len > 0, cov, 0.0)
        self.length_discrepancy = np.where(n_pieces == 1, (assem_len - ref_len).astype(np.float64), np.nan)
        # mean identity over NORMAL genes (core.py:395-396): the float32 sum arrives with numpy's own association
        # (kp_np_sum_f32); np.mean then divides by the count in float64 and rounds to float32
        counts = sums["n_normal"]
        with np.errstate(divide="ignore", invalid="ignore"):
            mean32 = (sums["ident_sum"].astype(np.float64) / counts).astype(np.float32)
        self.percent_identity = np.where(counts > 0, mean32, np.float32(0)).astype(np.float64)
        # confidence (core.py:445-459)
        unexpected = alive & inside & ~expected & ~extra & (state != 2)
        novel_inside = (alive & inside & (state == 3)).any(axis=1)
        self.typeable = (
            (self.completeness >= typer.min_completeness)
            & (unexpected.sum(axis=1) <= typer.max_other_genes)
            & (typer.allow_below_threshold | ~novel_inside)
        )
        # problem flags (models.py:538-558)
        p = (n_pieces > 1).astype(np.int32)
        p |= 2 * (alive & inside & ~expected & ~extra).any(axis=1)
        p |= 4 * ((self.completeness < 1.0) | (alive & ~inside & expected).any(axis=1))
        p |= 8 * novel_inside
        p |= 16 * (alive & inside & ((state == 2) | (state == 1))).any(axis=1)
        self.problems = p
        self.phenotype = self._phenotypes(db, alive, state)

    def _phenotypes(self, db, alive, state) -> list:
        """Phenotype rules for the whole batch (core.py:399-442), as matrix products over (assembly, rule, cluster)
        with the reference's int8 arithmetic; only assemblies with a valid suffix rule take the per-assembly path."""
        names = [db.serotypes[b] for b in self.best_locus]
        rules = db.phenotypes
        if len(rules) == 0 or len(names) == 0:
            return names
        n, n_clu = len(names), len(db.cluster_keys)
        applies = rules.locus_masks[:, self.best_locus].T  # [n, rules]
        if not applies.any():
            return names
        # clusters with a working (NORMAL / PARTIAL) copy among the kept hits
        rows, cols = np.nonzero(alive & ((state == 0) | (state == 1)))
        active = np.zeros((n, n_clu), dtype=bool)
        active[rows, db.gene_cluster_ids[self.kept["gene"][rows, cols]]] = True
        extras_ok = (active.astype(np.int8) @ rules.extra_masks.T) == rules.extra_counts[None, :]
        expected = self.typer._expected_clusters_per_locus()[self.best_locus]  # int8 [n, clusters]
        applicable = rules.inactive_masks[None, :, :] & expected[:, None, :]  # int8 [n, rules, clusters]
        knocked_out = (applicable * (~active).astype(np.int8)[:, None, :]).sum(axis=2, dtype=np.int8)
        inactive_ok = ~(rules.inactive_masks.sum(axis=1) > 0)[None, :] | ((applicable.sum(axis=2) > 0) & (knocked_out > 0))
        valid = applies & extras_ok & inactive_ok
        if not valid.any():
            return names
        # replacing rules: the highest priority wins, the first of equals (np.argmax over the ascending subset)
        prio = np.where(valid & ~rules.as_suffix[None, :], rules.priorities.astype(np.int16)[None, :], np.int16(-32768))
        top = prio.argmax(axis=1)
        for a in np.flatnonzero(prio.max(axis=1) > -32768):
            names[a] = rules.ids[top[a]].decode("utf-8")
        for a in np.flatnonzero((valid & rules.as_suffix[None, :]).any(axis=1)):
            appending = np.flatnonzero(valid[a] & rules.as_suffix)
            ranked = appending[np.argsort(-rules.priorities[appending])]
            names[a] += "".join(rules.ids[i].decode("utf-8") for i in ranked)
        return names

    def __len__(self) -> int:
        return len(self.sums)

    def result(self, i: int) -> SerotypingResult:
        s = self.sums[i]
        return assemble(
            self.typer, self.ids[i], s, self.kept[i, : s["n_kept"]], self.pieces[i, : s["n_pieces"]],
            self.best_score[i], genome=None if self.genomes is None else self.genomes[i],
        )  # fmt: skip

    def results(self) -> list:
        return [self.result(i) for i in range(len(self))]

    def rows(self) -> list:
        from kaptive_amd.serotyping.io import KaptiveRow

        return [bytes(KaptiveRow.from_result(self.result(i))) for i in range(len(self))]

    def tsv(self) -> bytes:
        """The TSV lines of the whole batch as one byte string (what ``kaptive assembly -o`` appends per batch), formatted
        by the native library from the device records and the columns above (kp_format_rows); byte for byte what
        ``KaptiveRow.from_result`` gives for ``result(i)`` (reference: src/kaptive/serotyping/io.py:191-296)."""
        fmt = getattr(self.typer, "_row_formatter", None)
        if fmt is None:
            from kaptive_amd import KAPTIVE_COMPAT_VERSION, _native

            fmt = self.typer._row_formatter = _native.RowFormatter(self.typer._db, KAPTIVE_COMPAT_VERSION)
        return fmt.format(self.ids, self.phenotype, self.sums, self.kept, self.best_locus, self.typeable, self.problems,
                          self.percent_identity, self.percent_coverage, self.length_discrepancy)  # fmt: skip


    def jsonl(self) -> bytes:
        """The JSON lines of the whole batch (``-j``), from the batch's columns and the genomes' text (``genomes`` must have
        been given): byte for byte ``dumps_line(result(i).to_dict())`` for every assembly (reference:
        src/kaptive/serotyping/cli.py:67-76) without an object per assembly (kp_format_json)."""
        if self.genomes is None:
            raise ValueError("JSON lines carry the extracted sequences: the batch needs its genomes")
        return self._formatter().format(self.ids, self.phenotype, self.sums, self.kept, self.pieces, self.best_locus, self.best_score,
                                        self.completeness, self.typeable, self.problems, self.percent_identity, self.percent_coverage,
                                        self.length_discrepancy, self.genomes)  # fmt: skip

    def _formatter(self):
        fmt = getattr(self.typer, "_json_formatter", None)
        if fmt is None:
            from kaptive_amd import KAPTIVE_COMPAT_VERSION, _native

            fmt = self.typer._json_formatter = _native.JsonFormatter(self.typer, KAPTIVE_COMPAT_VERSION)
        return fmt

    def fasta(self, kinds=("loci", "genes", "proteins")) -> dict:
        """``{kind: [bytes per assembly]}``: what ``result(i).locus_seqs / gene_seqs / translations .to_fasta()`` give -- the
        per-assembly files of ``-l / -g / -p`` (reference: src/kaptive/serotyping/cli.py:78-114) -- without an object per
        assembly (kp_format_fasta; ``genomes`` must have been given)."""
        if self.genomes is None:
            raise ValueError("the extracted sequences need the genomes' text: the batch was made without its genomes")
        return self._formatter().fasta(tuple(kinds), self.ids, self.phenotype, self.sums, self.kept, self.pieces, self.best_locus,
                                       self.best_score, self.completeness, self.typeable, self.problems, self.percent_identity,
                                       self.percent_coverage, self.length_discrepancy, self.genomes)  # fmt: skip

    def pha4ge(self) -> bytes:
        """The PHA4GE lines of the whole batch (``--pha4ge``), from the batch's columns: byte for byte what
        ``Pha4geRow.from_result(result(i))`` gives (reference: src/kaptive/serotyping/io.py, ``Pha4geRow``) without an object
        per assembly."""
        from kaptive_amd import KAPTIVE_COMPAT_VERSION
        from kaptive_amd.serotyping.io import _PHA4GE_PROBLEMS, Pha4geRow

        db = self.typer._db
        md = db.metadata
        fixed = dict(genotyping_schema_taxon=b"%s [NCBITaxon:%d]" % (md.organism.encode(), md.taxon),
                     genotyping_database_name=md.name.encode(), genotyping_database_version=md.version.encode(),
                     genotyping_software_version=KAPTIVE_COMPAT_VERSION.encode())  # fmt: skip
        notes = [(int(flag), text) for flag, text in _PHA4GE_PROBLEMS]
        out = []
        for i, genome in enumerate(self.ids):
            locus = db.loci.ids[int(self.best_locus[i])].encode()
            p = int(self.problems[i])
            if p:
                said = [(b"match broken into %d pieces" % int(self.n_pieces[i])) if text is None else text for flag, text in notes if p & flag]
                details = b"Best locus match: %b. Problems: %b" % (locus, b", ".join(said))
            else:
                details = b"Best locus match: %b." % locus
            out.append(bytes(Pha4geRow(sample=genome.encode(), genotype=locus, genotyping_details=details,
                                       genotype_confidence_value=b"Typeable" if self.typeable[i] else b"Untypeable",
                                       genotype_predicted_phenotype=self.phenotype[i].encode(), **fixed)))  # fmt: skip
        return b"".join(out)


class _HitsView:
    """The one attribute ``Serotyper._phenotype`` reads off a GeneHits."""

    def __init__(self, gene_indices) -> None:
        self.gene_indices = gene_indices
