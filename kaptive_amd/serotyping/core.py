"""Serotyper: gene hits -> best locus, locus pieces, gene states, phenotype, typeability.

Drop-in for the reference's ``kaptive.serotyping.core.Serotyper`` (src/kaptive/serotyping/core.py:32-486): same
constructor arguments, ``__call__(genome) -> SerotypingResult``. Where the reference builds a minimizer index per
assembly and calls ``rammappy``'s ``map_batch`` (src/kaptive/serotyping/core.py:147-155), this one hands the
2-bit-packed contigs to the HIP aligner through the C-ABI (``kaptive_amd._native``); there is no CPU aligner in the
product -- a missing HIP library raises.

``reduce`` is the single-genome restatement of the reference's reduction (core.py:157-486) in numpy; it is the
specification the batched GPU reduction (``kp_reduce.hip``) is tested against, and is itself pinned against golden
vectors produced by the reference (tests/golden). Phase markers below cite the reference lines they follow.
"""

from __future__ import annotations

from pathlib import Path
from typing import Iterator, Any, Callable, Sequence

import numpy as np

from kaptive_amd import KAPTIVE_COMPAT_VERSION
from kaptive_amd.core.alignment import Alignments
from kaptive_amd.core.genome import GenomeAssembly
from kaptive_amd.core.seq import Sequences
from kaptive_amd.db import Database
from kaptive_amd.serotyping.models import GeneHits, GeneState, LocusPieces, SerotypingResult

_NORMAL, _PARTIAL, _TRUNCATED, _NOVEL = (s.value for s in GeneState)


class Serotyper:
    def __init__(
        self,
        db: Database,
        max_other_genes: int = 1,
        min_completeness: float = 0.5,
        allow_below_threshold: bool = False,
        preset: Any = None,
        scoring_metric: str = "scores",
        min_gene_coverage: float = 0.20,
        partial_edge_tolerance: int = 5,
        *,
        device: int = 0,
        aligner: Callable[[GenomeAssembly], Alignments] | None = None,
        protein_aligner: Callable[[Sequences, Sequences], Any] | None = None,
    ) -> None:
        """``preset`` and ``scoring_metric`` are accepted and ignored, as in the reference.
        ``aligner`` / ``protein_aligner`` replace the HIP stages; tests use them to feed recorded hits."""
        self._db = db
        self.max_other_genes = max_other_genes
        self.min_completeness = min_completeness
        self.allow_below_threshold = allow_below_threshold
        self.preset = preset
        self.scoring_metric = scoring_metric
        self.min_gene_coverage = min_gene_coverage
        self.partial_edge_tolerance = partial_edge_tolerance
        self._device = device
        self._aligner = aligner
        self._protein_aligner = protein_aligner
        self._engine = None

        # expected (non-extra) genes per locus, floored at 1 -- core.py:102-108
        n_exp = np.zeros(len(db.loci), dtype=np.float32)
        np.add.at(n_exp, db.gene_locus_indices[~db.extra_genes], 1.0)
        self._expected_genes_per_locus = np.maximum(n_exp, 1.0)
        self._gene_names = tuple(str(i) for i in range(len(db.genes)))
        # per-gene text columns of GeneHits, encoded once (core.py:316-327 builds them per call)
        self._gene_ids_s32 = np.array([x.encode("utf-8") for x in db.genes.ids], dtype="S32")
        self._cluster_s10 = np.array(
            [db.cluster_keys[c].encode("utf-8") for c in db.gene_cluster_ids], dtype="S10"
        ).reshape(-1)
        self._product_s64 = np.array(
            [db.description_keys[d].encode("utf-8") for d in db.gene_description_ids], dtype="S64"
        ).reshape(-1)
        self._gene_ids_obj = np.array(db.genes.ids, dtype=object)

    # -- native stages --------------------------------------------------------------------------------------------
    @property
    def engine(self):
        """Lazily created HIP context with this database resident on the device."""
        if self._engine is None:
            from kaptive_amd.engine import Engine

            early = getattr(self, "_ctx_early", None)  # (a future of _native.Context: kaptive_amd/cli.py starts the runtime early)
            self._ctx_early = None
            self._engine = Engine(self._db, device=self._device, ctx=early.result() if early is not None else None)
        return self._engine

    def align(self, genome: GenomeAssembly) -> Alignments:
        if self._aligner is not None:
            return self._aligner(genome)
        return self.engine.align([genome])[0]

    def align_proteins(self, queries: Sequences, targets: Sequences):
        if self._protein_aligner is not None:
            return self._protein_aligner(queries, targets)
        return self.engine.protein_aligner(queries, targets)

    # -- public API -----------------------------------------------------------------------------------------------
    def __call__(self, genome: GenomeAssembly | str | Path) -> SerotypingResult | None:
        """One genome: a batch of one through the same device path as ``type_many`` (alignment, reduction, translation,
        protein DP and gene states on the GPU).  With an injected ``aligner`` / ``protein_aligner`` (tests, the oracle)
        the golden-pinned host statement of the reduction, ``reduce``, runs instead."""
        genome = GenomeAssembly.ensure(genome)
        if self._aligner is None and self._protein_aligner is None:
            return self.engine.type_many(self, [genome])[0]
        return self.reduce(genome, self.align(genome))

    def call_with_host_reduction(self, genome: GenomeAssembly | str | Path) -> SerotypingResult:
        """GPU alignment and protein DP, reduction in numpy (``reduce``): what tests compare the device reduction with."""
        genome = GenomeAssembly.ensure(genome)
        return self.reduce(genome, self.align(genome))

    def type_many(self, genomes: Sequence[GenomeAssembly | str | Path]) -> list[SerotypingResult]:
        """Batch entry point (not in the reference): one device submission for all assemblies."""
        loaded = [GenomeAssembly.ensure(g) for g in genomes]
        if self._aligner is not None:
            return [self.reduce(g, self._aligner(g)) for g in loaded]
        return self.engine.type_many(self, loaded)

    def tsv_from_files(self, paths: Sequence[str | Path], batch_size: int = 512, threads: int = 0) -> "Iterator[bytes]":
        """The ``kaptive assembly ... -o`` pipeline for library callers (not in the reference, whose batch entry is its CLI,
        src/kaptive/serotyping/cli.py:183-210): FASTA files -> chunks of ``batch_size`` parsed by the library's threads ->
        pinned shards -> the batched typing -> the ``KaptiveRow`` bytes of every chunk, in input order, without a Python
        object per assembly.  Yields one ``bytes`` per chunk (no header line: ``KaptiveRow.header()``)."""
        import argparse

        from kaptive_amd.cli import _TypingPipeline

        args = argparse.Namespace(out="-", threads=threads, json=None, loci=None, genes=None, proteins=None, pha4ge=None)
        paths = [str(p) for p in paths]
        chunks = [(k, paths[i : i + batch_size]) for k, i in enumerate(range(0, len(paths), batch_size))]
        pipe = _TypingPipeline(args, 0, typer=self)  # (the device is the one this Serotyper was made for)
        try:
            for _, out in pipe.run(chunks):
                yield out["tsv"]
        finally:
            pipe.close()

    # -- reduction ------------------------------------------------------------------------------------------------
    def score_loci(self, alns: Alignments, gene_idx: np.ndarray) -> tuple[np.ndarray, np.ndarray, int]:
        """Scoring phase (core.py:164-207): returns (locus_scores f64, penalised scores f64, best locus index)."""
        db = self._db
        q_covs = alns.q_covs
        ok = q_covs >= self.min_gene_coverage
        g, cov, sc = gene_idx[ok], q_covs[ok], alns.scores[ok]
        order = np.lexsort((-sc, -cov, g))
        g, cov = g[order], cov[order]
        _, first = np.unique(g, return_index=True)
        best_g, best_cov = g[first], cov[first]
        counted = ~db.extra_genes[best_g]
        locus_scores = np.zeros(len(db.loci), dtype=np.float64)
        np.add.at(locus_scores, db.gene_locus_indices[best_g][counted], best_cov[counted])
        counts = np.zeros(len(db.loci), dtype=np.float32)
        np.add.at(counts, db.gene_locus_indices[best_g[counted]], 1.0)
        completeness = counts / self._expected_genes_per_locus
        final = locus_scores * (completeness**3)
        self._last_scores, self._last_completeness = final.copy(), completeness.copy()
        return locus_scores, final, int(np.argmax(final))

    def reduce(self, genome: GenomeAssembly, gene_alns: Alignments) -> SerotypingResult:
        db = self._db
        n_genes = len(db.genes)
        gene_idx_all = gene_alns.q_names.astype(np.int32)

        # provisional per-gene coverage over every hit -- core.py:158-162
        total_q_covs = np.zeros(n_genes, dtype=np.float32)
        np.add.at(total_q_covs, gene_idx_all, gene_alns.q_aln_lens)
        total_q_covs /= db.genes.lengths

        locus_scores, _, best = self.score_loci(gene_alns, gene_idx_all)

        # reconstruction: cull all hits, best-locus genes first -- core.py:210-223
        culled = gene_alns.cull_overlaps(
            by_query=False, priority_mask=db.gene_locus_indices[gene_idx_all] == best, max_overlap_fraction=0.1
        )
        g = culled.q_names.astype(np.int32)
        n = len(culled)
        t_idx = np.array([genome.id_map[name] for name in culled.t_names], dtype=np.uint32)
        spans = culled.to_intervals(by_query=False)
        piece_ids = spans.cluster_spatial(tolerance=db.max_locus_length, group_by=t_idx)

        is_extra = db.extra_genes[g]
        is_expected = (db.gene_locus_indices[g] == best) & ~is_extra
        coverages = np.clip(total_q_covs[g] * 100.0, 0.0, 100.0)

        # top-scoring hit of each expected gene anchors the piece boundaries -- core.py:236-245
        primary = np.zeros(n, dtype=bool)
        exp_rows = np.flatnonzero(is_expected)
        if len(exp_rows):
            order = np.lexsort((-culled.scores[exp_rows], g[exp_rows]))
            _, first = np.unique(g[exp_rows][order], return_index=True)
            primary[exp_rows[order[first]]] = True

        # one piece per cluster that holds a primary hit -- core.py:248-288
        p_ctg, p_start, p_end, p_strand, p_mean = [], [], [], [], []
        for cid in np.unique(piece_ids[is_expected]):
            in_piece = piece_ids == cid
            anchors = in_piece & primary
            if not anchors.any():
                continue
            genes_here = g[anchors]
            p_ctg.append(t_idx[in_piece][0])
            p_start.append(np.min(spans.starts[anchors]))
            p_end.append(np.max(spans.ends[anchors]))
            p_mean.append(np.mean(db.gene_positions[genes_here]))
            agree = np.sum(culled.strands[anchors] * db.gene_intervals.strands[genes_here])
            p_strand.append(-1 if agree < 0 else 1)

        is_inside = np.zeros(n, dtype=bool)
        for c, s, e in zip(p_ctg, p_start, p_end):
            is_inside |= (t_idx == c) & (spans.starts <= e) & (spans.ends >= s)

        by_position = np.argsort(p_mean)
        pieces = LocusPieces(
            np.array(p_ctg, dtype=np.uint32)[by_position],
            np.array(p_start, dtype=np.int32)[by_position],
            np.array(p_end, dtype=np.int32)[by_position],
            np.array(p_strand, dtype=np.int8)[by_position],
        )

        # completeness of the reconstructed locus -- core.py:291-301
        expected_genes = np.flatnonzero((db.gene_locus_indices == best) & ~db.extra_genes)
        missing = np.setdiff1d(expected_genes, g[is_expected & is_inside], assume_unique=True)
        completeness = 1.0 - (len(missing) / len(expected_genes)) if len(expected_genes) > 0 else 1.0

        hits = self.gene_hits_table(
            g, culled.q_starts, culled.q_ends, t_idx, culled.t_starts, culled.t_ends, culled.strands, is_expected,
            is_inside, is_extra, coverages,
        )

        # gene states -- core.py:352-379
        gene_seqs = genome.contigs.extract_intervals(
            hits.t_indices, hits.t_intervals, new_ids=tuple(db.genes.ids[i] for i in g)
        )
        prot_seqs = gene_seqs.translate(frames=hits.frames, to_stop=True)
        states = np.full(n, _NORMAL, dtype=np.int8)
        is_partial = culled.is_partial(self.partial_edge_tolerance)
        prot_covs = (prot_seqs.lengths * 3.0) / db.genes.lengths[g]
        hits.coverages[:] = np.clip(prot_covs * 100.0, 0.0, 100.0)
        states[is_partial] = _PARTIAL
        states[~is_partial & (prot_covs < 0.90)] = _TRUNCATED
        prot_alns = self.align_proteins(prot_seqs, db.translations[g])
        idents = prot_alns.pidents.astype(np.float32)

        # weak homologues outside the locus are dropped; weak NORMAL genes become NOVEL -- core.py:383-394
        threshold = db.metadata.id_threshold
        spurious = ~hits.is_inside & (idents < threshold)
        if spurious.any():
            keep = ~spurious
            hits, gene_seqs, prot_seqs = hits[keep], gene_seqs[keep], prot_seqs[keep]
            states, idents = states[keep], idents[keep]
        states[(states == _NORMAL) & (idents < threshold)] = _NOVEL

        locus_seqs = (
            genome.contigs.extract(pieces.ctg_indices, pieces.starts, pieces.ends, pieces.strands)
            if len(pieces)
            else Sequences.empty()
        )
        return self.finish(
            genome.id, best, locus_scores[best], completeness, hits, states, idents, pieces,
            tuple(db.genes.ids[i] for i in missing), locus_seqs, gene_seqs, prot_seqs,
        )

    def gene_hits_table(self, g, q_starts, q_ends, t_idx, t_starts, t_ends, strands, is_expected, is_inside, is_extra,
                        coverages) -> GeneHits:
        """GeneHits with the database-derived columns filled in (core.py:303-329)."""
        db = self._db
        return GeneHits(
            gene_indices=g, q_starts=q_starts, q_ends=q_ends, t_indices=t_idx, t_starts=t_starts, t_ends=t_ends,
            strands=strands, is_expected=is_expected, is_inside=is_inside, is_extra=is_extra,
            expected_positions=db.gene_positions[g].astype(np.int32),
            expected_strands=db.gene_intervals.strands[g],
            gene_ids=self._gene_ids_s32[g], cluster_names=self._cluster_s10[g],
            product_descriptions=self._product_s64[g], coverages=coverages,
        )  # fmt: skip

    def finish(self, genome_id, best, best_score, completeness, hits, states, idents, pieces, missing_ids, locus_seqs,
               gene_seqs, prot_seqs) -> SerotypingResult:
        """Everything after the per-hit work: locus coverage, mean identity, phenotype, confidence, result object
        (core.py:343-349, 395-486).  Shared by the single-genome reduction and the batched GPU reduction."""
        db = self._db
        assem_len = np.sum(pieces.ends - pieces.starts)
        ref_len = db.loci.lengths[best]
        pcov = float(min(100.0, (assem_len / ref_len) * 100.0)) if ref_len > 0 else 0.0
        discrepancy = float(assem_len - ref_len) if len(pieces) == 1 else float("nan")
        normal_idents = idents[states == _NORMAL]
        pident = float(np.mean(normal_idents)) if normal_idents.size > 0 else 0.0
        phenotype = self._phenotype(best, hits, states)

        # confidence -- core.py:445-459
        unexpected = hits.is_inside & ~hits.is_expected & ~hits.is_extra & (states != _TRUNCATED)
        typeable = (
            completeness >= self.min_completeness
            and np.count_nonzero(unexpected) <= self.max_other_genes
            and (self.allow_below_threshold or not np.any(hits.is_inside & (states == _NOVEL)))
        )
        meta = db.metadata
        return SerotypingResult(
            kaptive_version=KAPTIVE_COMPAT_VERSION,
            database_name=meta.name,
            database_version=meta.version,
            database_organism=meta.organism,
            database_taxon=meta.taxon,
            genome=genome_id,
            best_locus_idx=best,
            best_locus_name=db.loci.ids[best],
            best_locus_score=best_score,
            best_locus_completeness=completeness,
            length_discrepancy=discrepancy,
            gene_hits=hits,
            gene_states=states,
            locus_pieces=pieces,
            locus_seqs=locus_seqs,
            gene_seqs=gene_seqs,
            translations=prot_seqs,
            percent_identity=pident,
            percent_coverage=pcov,
            protein_identities=idents,
            phenotype=phenotype,
            typeable=bool(typeable),
            missing_expected_genes=missing_ids,
        )

    def _expected_clusters_per_locus(self) -> np.ndarray:
        """int8 [n_loci, n_clusters]: 1 where the locus has a gene of the cluster (rows of `expected` in _phenotype)."""
        cached = getattr(self, "_expected_clusters", None)
        if cached is None:
            db = self._db
            cached = np.zeros((len(db.loci), len(db.cluster_keys)), dtype=np.int8)
            for li, (o, ln) in enumerate(zip(db.locus_gene_offsets.tolist(), db.locus_gene_lengths.tolist())):
                cached[li, db.gene_cluster_ids[o : o + ln]] = 1
            self._expected_clusters = cached
        return cached

    def _phenotype(self, best: int, hits: GeneHits, states: np.ndarray) -> str:
        """Apply the database's phenotype rules to the best locus' serotype -- core.py:399-442."""
        db = self._db
        name = db.serotypes[best]
        rules = db.phenotypes
        if len(rules) == 0:
            return name
        active = np.zeros(len(db.cluster_keys), dtype=bool)
        working = (states == _NORMAL) | (states == _PARTIAL)
        active[db.gene_cluster_ids[hits.gene_indices[working]]] = True

        extras_ok = np.dot(rules.extra_masks, active.astype(np.int8)) == rules.extra_counts
        expected = np.zeros(len(db.cluster_keys), dtype=np.int8)
        o, ln = db.locus_gene_offsets[best], db.locus_gene_lengths[best]
        expected[db.gene_cluster_ids[o : o + ln]] = 1
        applicable = rules.inactive_masks & expected
        knocked_out = np.dot(applicable, (~active).astype(np.int8))
        inactive_ok = ~(rules.inactive_masks.sum(axis=1) > 0) | ((applicable.sum(axis=1) > 0) & (knocked_out > 0))

        valid = np.flatnonzero(rules.locus_masks[:, best] & extras_ok & inactive_ok)
        if len(valid) == 0:
            return name
        suffix = rules.as_suffix[valid]
        replacing, appending = valid[~suffix], valid[suffix]
        if len(replacing):
            name = rules.ids[replacing[np.argmax(rules.priorities[replacing])]].decode("utf-8")
        if len(appending):
            ranked = appending[np.argsort(-rules.priorities[appending])]
            name += "".join(rules.ids[i].decode("utf-8") for i in ranked)
        return name
