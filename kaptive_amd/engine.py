"""Engine: a database resident on one GPU, plus the batched entry points built on the C ABI.

This is the host-side replacement for what the reference does per genome inside ``Serotyper.__call__`` before its
reduction starts: build an index, construct an ``Aligner`` and call ``map_batch`` (src/kaptive/serotyping/core.py:
145-155).  Here the database is uploaded once, assemblies go to the device as packed batches, and hit tables come
back as columns.
"""

from __future__ import annotations

from typing import Sequence

import numpy as np

from kaptive_amd import _native
from kaptive_amd.core.alignment import Alignments
from kaptive_amd.core.genome import GenomeAssembly
from kaptive_amd.core.pairwise import PairwiseAlignments
from kaptive_amd.core.seq import Sequences
from kaptive_amd.db import Database
from kaptive_amd.pack import pack_sequences_flat


class Engine:
    """One database -- or several that are typed together (K and O loci) -- resident on one GPU.

    With a list of databases the genes of all of them go into one seed index, so a batch is scanned, chained and
    aligned once for all of them (the reference runs one ``map_batch`` per database); every database keeps its own
    typing tables as a *group* of the context and ``view(i)`` gives the engine as database ``i`` sees it.  The
    single-genome entry points (``align``, ``hits_to_alignments``) are those of a one-database engine."""

    def __init__(self, db: "Database | Sequence[Database]", device: int = 0, ctx: "_native.Context | None" = None) -> None:
        """``ctx``: a context of ``device`` the caller created ahead of time (the command line starts the runtime on a thread
        of its own while the database file is still being read); otherwise one is created here."""
        dbs = list(db) if isinstance(db, (list, tuple)) else [db]
        self.dbs = dbs
        self.db = dbs[0]
        self.group = 0
        self.device = device
        self.ctx = ctx if ctx is not None else _native.Context(device)
        for d in dbs:  # KP_MAX_GENE_LEN (include/kp_spec.h): query positions are 16-bit fields of the anchor and hit keys
            too_long = np.flatnonzero(np.asarray(d.genes.lengths) > _native.MAX_GENE_LEN)
            if len(too_long):
                g = int(too_long[0])
                raise ValueError(f"database {d.metadata.keyword!r}: gene {d.genes.ids[g]!r} is {int(d.genes.lengths[g])} bases long; "
                                 f"the aligner handles genes up to {_native.MAX_GENE_LEN} bases (KP_MAX_GENE_LEN) -- "
                                 f"{len(too_long)} gene(s) exceed it")  # fmt: skip
        packed = [pack_sequences_flat(d.genes) for d in dbs]
        counts = [len(d.genes) for d in dbs]
        self.gene_ranges = [(sum(counts[:i]), sum(counts[: i + 1])) for i in range(len(dbs))]
        codes = np.concatenate([c for c, _ in packed]) if len(dbs) > 1 else packed[0][0]
        if len(dbs) > 1:
            base = np.cumsum([0] + [len(c) for c, _ in packed[:-1]])
            off = np.concatenate([o[:-1] + b for (_, o), b in zip(packed, base)] + [[len(codes)]]).astype(packed[0][1].dtype)
        else:
            off = packed[0][1]
        self.ctx.load_genes(codes, off)
        for g, (d, (lo, hi)) in enumerate(zip(dbs, self.gene_ranges)):
            self.ctx.load_typing(d, group=g, gene_lo=lo, gene_hi=hi)
        self._gene_names = tuple(str(i) for i in range(len(self.db.genes)))

    def view(self, group: int) -> "Engine":
        """This engine as database ``group`` sees it: same context and batches, that database's typing group."""
        if group == self.group and self.db is self.dbs[group]:
            return self
        v = object.__new__(Engine)
        v.__dict__.update(self.__dict__)
        v.db, v.group = self.dbs[group], group
        v._gene_names = tuple(str(i) for i in range(len(v.db.genes)))
        return v

    def close(self) -> None:
        self.ctx.close()

    # -- stages -------------------------------------------------------------------------------------------------
    def align_packed(self, packed: list):
        """Hit table for a list of PackedAssembly: (hits, hit_off, stats)."""
        batch = self.ctx.batch(packed)
        try:
            hits, off = batch.align()
            return hits, off, batch.stats()
        finally:
            batch.close()

    def hits_to_alignments(self, genome: GenomeAssembly, hits: np.ndarray) -> Alignments:
        if len(hits) == 0:
            return Alignments.empty()
        db = self.db
        return Alignments.from_hit_table(
            self._gene_names, genome.contigs.ids,
            q_ids=hits["gene"], q_lengths=db.genes.lengths[hits["gene"]], q_starts=hits["q_start"],
            q_ends=hits["q_end"], t_ids=hits["contig"], t_lengths=genome.contigs.lengths[hits["contig"]],
            t_starts=hits["t_start"], t_ends=hits["t_end"], strands=hits["strand"], block_lens=hits["block_len"],
            matches=hits["matches"], scores=hits["score"], mapqs=hits["mapq"],
        )  # fmt: skip

    def align(self, genomes: Sequence[GenomeAssembly]) -> list[Alignments]:
        hits, off, _ = self.align_packed([g.packed() for g in genomes])
        return [self.hits_to_alignments(g, hits[off[i] : off[i + 1]]) for i, g in enumerate(genomes)]

    def protein_aligner(self, queries: Sequences, targets: Sequences) -> PairwiseAlignments:
        if len(queries.offsets) != len(targets.offsets):
            raise ValueError("Query and target batches must have the same number of sequences.")
        if len(queries.offsets) == 0:
            return PairwiseAlignments.empty()
        return PairwiseAlignments.from_table(
            self.ctx.protein_align(queries.seqs, queries.offsets, queries.lengths, targets.seqs, targets.offsets,
                                   targets.lengths)
        )  # fmt: skip

    def typing_params(self, typer) -> "_native.TypingParams":
        db = self.db
        return _native.TypingParams(
            typer.min_gene_coverage, float(np.float32(db.metadata.id_threshold)), db.max_locus_length,
            typer.partial_edge_tolerance,
        )

    def type_batch(self, typer, batch, ids: Sequence[str], genomes: Sequence[GenomeAssembly] | None = None,
                   aligned: bool = False):
        """Whole typing of a resident batch: alignment and reduction on the device, the numpy float steps, and the
        per-assembly decisions as columns.  Returns a ``BatchTyping``; ``.results()`` / ``.rows()`` materialise objects
        and TSV lines.  ``genomes`` (optional) lets the result objects carry the extracted sequences; ``aligned`` says
        that ``batch.align_async()`` has already been enqueued."""
        from kaptive_amd.serotyping import batch as B

        if not aligned:
            batch.align_async()
        scores, counts = batch.score(typer.min_gene_coverage, self.group)
        best, _, _ = B.choose_best_loci(scores, counts, typer._expected_genes_per_locus)
        batch.reduce_async(best, self.typing_params(typer), self.group)
        sums, kept, pieces = batch.typing(self.group)
        return B.BatchTyping(typer, ids, sums, kept, pieces, scores, best, genomes)

    def type_batches(self, typer, batches: Sequence, ids: Sequence[Sequence[str]], aligned: bool = False) -> list:
        """Any number of resident batches through the same context, software-pipelined as a sliding window.

        A context keeps the results of its ``_native.WORK_SLOTS`` most recent alignment passes (the work sets rotate
        at ``kp_batch_align``), so at most that many batches are between "alignment enqueued" and "records
        collected" at any time: batch i's scores are read and its reduction enqueued, then batch i-1's records are
        collected (its reduction ran while i was being scored), and alignment passes are enqueued ahead only as far
        as the free work sets allow.  The stream never waits for the host and no pass is displaced before it was
        read.  ``aligned=True`` (the caller already enqueued every pass) is only possible for up to WORK_SLOTS batches."""
        from kaptive_amd.serotyping import batch as B

        n, depth = len(batches), _native.WORK_SLOTS
        if aligned and n > depth:
            raise ValueError(f"{n} batches were aligned up front but a context keeps only {depth} alignment results "
                             "(WORK_SLOTS); pass aligned=False and let type_batches schedule the passes")  # fmt: skip
        out: list = []
        enqueued = n if aligned else 0
        pending = None  # (index, scores, best) of the batch whose reduction is enqueued but not yet collected

        def collect(item) -> None:
            i, scores, best = item
            sums, kept, pieces = batches[i].typing(self.group)
            out.append(B.BatchTyping(typer, ids[i], sums, kept, pieces, scores, best))

        for i in range(n):
            first_live = pending[0] if pending is not None else i
            while enqueued < n and enqueued < first_live + depth:
                batches[enqueued].align_async()
                enqueued += 1
            scores, counts = batches[i].score(typer.min_gene_coverage, self.group)
            best, _, _ = B.choose_best_loci(scores, counts, typer._expected_genes_per_locus)
            batches[i].reduce_async(best, self.typing_params(typer), self.group)
            if pending is not None:
                collect(pending)
            pending = (i, scores, best)
        if pending is not None:
            collect(pending)
        return out

    def type_stream(self, typer, source):
        """``type_batches`` for a stream whose length is not known in advance (the CLI: batches are made while files are
        still being read).  ``source`` yields ``(batch, ids, genomes_or_None)``; what comes out, in the same order, is
        ``(BatchTyping, batch)`` -- the caller closes the batch.  Same sliding window: at most ``_native.WORK_SLOTS``
        batches are between "alignment enqueued" and "records collected", the next batch is only pulled from ``source``
        (created and uploaded) when a work set is free for it, and batch i's records are collected after batch i + 1's
        reduction has been enqueued -- or at once, when the source has no further batch ready yet."""
        from collections import deque

        from kaptive_amd.serotyping import batch as B

        depth = _native.WORK_SLOTS
        it = iter(source)
        live: deque = deque()  # aligned, not yet scored
        pending = None  # reduction enqueued, records not yet collected
        exhausted = False

        def collect(item):
            batch, ids, genomes, scores, best = item
            sums, kept, pieces = batch.typing(self.group)
            return B.BatchTyping(typer, ids, sums, kept, pieces, scores, best, genomes), batch

        # A source may say whether its next item can be had without waiting (``ready()``: the CLI's reader pipeline): the
        # window is then only topped up with what is there, and the driver waits for the source only when it has nothing
        # else to do -- the first chunk's rows leave as soon as its reduction is through, not after two more chunks were read.
        can_pull = getattr(source, "ready", None)

        def fill_ready() -> None:
            nonlocal exhausted
            while not exhausted and len(live) + (1 if pending is not None else 0) < depth:
                if can_pull is not None and (live or pending is not None) and not can_pull():
                    return
                try:
                    item = next(it)
                except StopIteration:
                    exhausted = True
                    return
                item[0].align_async()
                live.append(item)

        try:
            while True:
                fill_ready()
                if live:
                    batch, ids, genomes = live.popleft()
                    scores, counts = batch.score(typer.min_gene_coverage, self.group)
                    best, _, _ = B.choose_best_loci(scores, counts, typer._expected_genes_per_locus)
                    batch.reduce_async(best, self.typing_params(typer), self.group)
                    if pending is not None:
                        done, pending = pending, None
                        yield collect(done)
                    pending = (batch, ids, genomes, scores, best)
                elif pending is not None:
                    done, pending = pending, None
                    yield collect(done)
                elif exhausted:
                    break
        finally:  # (an error, or a consumer that stopped early: whatever is still in the window is closed before the context goes)
            for item in list(live) + ([pending] if pending is not None else []):
                try:
                    item[0].close()
                except Exception:  # noqa: BLE001
                    pass

    def reduce_batches(self, typer, batches: Sequence, aligned: bool = False) -> list:
        """Scores back, best loci chosen (numpy), reductions enqueued, for up to WORK_SLOTS batches at once.  Returns what
        ``collect_batches`` needs; callers driving several databases put the other database's host work in between
        (bench.py does, one batch at a time).  Longer lists go through ``type_batches``, which slides a window."""
        if len(batches) > _native.WORK_SLOTS:
            raise ValueError(f"{len(batches)} batches at once, but a context keeps only {_native.WORK_SLOTS} alignment "
                             "results (WORK_SLOTS): use type_batches")  # fmt: skip
        if not aligned:
            for b in batches:
                b.align_async()
        return self.enqueue_reductions(typer, batches, self.score_batches(typer, batches))

    def score_batches(self, typer, batches: Sequence) -> list:
        """Locus scores of every batch read back and the best loci chosen (numpy argmax: part of the bit-exact
        contract).  Cheap on the device; callers with several databases score all of them before any reduction is
        enqueued, so that no database's scores queue up behind another one's reduction kernels."""
        from kaptive_amd.serotyping import batch as B

        staged = []
        for b in batches:
            scores, counts = b.score(typer.min_gene_coverage, self.group)
            best, _, _ = B.choose_best_loci(scores, counts, typer._expected_genes_per_locus)
            staged.append((scores, best))
        return staged

    def enqueue_reductions(self, typer, batches: Sequence, staged: list) -> list:
        for b, (_, best) in zip(batches, staged):
            b.reduce_async(best, self.typing_params(typer), self.group)
        return staged

    def collect_batches(self, typer, batches: Sequence, ids: Sequence[Sequence[str]], staged: list) -> list:
        """Second half: fetch the reduction records and finish them column-wise (``BatchTyping``)."""
        from kaptive_amd.serotyping import batch as B

        out = []
        for b, i, (scores, best) in zip(batches, ids, staged):
            sums, kept, pieces = b.typing(self.group)
            out.append(B.BatchTyping(typer, i, sums, kept, pieces, scores, best))
        return out

    def type_many(self, typer, genomes: Sequence[GenomeAssembly]) -> list:
        """One device submission for all genomes: alignment and reduction both run on the GPU."""
        batch = self.ctx.batch([g.packed() for g in genomes])
        try:
            return self.type_batch(typer, batch, [g.id for g in genomes], genomes).results()
        finally:
            batch.close()
