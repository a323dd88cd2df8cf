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
     ("n_expected", "<i4"), ("n_missing", "<i4"), ("overflow", "<i4"), ("missing_mask", "<u8", MAX_LOCUS_GENES // 64)]
)  # fmt: skip
assert KEPT_DTYPE.itemsize == 84 and PIECE_DTYPE.itemsize == 24 and SUMMARY_DTYPE.itemsize == 64


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
