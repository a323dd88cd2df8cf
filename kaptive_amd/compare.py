"""All-vs-all protein comparison of loci (reference: src/kaptive/compare.py:195-396).

``LocusComparator`` takes a list of loci (their proteins and gene coordinates), finds for every protein of locus i its
best-matching protein in every later locus j with randstrobe seeds (``core.kmers``) and aligns each such pair with the
banded protein kernel in its seeded mode, on the GPU (``PairwiseAligner.align_seeds`` -> kp_protein_align_seeded).  The
result containers carry the reference's field names, so plotting code written against it keeps working: SURVEY.md
section 8 row f4.
"""

from __future__ import annotations

from dataclasses import dataclass
from typing import Any, Iterable, Sequence

import numpy as np

from kaptive_amd.core.interval import Intervals
from kaptive_amd.core.kmers import RandstrobeIndex
from kaptive_amd.core.pairwise import PairwiseAligner, PairwiseAlignments
from kaptive_amd.core.seq import Sequences

_EDGE_COLS = ("query_locus_indices", "target_locus_indices", "query_indices", "target_indices", "global_query_indices",
              "global_target_indices")  # fmt: skip


@dataclass(slots=True, frozen=True)
class LocusComparisonEdges:
    """Protein-level hits between loci: index columns plus the alignment statistics of every edge (compare.py:33-135)."""

    query_locus_indices: np.ndarray
    target_locus_indices: np.ndarray
    query_indices: np.ndarray
    target_indices: np.ndarray
    global_query_indices: np.ndarray
    global_target_indices: np.ndarray
    alignments: PairwiseAlignments

    def __len__(self) -> int:
        return len(self.query_locus_indices)

    def __getitem__(self, item: Any) -> "LocusComparisonEdges":
        if isinstance(item, (int, np.integer)):
            raise NotImplementedError("Single item access not implemented for LocusComparisonEdges")
        return LocusComparisonEdges(*(getattr(self, c)[item] for c in _EDGE_COLS), alignments=self.alignments[item])

    @classmethod
    def empty(cls) -> "LocusComparisonEdges":
        return cls(*(np.empty(0, dtype=np.int32) for _ in _EDGE_COLS), alignments=PairwiseAlignments.empty())

    @classmethod
    def concat(cls, batches: Iterable["LocusComparisonEdges"]) -> "LocusComparisonEdges":
        bs = list(batches)
        if not bs:
            return cls.empty()
        return cls(*(np.concatenate([getattr(b, c) for b in bs]) for c in _EDGE_COLS),
                   alignments=PairwiseAlignments.concat([b.alignments for b in bs]))  # fmt: skip


@dataclass(slots=True, frozen=True)
class LocusComparisons:
    """Edges plus per-locus and per-gene metadata, coordinates normalised for plotting (compare.py:138-171)."""

    edges: LocusComparisonEdges
    locus_names: tuple[str, ...]
    locus_lengths: np.ndarray
    locus_offsets: np.ndarray
    gene_names: np.ndarray
    gene_descriptions: np.ndarray
    gene_states: np.ndarray
    gene_intervals: Intervals


@dataclass(slots=True, frozen=True)
class LocusData:
    """One locus as the comparator takes it (compare.py:174-192)."""

    proteins: Sequences
    name: str
    backbone: Intervals
    pieces: Any = None  # LocusPieces of a fragmented locus
    gene_ctg_indices: np.ndarray | None = None
    gene_states: np.ndarray | None = None
    gene_descriptions: Any = None


def _descriptions(inp: LocusData, n: int) -> np.ndarray:
    if inp.gene_descriptions is None:
        return np.array([""] * n, dtype=object)
    raw = np.asarray(inp.gene_descriptions)
    if raw.dtype.kind == "S":
        out = np.asarray(np.char.decode(raw, "utf-8"), dtype=object)
    else:
        out = np.asarray([x.decode("utf-8") if isinstance(x, (bytes, np.bytes_)) else ("" if x is None else str(x))
                          for x in raw.flat], dtype=object).reshape(raw.shape)  # fmt: skip
    if len(out) != n:
        raise ValueError(f"Locus '{inp.name}': gene_descriptions length ({len(out)}) does not match protein count ({n})")
    return out


class LocusComparator:
    """Forward upper-triangle comparison: proteins of locus i against their best hit in every locus j > i."""

    def __init__(self, k: int = 10, s: int = 5, min_score: int = 1, aligner_kwargs: dict | None = None) -> None:
        self.k, self.s, self.min_score = k, s, min_score
        self.aligner = PairwiseAligner(**(aligner_kwargs or {}))

    def __call__(self, inputs: Sequence[LocusData]) -> LocusComparisons:
        loci = [inp.proteins for inp in inputs]
        n_loci = len(loci)
        global_seqs = Sequences.concat(loci) if n_loci > 0 else Sequences.empty()
        desc, states = [], []
        for inp in inputs:
            n = len(inp.proteins)
            if len(inp.backbone) != n:
                raise ValueError(f"Locus '{inp.name}': backbone length ({len(inp.backbone)}) does not match protein count ({n})")
            desc.append(_descriptions(inp, n))
            if inp.gene_states is not None:
                st = np.asarray(inp.gene_states, dtype=np.int8)
                if len(st) != n:
                    raise ValueError(f"Locus '{inp.name}': gene_states length ({len(st)}) does not match protein count ({n})")
                states.append(st)
            else:
                states.append(np.zeros(n, dtype=np.int8))  # GeneState.NORMAL
        gene_descriptions = np.concatenate(desc) if n_loci else np.empty(0, dtype=object)
        gene_states = np.concatenate(states).astype(np.int8) if n_loci else np.empty(0, dtype=np.int8)

        # gene coordinates on one axis per locus: pieces of a fragmented locus side by side, else shifted to start at 0
        norm = []
        for inp in inputs:
            bb, lp = inp.backbone, inp.pieces
            if lp is not None:
                piece = np.zeros(len(bb), dtype=np.int32)
                for p in range(len(lp)):
                    inside = (bb.starts >= lp.starts[p]) & (bb.ends <= lp.ends[p])
                    if inp.gene_ctg_indices is not None:
                        inside &= inp.gene_ctg_indices == lp.ctg_indices[p]
                    piece[inside] = p
                norm.append(bb.arrange(piece, np.arange(len(lp), dtype=np.int32), lp.starts, lp.ends, lp.strands))
            else:
                norm.append(bb.shift(-np.min(bb.starts)) if len(bb) > 0 else bb)
        if norm:
            gene_intervals = Intervals(
                np.concatenate([b.starts for b in norm]), np.concatenate([b.ends for b in norm]),
                np.concatenate([b.strands for b in norm]), np.concatenate([b.original_indices for b in norm]),
            )  # fmt: skip
        else:
            gene_intervals = Intervals(np.empty(0, np.int32), np.empty(0, np.int32), np.empty(0, np.int8))

        locus_lengths = np.array([len(x) for x in loci], dtype=np.int32)
        locus_offsets = np.zeros(n_loci, dtype=np.int32)
        if n_loci > 1:
            np.cumsum(locus_lengths[:-1], out=locus_offsets[1:])

        batches = []
        if n_loci > 1:
            targets = [RandstrobeIndex.build(x, k=self.k, s=self.s, sort_by_hash=True) for x in loci]
            queries = [RandstrobeIndex.build(x, k=self.k, s=self.s, sort_by_hash=False) for x in loci]
            for i in range(n_loci):
                for j in range(i + 1, n_loci):
                    seeds = targets[j].top_hits(queries[i], min_score=self.min_score)
                    if len(seeds) == 0:
                        continue
                    qi, ti = seeds.query_indices.astype(np.int32), seeds.target_indices.astype(np.int32)
                    batches.append(LocusComparisonEdges(
                        np.full(len(seeds), i, np.int32), np.full(len(seeds), j, np.int32), qi, ti,
                        qi + locus_offsets[i], ti + locus_offsets[j],
                        alignments=self.aligner.align_seeds(loci[i], loci[j], seeds),
                    ))  # fmt: skip
        return LocusComparisons(
            edges=LocusComparisonEdges.concat(batches), locus_names=tuple(inp.name for inp in inputs),
            locus_lengths=locus_lengths, locus_offsets=locus_offsets, gene_names=np.array(global_seqs.ids, dtype=object),
            gene_descriptions=gene_descriptions, gene_states=gene_states, gene_intervals=gene_intervals,
        )  # fmt: skip
