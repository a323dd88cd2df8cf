"""PairwiseAlignments container and the batched protein aligner front end.

Interface of the reference's ``kaptive.core.pairwise`` (src/kaptive/core/pairwise.py:28-339).  The reference runs
its banded Smith-Waterman-Gotoh (BLOSUM62, gap 11/1, band ``max(20, |len difference| + 1)``, full traceback) as a numba
kernel (pairwise.py:395-584); here ``PairwiseAligner.__call__`` launches the HIP kernel ``kp_prot.hip`` through the
C-ABI.  There is no CPU implementation in the product: without the HIP library the call raises.
"""

from __future__ import annotations

from dataclasses import dataclass
from typing import Any, Iterable

import numpy as np

from kaptive_amd.core.seq import Sequences

_COLS = ("scores", "matches", "mismatches", "gaps", "q_starts", "q_ends", "t_starts", "t_ends")


@dataclass(frozen=True, slots=True)
class PairwiseAlignment:
    score: int
    matches: int
    mismatches: int
    gaps: int
    q_start: int
    q_end: int
    t_start: int
    t_end: int

    @property
    def pident(self) -> float:
        total = self.matches + self.mismatches + self.gaps
        return (self.matches / total) * 100.0 if total > 0 else 0.0


@dataclass(frozen=True, slots=True)
class PairwiseAlignments:
    scores: np.ndarray
    matches: np.ndarray
    mismatches: np.ndarray
    gaps: np.ndarray
    q_starts: np.ndarray
    q_ends: np.ndarray
    t_starts: np.ndarray
    t_ends: np.ndarray

    def __len__(self) -> int:
        return len(self.scores)

    def __getitem__(self, item: Any) -> "PairwiseAlignment | PairwiseAlignments":
        if isinstance(item, (int, np.integer)):
            i = item + len(self) if item < 0 else item
            if not 0 <= i < len(self):
                raise IndexError("Batch index out of range")
            return PairwiseAlignment(*(int(getattr(self, c)[i]) for c in _COLS))
        return PairwiseAlignments(*(getattr(self, c)[item] for c in _COLS))

    @classmethod
    def empty(cls) -> "PairwiseAlignments":
        return cls(*(np.empty(0, dtype=np.int32) for _ in _COLS))

    @classmethod
    def from_table(cls, table: np.ndarray) -> "PairwiseAlignments":
        """``table`` int32 [n, 8] in column order score, matches, mismatches, gaps, qs, qe, ts, te."""
        t = np.ascontiguousarray(table, dtype=np.int32).reshape(-1, 8)
        return cls(*(t[:, i].copy() for i in range(8)))

    @classmethod
    def concat(cls, batches: Iterable["PairwiseAlignments"]) -> "PairwiseAlignments":
        bs = list(batches)
        return cls(*(np.concatenate([getattr(b, c) for b in bs]) for c in _COLS)) if bs else cls.empty()

    def to_dict(self) -> dict[str, np.ndarray]:
        return {c: getattr(self, c) for c in _COLS}

    @classmethod
    def from_dict(cls, d: dict[str, Any]) -> "PairwiseAlignments":
        return cls(*(np.array(d[c], dtype=np.int32) for c in _COLS))

    @property
    def pidents(self) -> np.ndarray:
        total = self.matches + self.mismatches + self.gaps
        return np.divide(self.matches * 100.0, total, out=np.zeros(len(self), dtype=np.float64), where=total > 0)


@dataclass(frozen=True, slots=True)
class PairwiseAligner:
    """Batched protein aligner on the GPU (reference: src/kaptive/core/pairwise.py:240-339).  The gap penalties are
    built into the kernel (11 / 1, the reference's defaults); ``k`` is free in the seeded mode and 20 otherwise."""

    gap_open: int = 11
    gap_extend: int = 1
    k: int = 20
    device: int = 0

    def __call__(self, queries: Sequences, targets: Sequences, seeds: Any = None) -> PairwiseAlignments:
        if len(queries.offsets) != len(targets.offsets):
            raise ValueError("Query and target batches must have the same number of sequences.")
        if (self.gap_open, self.gap_extend) != (11, 1) or (seeds is None and self.k != 20):
            raise NotImplementedError("the HIP kernel is built for gap 11/1 (and k=20 in the unseeded mode)")
        if len(queries.offsets) == 0:
            return PairwiseAlignments.empty()
        from kaptive_amd import _native

        ctx = _native.default_context(self.device)
        if seeds is not None:  # band k around each pair's seed diagonal (pairwise.py:449-451)
            table = ctx.protein_align_seeded(queries.seqs, queries.offsets, queries.lengths, targets.seqs, targets.offsets,
                                             targets.lengths, seeds.offsets, self.k)  # fmt: skip
        else:
            table = ctx.protein_align(queries.seqs, queries.offsets, queries.lengths, targets.seqs, targets.offsets,
                                      targets.lengths)  # fmt: skip
        return PairwiseAlignments.from_table(table)

    def align_seeds(self, queries: Sequences, targets: Sequences, seeds: Any) -> PairwiseAlignments:
        """One alignment per seed: query ``seeds.query_indices[i]`` against target ``seeds.target_indices[i]``
        (pairwise.py:327-339)."""
        paired_queries, paired_targets = seeds.extract_sequences(queries, targets)
        return self(paired_queries, paired_targets, seeds)
