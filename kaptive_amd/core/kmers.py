"""Randstrobe seeds for protein-vs-protein comparison.

Interface of the part of the reference's ``kaptive.core.kmers`` that ``compare.LocusComparator`` uses
(src/kaptive/core/kmers.py: ``Seeds`` 62-273, ``RandstrobeIndex`` 536-655, ``BaseKmerIndex.top_hits`` 356-379).  The
numba kernels behind them are restated natively (csrc/kp_kmers.cpp via ``_native.randstrobes`` /
``_native.randstrobe_top_hits``); the FracMinHash index, which the typing path never reads, is
not part of this package.
"""

from __future__ import annotations

from dataclasses import dataclass
from functools import cache
from typing import Any, Iterable

import numpy as np

from kaptive_amd import _native
from kaptive_amd.core.seq import Sequences

RANDSTROBE_DTYPE = _native.RANDSTROBE_DTYPE


@cache
def mmseqs12_lut(fill_value: int = 12) -> np.ndarray:
    """256-byte table of the 12-letter reduced amino-acid alphabet (kmers.py:660-694); unknown residues -> 12."""
    groups = ("AST", "LM", "IV", "KR", "EQ", "ND", "FY", "C", "G", "H", "P", "W")
    lut = np.full(256, fill_value, dtype=np.uint8)
    for value, letters in enumerate(groups):
        for ch in letters:
            lut[ord(ch)] = lut[ord(ch.lower())] = value
    lut.flags.writeable = False
    return lut


@dataclass(frozen=True, slots=True)
class Seeds:
    """One seed per row: query sequence, target sequence, shared-record count, diagonal offset (query - target)."""

    query_indices: np.ndarray  # uint32
    target_indices: np.ndarray  # uint32
    scores: np.ndarray  # uint32
    offsets: np.ndarray  # int32

    def __len__(self) -> int:
        return len(self.query_indices)

    @classmethod
    def empty(cls) -> "Seeds":
        return cls(np.empty(0, np.uint32), np.empty(0, np.uint32), np.empty(0, np.uint32), np.empty(0, np.int32))

    def filter(self, mask: np.ndarray) -> "Seeds":
        return Seeds(self.query_indices[mask], self.target_indices[mask], self.scores[mask], self.offsets[mask])

    def __getitem__(self, item: Any) -> "Seeds":
        if isinstance(item, slice):
            idx = np.arange(len(self))[item]
        else:
            arr = np.asarray(item)
            idx = np.nonzero(arr)[0] if arr.dtype.kind == "b" else arr
        return Seeds(self.query_indices[idx], self.target_indices[idx], self.scores[idx], self.offsets[idx])

    @classmethod
    def concat(cls, batches: Iterable["Seeds"]) -> "Seeds":
        bs = list(batches)
        if not bs:
            return cls.empty()
        return cls(*(np.concatenate([getattr(b, f) for b in bs]) for f in ("query_indices", "target_indices", "scores", "offsets")))

    def extract_sequences(self, queries: Sequences, targets: Sequences) -> tuple[Sequences, Sequences]:
        return queries[self.query_indices], targets[self.target_indices]


@dataclass(frozen=True, slots=True, kw_only=True)
class RandstrobeIndex:
    """Syncmer-linked order-2 randstrobes of a batch of proteins (kmers.py:536-655)."""

    records: np.ndarray
    n_seqs: int = 0
    is_sorted: bool = False
    k: int = 10
    s: int = 5
    w_min: int = 1
    w_max: int = 5
    lut: np.ndarray | None = None

    def __len__(self) -> int:
        return len(self.records)

    @classmethod
    def empty(cls) -> "RandstrobeIndex":
        return cls(records=np.empty(0, dtype=RANDSTROBE_DTYPE))

    @classmethod
    def build(cls, batch: Sequences, k: int = 10, s: int = 5, w_min: int = 1, w_max: int = 5, sort_by_hash: bool = False,
              lut: np.ndarray | None = None, **_: Any) -> "RandstrobeIndex":  # fmt: skip
        if s >= k:
            raise ValueError("Sub-k-mer size (s) must be strictly less than k-mer size (k).")
        if len(batch) == 0:
            return cls.empty()
        records = _native.randstrobes(batch.seqs, batch.offsets, batch.lengths, lut if lut is not None else mmseqs12_lut(),
                                      k, s, w_min, w_max, sort_by_hash)  # fmt: skip
        if len(records) == 0:
            return cls.empty()
        return cls(records=records, n_seqs=len(batch), is_sorted=sort_by_hash, k=k, s=s, w_min=w_min, w_max=w_max, lut=lut)

    def top_hits(self, queries: "RandstrobeIndex | Sequences", min_score: int = 1) -> Seeds:
        """Best-matching sequence of this (hash-sorted) index for every query sequence (kmers.py:356-379)."""
        if len(queries) == 0 or len(self) == 0:
            return Seeds.empty()
        if isinstance(queries, Sequences):
            queries = self.build(queries, k=self.k, s=self.s, w_min=self.w_min, w_max=self.w_max, lut=self.lut)
            if len(queries) == 0:
                return Seeds.empty()
        if not self.is_sorted:
            raise ValueError("the target index must be built with sort_by_hash=True")
        best_t, score, off = _native.randstrobe_top_hits(queries.records, queries.n_seqs, self.records, self.n_seqs)
        seeds = Seeds(np.arange(queries.n_seqs, dtype=np.uint32), best_t, score, off)
        return seeds.filter(seeds.scores >= min_score) if min_score > 0 else seeds
