"""Alignments: SoA table of gene-vs-contig hits.

Column names follow the reference's ``kaptive.core.alignment.Alignments`` (src/kaptive/core/alignment.py:262-317) so
code written against it keeps working.  The way in is ``from_hit_table``: the HIP aligner already returns columns,
nothing is parsed per hit (the reference loops over hit objects, alignment.py:392-474).  The aligner produces no CIGAR
strings -- the reference parses them and never reads them on the typing path -- so the ``cigars``
column exists for shape compatibility only and is always empty; the reference's CIGAR parser, ``swap_sides`` and ``best``
have no counterpart here.

The reductions used by typing (``q_covs``, ``cull_overlaps``, ``is_partial``) follow
src/kaptive/core/alignment.py:355-367, 643-686 and 774-809.
"""

from __future__ import annotations

from dataclasses import dataclass, fields
from typing import Any, Iterable, NamedTuple

import numpy as np

from kaptive_amd.core.interval import Intervals, Strand

def _ragged_take(data: np.ndarray, offsets: np.ndarray, lengths: np.ndarray, idx: np.ndarray):
    new_len = lengths[idx]
    new_off = np.zeros(len(new_len), dtype=np.int32)
    if len(new_len) > 1:
        np.cumsum(new_len[:-1], out=new_off[1:])
    parts = [data[offsets[i] : offsets[i] + lengths[i]] for i in idx]
    return (np.concatenate(parts) if parts else data[:0]), new_off, new_len


@dataclass(frozen=True, slots=True)
class Cigars:
    data: np.ndarray  # uint32
    offsets: np.ndarray  # int32
    lengths: np.ndarray  # int32

    def __len__(self) -> int:
        return len(self.offsets)

    def __getitem__(self, item: Any) -> "np.ndarray | Cigars":
        if isinstance(item, (int, np.integer)):
            i = item + len(self) if item < 0 else item
            if not 0 <= i < len(self):
                raise IndexError("Batch index out of range")
            return self.data[self.offsets[i] : self.offsets[i] + self.lengths[i]]
        if isinstance(item, slice):
            idx = np.arange(len(self))[item]
        else:
            idx = np.asarray(item)
            if idx.dtype.kind == "b":
                idx = np.nonzero(idx)[0]
        if len(idx) == 0:
            return self.empty()
        if len(self.data) == 0:  # native path carries no CIGARs: nothing to gather
            return self.empty(len(idx))
        return Cigars(*_ragged_take(self.data, self.offsets, self.lengths, idx))

    @classmethod
    def empty(cls, n: int = 0) -> "Cigars":
        return cls(np.empty(0, np.uint32), np.zeros(n, np.int32), np.zeros(n, np.int32))

    @classmethod
    def from_lists(cls, cigars: list[np.ndarray]) -> "Cigars":
        if not cigars:
            return cls.empty()
        lengths = np.array([len(c) for c in cigars], dtype=np.int32)
        offsets = np.zeros(len(lengths), dtype=np.int32)
        if len(lengths) > 1:
            np.cumsum(lengths[:-1], out=offsets[1:])
        return cls(np.concatenate(cigars).astype(np.uint32), offsets, lengths)

    @classmethod
    def concat(cls, batches: Iterable["Cigars"]) -> "Cigars":
        return cls.from_lists([b[i] for b in batches for i in range(len(b))])

class Alignment(NamedTuple):
    idx: int
    q_name: str
    q_length: int
    q_start: int
    q_end: int
    t_name: str
    t_length: int
    t_start: int
    t_end: int
    strand: Strand
    length: int
    match: int
    mismatch: int
    score: int
    quality: int
    cigar: np.ndarray
    is_primary: bool
    is_supplementary: bool
    is_spliced: bool
    divergence: float
    cs: bytes | None
    md: bytes | None


_DTYPES = {
    "q_name_ids": np.int32, "q_lengths": np.int32, "q_starts": np.int32, "q_ends": np.int32,
    "t_name_ids": np.int32, "t_lengths": np.int32, "t_starts": np.int32, "t_ends": np.int32,
    "strands": np.int8, "lengths": np.int32, "matches": np.int32, "mismatches": np.int32,
    "scores": np.int32, "qualities": np.uint8, "is_primary": np.bool_, "is_supplementary": np.bool_,
    "is_spliced": np.bool_, "divergence": np.float64, "cs": object, "md": object,
}  # fmt: skip


@dataclass(frozen=True, slots=True)
class Alignments:
    q_name_ids: np.ndarray
    q_names_dict: tuple[str, ...]
    q_lengths: np.ndarray
    q_starts: np.ndarray
    q_ends: np.ndarray
    t_name_ids: np.ndarray
    t_names_dict: tuple[str, ...]
    t_lengths: np.ndarray
    t_starts: np.ndarray
    t_ends: np.ndarray
    strands: np.ndarray
    lengths: np.ndarray
    matches: np.ndarray
    mismatches: np.ndarray
    scores: np.ndarray
    qualities: np.ndarray
    cigars: Cigars
    is_primary: np.ndarray
    is_supplementary: np.ndarray
    is_spliced: np.ndarray
    divergence: np.ndarray
    cs: np.ndarray
    md: np.ndarray

    def __len__(self) -> int:
        return len(self.q_starts)

    # -- construction -------------------------------------------------------------------------------------------
    @classmethod
    def empty(cls) -> "Alignments":
        cols = {k: np.empty(0, dtype=dt) for k, dt in _DTYPES.items()}
        return cls(q_names_dict=(), t_names_dict=(), cigars=Cigars.empty(), **cols)

    @classmethod
    def from_hit_table(
        cls,
        q_names: tuple[str, ...],
        t_names: tuple[str, ...],
        *,
        q_ids: np.ndarray,
        q_lengths: np.ndarray,
        q_starts: np.ndarray,
        q_ends: np.ndarray,
        t_ids: np.ndarray,
        t_lengths: np.ndarray,
        t_starts: np.ndarray,
        t_ends: np.ndarray,
        strands: np.ndarray,
        block_lens: np.ndarray,
        matches: np.ndarray,
        scores: np.ndarray,
        mapqs: np.ndarray,
    ) -> "Alignments":
        """Wrap columns produced by the native aligner. ``q_ids``/``t_ids`` index ``q_names``/``t_names`` directly
        (the native path numbers contigs by their order in the assembly, not by first appearance in the hits)."""
        n = len(q_ids)
        block_lens = np.asarray(block_lens, np.int32)
        matches = np.asarray(matches, np.int32)
        with np.errstate(divide="ignore", invalid="ignore"):
            div = np.where(block_lens > 0, (block_lens - matches) / np.maximum(block_lens, 1), 0.0)
        primary = np.asarray(mapqs) > 0
        return cls(
            q_name_ids=np.asarray(q_ids, np.int32), q_names_dict=tuple(q_names),
            q_lengths=np.asarray(q_lengths, np.int32), q_starts=np.asarray(q_starts, np.int32),
            q_ends=np.asarray(q_ends, np.int32),
            t_name_ids=np.asarray(t_ids, np.int32), t_names_dict=tuple(t_names),
            t_lengths=np.asarray(t_lengths, np.int32), t_starts=np.asarray(t_starts, np.int32),
            t_ends=np.asarray(t_ends, np.int32),
            strands=np.asarray(strands, np.int8), lengths=block_lens, matches=matches,
            mismatches=(block_lens - matches).astype(np.int32), scores=np.asarray(scores, np.int32),
            qualities=np.asarray(mapqs, np.uint8), cigars=Cigars.empty(n),
            is_primary=primary, is_supplementary=np.zeros(n, np.bool_), is_spliced=np.zeros(n, np.bool_),
            divergence=div.astype(np.float64), cs=np.full(n, None, dtype=object), md=np.full(n, None, dtype=object),
        )  # fmt: skip

    @classmethod
    def from_records(cls, records: Iterable[Alignment]) -> "Alignments":
        recs = list(records)
        if not recs:
            return cls.empty()
        q_index: dict[str, int] = {}
        t_index: dict[str, int] = {}
        qi = [q_index.setdefault(r.q_name, len(q_index)) for r in recs]
        ti = [t_index.setdefault(r.t_name, len(t_index)) for r in recs]
        per_field = {
            "q_name_ids": qi, "q_lengths": [r.q_length for r in recs], "q_starts": [r.q_start for r in recs],
            "q_ends": [r.q_end for r in recs], "t_name_ids": ti, "t_lengths": [r.t_length for r in recs],
            "t_starts": [r.t_start for r in recs], "t_ends": [r.t_end for r in recs],
            "strands": [int(r.strand) for r in recs], "lengths": [r.length for r in recs],
            "matches": [r.match for r in recs], "mismatches": [r.mismatch for r in recs],
            "scores": [r.score for r in recs], "qualities": [r.quality for r in recs],
            "is_primary": [r.is_primary for r in recs], "is_supplementary": [r.is_supplementary for r in recs],
            "is_spliced": [r.is_spliced for r in recs], "divergence": [r.divergence for r in recs],
            "cs": [r.cs for r in recs], "md": [r.md for r in recs],
        }  # fmt: skip
        cols = {k: np.array(v, dtype=_DTYPES[k]) for k, v in per_field.items()}
        return cls(
            q_names_dict=tuple(q_index), t_names_dict=tuple(t_index),
            cigars=Cigars.from_lists([np.asarray(r.cigar, np.uint32) for r in recs]), **cols,
        )  # fmt: skip

    @classmethod
    def concat(cls, batches: Iterable["Alignments"]) -> "Alignments":
        bs = list(batches)
        if not bs:
            raise ValueError("Cannot concatenate an empty iterable of batches")
        q_index: dict[str, int] = {}
        t_index: dict[str, int] = {}
        q_ids, t_ids = [], []
        for b in bs:
            q_map = np.array([q_index.setdefault(n, len(q_index)) for n in b.q_names_dict], dtype=np.int32)
            t_map = np.array([t_index.setdefault(n, len(t_index)) for n in b.t_names_dict], dtype=np.int32)
            q_ids.append(q_map[b.q_name_ids] if len(b) else b.q_name_ids)
            t_ids.append(t_map[b.t_name_ids] if len(b) else b.t_name_ids)
        cols = {k: np.concatenate([getattr(b, k) for b in bs]) for k in _DTYPES if not k.endswith("name_ids")}
        return cls(
            q_name_ids=np.concatenate(q_ids), t_name_ids=np.concatenate(t_ids), q_names_dict=tuple(q_index),
            t_names_dict=tuple(t_index), cigars=Cigars.concat([b.cigars for b in bs]), **cols,
        )  # fmt: skip

    # -- access -------------------------------------------------------------------------------------------------
    def __getitem__(self, item: Any) -> "Alignment | Alignments":
        if isinstance(item, (int, np.integer)):
            i = item + len(self) if item < 0 else item
            if not 0 <= i < len(self):
                raise IndexError("Batch index out of range")
            return Alignment(
                i, self.q_names_dict[self.q_name_ids[i]], self.q_lengths[i], self.q_starts[i], self.q_ends[i],
                self.t_names_dict[self.t_name_ids[i]], self.t_lengths[i], self.t_starts[i], self.t_ends[i],
                Strand(self.strands[i]), self.lengths[i], self.matches[i], self.mismatches[i], self.scores[i],
                self.qualities[i], self.cigars[i], self.is_primary[i], self.is_supplementary[i],
                self.is_spliced[i], self.divergence[i], self.cs[i], self.md[i],
            )  # fmt: skip
        cols = {k: getattr(self, k)[item] for k in _DTYPES}
        return Alignments(
            q_names_dict=self.q_names_dict, t_names_dict=self.t_names_dict, cigars=self.cigars[item], **cols
        )

    @property
    def q_names(self) -> np.ndarray:
        return np.array([self.q_names_dict[i] for i in self.q_name_ids], dtype=object)

    @property
    def t_names(self) -> np.ndarray:
        return np.array([self.t_names_dict[i] for i in self.t_name_ids], dtype=object)

    @property
    def q_aln_lens(self) -> np.ndarray:
        return self.q_ends - self.q_starts

    @property
    def t_aln_lens(self) -> np.ndarray:
        return self.t_ends - self.t_starts

    @staticmethod
    def _fraction(num: np.ndarray, den: np.ndarray) -> np.ndarray:
        return np.divide(num, den, out=np.zeros(len(den), dtype=np.float64), where=den > 0)

    @property
    def q_covs(self) -> np.ndarray:
        return self._fraction(self.q_aln_lens, self.q_lengths)

    @property
    def t_covs(self) -> np.ndarray:
        return self._fraction(self.t_aln_lens, self.t_lengths)

    def to_intervals(self, by_query: bool = False) -> Intervals:
        s, e = (self.q_starts, self.q_ends) if by_query else (self.t_starts, self.t_ends)
        return Intervals(s, e, self.strands, np.arange(len(self), dtype=np.int32))

    # -- reductions ---------------------------------------------------------------------------------------------
    def cull_order(self, priority_mask: np.ndarray | None = None) -> np.ndarray:
        """Visit order of the overlap cull: score (+1e9 when prioritised) desc, matches desc, then the uint8-wrapped
        negated MAPQ (reference: src/kaptive/core/alignment.py:669-675)."""
        scores = self.scores.astype(np.float64)
        if priority_mask is not None:
            scores[priority_mask] += 1e9
        return np.lexsort((-self.qualities, -self.matches, -scores)).astype(np.int32)

    def cull_overlaps(
        self,
        max_overlap_fraction: float = 0.1,
        group_by: np.ndarray | None = None,
        priority_mask: np.ndarray | None = None,
        by_query: bool = True,
    ) -> "Alignments":
        n = len(self)
        if n < 2:
            return self
        names = self.q_name_ids if by_query else self.t_name_ids
        kept = self.to_intervals(by_query=by_query).cull_overlaps(
            order=self.cull_order(priority_mask),
            max_overlap_fraction=max_overlap_fraction,
            group_by=names,
            secondary_group_by=np.zeros(n, np.int32) if group_by is None else group_by,
        )
        return self[kept]  # type: ignore[return-value]

    def is_partial_left(self, edge_tolerance: int = 0) -> np.ndarray:
        clipped = np.where(self.strands == 1, self.q_starts > 0, self.q_ends < self.q_lengths)
        return (self.t_starts <= edge_tolerance) & clipped

    def is_partial_right(self, edge_tolerance: int = 0) -> np.ndarray:
        clipped = np.where(self.strands == 1, self.q_ends < self.q_lengths, self.q_starts > 0)
        return (self.t_ends >= self.t_lengths - edge_tolerance) & clipped

    def is_partial(self, edge_tolerance: int = 0) -> np.ndarray:
        return self.is_partial_left(edge_tolerance) | self.is_partial_right(edge_tolerance)


assert [f.name for f in fields(Alignments) if f.name in _DTYPES] == [k for k in _DTYPES]
