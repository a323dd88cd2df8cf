"""Strand / Interval / Intervals: 1-D interval SoA with the two interval reductions the typing path uses.

Mirrors the interface of the reference's ``kaptive.core.interval`` (src/kaptive/core/interval.py:25-493) for the
members the typing path touches. The two kernels the reference JIT-compiles with numba are restated here in numpy:

* greedy overlap cull  -- reference ``_cull_overlaps_kernel`` (src/kaptive/core/interval.py:698-751)
* single-linkage 1-D clustering -- reference ``_cluster_kernel`` (src/kaptive/core/interval.py:595-639)

The batched GPU path runs the same two reductions inside ``kp_reduce.hip``; these host versions serve the
single-genome API and are what the GPU versions are tested against.
"""

from __future__ import annotations

from dataclasses import dataclass
from enum import IntEnum
from typing import Any, Iterable

import numpy as np


class Strand(IntEnum):
    FORWARD = 1
    REVERSE = -1
    UNSTRANDED = 0

    @classmethod
    def _missing_(cls, value: object) -> "Strand":
        if isinstance(value, bytes):
            value = value.decode("ascii")
        if value in ("+", "1", "+1"):
            return cls.FORWARD
        if value in ("-", "-1"):
            return cls.REVERSE
        return cls.UNSTRANDED

    def __str__(self) -> str:
        return {1: "+", -1: "-"}.get(int(self), ".")


@dataclass(frozen=True, slots=True)
class Interval:
    start: int
    end: int
    strand: Strand = Strand.UNSTRANDED

    def __len__(self) -> int:
        return self.end - self.start

    def __contains__(self, item: Any) -> bool:
        if isinstance(item, (int, np.integer)):
            return self.start <= item < self.end
        other = Interval.from_item(item)
        return self.start <= other.start and other.end <= self.end

    def __add__(self, other: Any) -> "Interval":
        o = Interval.from_item(other)
        strand = self.strand if self.strand == o.strand else Strand.UNSTRANDED
        return Interval(min(self.start, o.start), max(self.end, o.end), strand)

    __radd__ = __add__

    def shift(self, x: int, y: int | None = None) -> "Interval":
        return Interval(self.start + x, self.end + (x if y is None else y), self.strand)

    def expand(self, left: int, right: int, clip_length: int | None = None) -> "Interval":
        end = self.end + right
        return Interval(max(0, self.start - left), end if clip_length is None else min(end, clip_length), self.strand)

    def reverse_complement(self, length: int | None = None) -> "Interval":
        n = self.end if length is None else length
        return Interval(n - self.end, n - self.start, Strand(-int(self.strand)))

    @classmethod
    def from_item(cls, item: Any, strand: Strand = Strand.UNSTRANDED, length: int | None = None) -> "Interval":
        if isinstance(item, cls):
            return item
        if isinstance(item, (int, np.integer)):
            i = int(item) + (length if (item < 0 and length is not None) else 0)
            return cls(i, i + 1, strand)
        if isinstance(item, slice):
            start = 0 if item.start is None else item.start
            stop = length if item.stop is None else item.stop
            if stop is None:
                raise ValueError("slice without stop needs 'length'")
            return cls(stop + 1, start + 1, strand) if item.step == -1 else cls(start, stop, strand)
        if hasattr(item, "start") and callable(item.start):  # re.Match
            return cls(item.start(), item.end(), strand)
        raise TypeError(item)


@dataclass(frozen=True, slots=True)
class Intervals:
    starts: np.ndarray  # int32
    ends: np.ndarray  # int32
    strands: np.ndarray  # int8
    original_indices: np.ndarray | None = None

    def __post_init__(self) -> None:
        if self.original_indices is None:
            object.__setattr__(self, "original_indices", np.arange(len(self.starts), dtype=np.int32))

    def __len__(self) -> int:
        return len(self.starts)

    def __getitem__(self, item: Any) -> "Interval | Intervals":
        if isinstance(item, (int, np.integer)):
            n = len(self)
            i = item + n if item < 0 else item
            if not 0 <= i < n:
                raise IndexError("Batch index out of range")
            return Interval(self.starts[i], self.ends[i], self.strands[i])
        return Intervals(self.starts[item], self.ends[item], self.strands[item], self.original_indices[item])

    @classmethod
    def empty(cls) -> "Intervals":
        z = np.empty(0, dtype=np.int32)
        return cls(z, z.copy(), np.empty(0, dtype=np.int8), z.copy())

    @classmethod
    def from_intervals(cls, intervals: Iterable[Interval]) -> "Intervals":
        rows = [(i.start, i.end, int(i.strand)) for i in intervals]
        if not rows:
            return cls.empty()
        a = np.asarray(rows, dtype=np.int64)
        return cls(a[:, 0].astype(np.int32), a[:, 1].astype(np.int32), a[:, 2].astype(np.int8))

    @classmethod
    def concat(cls, batches: Iterable["Intervals"]) -> "Intervals":
        bs = list(batches)
        if not bs:
            raise ValueError("Cannot concatenate empty list of batches")
        return cls(
            np.concatenate([b.starts for b in bs]),
            np.concatenate([b.ends for b in bs]),
            np.concatenate([b.strands for b in bs]),
            np.concatenate([b.original_indices for b in bs]),
        )

    def to_dict(self) -> dict[str, list]:
        return {"starts": self.starts.tolist(), "ends": self.ends.tolist(), "strands": self.strands.tolist()}

    @classmethod
    def from_dict(cls, d: dict) -> "Intervals":
        return cls(np.array(d["starts"], np.int32), np.array(d["ends"], np.int32), np.array(d["strands"], np.int8))

    def shift(self, x: Any, y: Any = None) -> "Intervals":
        if len(self) == 0:
            return self
        return Intervals(
            np.asarray(self.starts + x, dtype=np.int32),
            np.asarray(self.ends + (x if y is None else y), dtype=np.int32),
            self.strands,
            self.original_indices,
        )

    def arrange(self, indices: np.ndarray, order: np.ndarray, starts: np.ndarray, ends: np.ndarray, strands: np.ndarray,
                gap: int = 500) -> "Intervals":
        """Lay intervals that live on several pieces (contig fragments of a locus) out on one axis: pieces in ``order``,
        ``gap`` apart, reverse-strand pieces mirrored (reference: src/kaptive/core/interval.py:529-591).
        ``indices[i]`` = piece of interval i."""
        if len(self) == 0:
            return self
        starts, ends = np.asarray(starts, np.int64), np.asarray(ends, np.int64)
        plot_start = np.zeros(len(starts), np.int64)
        x = 0
        for i in np.asarray(order).tolist():
            plot_start[i] = x
            x += int(ends[i] - starts[i]) + gap
        piece = np.asarray(indices)
        placed = (piece >= 0) & (piece < len(starts))  # an interval of no piece stays at 0, as in the reference
        p = np.where(placed, piece, 0)
        forward = np.asarray(strands)[p] >= 0
        new_starts = np.where(forward, plot_start[p] + (self.starts - starts[p]), plot_start[p] + (ends[p] - self.ends))
        new_ends = np.where(forward, plot_start[p] + (self.ends - starts[p]), plot_start[p] + (ends[p] - self.starts))
        new_strands = np.where(forward, self.strands, -self.strands)
        zero = ~placed
        return Intervals(
            np.where(zero, 0, new_starts).astype(np.int32), np.where(zero, 0, new_ends).astype(np.int32),
            np.where(zero, 0, new_strands).astype(np.int8), self.original_indices,
        )  # fmt: skip

    # -- reductions ---------------------------------------------------------------------------------------------
    def cull_overlaps(
        self,
        order: np.ndarray,
        max_overlap_fraction: float = 0.1,
        group_by: np.ndarray | None = None,
        secondary_group_by: np.ndarray | None = None,
    ) -> np.ndarray:
        """Greedy cull: visit intervals in ``order``; keep one unless it overlaps an already-kept interval of the
        same (group, secondary group) by more than ``max_overlap_fraction`` of the shorter of the two."""
        n = len(self)
        kept = np.zeros(n, dtype=np.bool_)
        if n == 0:
            return kept
        g1 = np.zeros(n, np.int64) if group_by is None else np.asarray(group_by, dtype=np.int64)
        g2 = np.zeros(n, np.int64) if secondary_group_by is None else np.asarray(secondary_group_by, dtype=np.int64)
        s = self.starts.astype(np.int64)
        e = self.ends.astype(np.int64)
        # kept intervals are appended to dense arrays so each candidate is one vectorised comparison
        ks = np.empty(n, np.int64)
        ke = np.empty(n, np.int64)
        kg1 = np.empty(n, np.int64)
        kg2 = np.empty(n, np.int64)
        nk = 0
        for idx in np.asarray(order):
            length = e[idx] - s[idx]
            if length <= 0:
                continue
            if nk:
                ov = np.minimum(e[idx], ke[:nk]) - np.maximum(s[idx], ks[:nk])
                shorter = np.minimum(length, ke[:nk] - ks[:nk])
                with np.errstate(divide="ignore", invalid="ignore"):
                    clash = (kg1[:nk] == g1[idx]) & (kg2[:nk] == g2[idx]) & (ov > 0) & (
                        ov / shorter > max_overlap_fraction
                    )
                if clash.any():
                    continue
            kept[idx] = True
            ks[nk], ke[nk], kg1[nk], kg2[nk] = s[idx], e[idx], g1[idx], g2[idx]
            nk += 1
        return kept

    def cluster_spatial(self, tolerance: int = 0, group_by: np.ndarray | None = None) -> np.ndarray:
        """Single-linkage clustering along the axis within each group; returns int32 cluster ids (0-based, in
        (group, start, end) order)."""
        n = len(self)
        ids = np.empty(n, dtype=np.int32)
        if n == 0:
            return ids
        g = np.zeros(n, np.int32) if group_by is None else np.asarray(group_by, dtype=np.int32)
        order = np.lexsort((self.ends, self.starts, g))
        s, e, gg = self.starts[order].astype(np.int64), self.ends[order].astype(np.int64), g[order]
        # running max of ends restarts at each group change; a new cluster opens where start exceeds it + tolerance
        cur = 0
        cur_e, cur_g = e[0], gg[0]
        out = np.empty(n, dtype=np.int32)
        out[0] = 0
        for i in range(1, n):
            if gg[i] == cur_g and s[i] <= cur_e + tolerance:
                if e[i] > cur_e:
                    cur_e = e[i]
            else:
                cur += 1
                cur_e, cur_g = e[i], gg[i]
            out[i] = cur
        ids[order] = out
        return ids
