"""Result containers of the typing path (reference: src/kaptive/serotyping/models.py:31-654).

Same field names, dtypes and dict wire format as the reference so rows written by either side convert with
``kaptive convert``. The column lists below drive construction, slicing and (de)serialisation in one place.
"""

from __future__ import annotations

from dataclasses import dataclass, field
from enum import IntEnum, IntFlag
from typing import Any, ClassVar, Iterable

import numpy as np

from kaptive_amd.core.interval import Intervals
from kaptive_amd.core.seq import Sequences


class GeneState(IntEnum):
    NORMAL = 0
    PARTIAL = 1
    TRUNCATED = 2
    NOVEL = 3


class SerotypingProblem(IntFlag):
    NONE = 0
    FRAGMENTED = 1
    UNEXPECTED_GENES = 2
    MISSING_GENES = 4
    NOVEL_GENES = 8
    TRUNCATED_GENES = 16

    SYMBOLS: ClassVar[tuple[bytes, ...]]

    def to_symbols(self) -> bytes:
        return self.SYMBOLS[self.value]


# symbol string for every flag combination, in the fixed order ? + - * !
SerotypingProblem.SYMBOLS = tuple(
    b"".join(sym for bit, sym in enumerate((b"?", b"+", b"-", b"*", b"!")) if v >> bit & 1) for v in range(32)
)

_HIT_COLS: tuple[tuple[str, Any], ...] = (
    ("gene_indices", np.int32), ("q_starts", np.int32), ("q_ends", np.int32), ("t_indices", np.uint32),
    ("t_starts", np.int32), ("t_ends", np.int32), ("strands", np.int8), ("is_expected", np.bool_),
    ("is_inside", np.bool_), ("is_extra", np.bool_), ("expected_positions", np.int32),
    ("expected_strands", np.int8), ("gene_ids", "S32"), ("cluster_names", "S10"),
    ("product_descriptions", "S64"), ("coverages", np.float32),
)  # fmt: skip
_HIT_TEXT = {"gene_ids": "S32", "cluster_names": "S10", "product_descriptions": "S64"}


def _bytes_column(val: Any, dtype: str) -> np.ndarray:
    if isinstance(val, np.ndarray) and val.dtype.kind == "S":
        return val
    if val is None or len(val) == 0:
        return np.empty(0, dtype=dtype)
    flat = np.asarray(val, dtype=object).ravel() if not isinstance(val, np.ndarray) else val.ravel()
    return np.array([x.encode("utf-8") if isinstance(x, str) else x for x in flat], dtype=dtype)


@dataclass(slots=True, frozen=True)
class GeneHits:
    gene_indices: np.ndarray
    q_starts: np.ndarray
    q_ends: np.ndarray
    t_indices: np.ndarray
    t_starts: np.ndarray
    t_ends: np.ndarray
    strands: np.ndarray
    is_expected: np.ndarray
    is_inside: np.ndarray
    is_extra: np.ndarray
    expected_positions: np.ndarray
    expected_strands: np.ndarray
    gene_ids: np.ndarray  # S32 (longer names are cut)
    cluster_names: np.ndarray  # S10
    product_descriptions: np.ndarray  # S64
    coverages: np.ndarray

    def __post_init__(self) -> None:
        for name, dt in _HIT_TEXT.items():
            object.__setattr__(self, name, _bytes_column(getattr(self, name), dt))

    def __len__(self) -> int:
        return len(self.gene_indices)

    def __getitem__(self, item: Any) -> "GeneHits":
        return GeneHits(*(getattr(self, c)[item] for c, _ in _HIT_COLS))

    @classmethod
    def empty(cls) -> "GeneHits":
        return cls(*(np.empty(0, dtype=dt) for _, dt in _HIT_COLS))

    @classmethod
    def concat(cls, batches: Iterable["GeneHits"]) -> "GeneHits":
        bs = list(batches)
        return cls(*(np.concatenate([getattr(b, c) for b in bs]) for c, _ in _HIT_COLS)) if bs else cls.empty()

    @property
    def frames(self) -> np.ndarray:
        """Bases to skip so translation starts on a codon boundary of the reference gene."""
        return (-self.q_starts) % 3

    @property
    def query_lengths(self) -> np.ndarray:
        return self.q_ends - self.q_starts

    @property
    def target_lengths(self) -> np.ndarray:
        return self.t_ends - self.t_starts

    @property
    def q_intervals(self) -> Intervals:
        return Intervals(self.q_starts, self.q_ends, self.strands)

    @property
    def t_intervals(self) -> Intervals:
        return Intervals(self.t_starts, self.t_ends, self.strands)

    def to_dict(self) -> dict[str, Any]:
        d: dict[str, Any] = {c: getattr(self, c) for c, _ in _HIT_COLS if c not in _HIT_TEXT}
        for c in _HIT_TEXT:
            d[c] = np.char.decode(getattr(self, c), "utf-8").tolist()
        return d

    @classmethod
    def from_dict(cls, data: dict[str, Any]) -> "GeneHits":
        cols = []
        for c, dt in _HIT_COLS:
            v = data[c] if c in ("gene_indices", "q_starts", "q_ends", "t_indices", "t_starts", "t_ends", "strands",
                                 "is_expected", "is_inside", "is_extra") else data.get(c, [])  # fmt: skip
            cols.append(_bytes_column(v, dt) if c in _HIT_TEXT else np.array(v, dtype=dt))
        return cls(*cols)


_PIECE_COLS = (("ctg_indices", np.uint32), ("starts", np.int32), ("ends", np.int32), ("strands", np.int8))


@dataclass(slots=True, frozen=True)
class LocusPieces:
    ctg_indices: np.ndarray
    starts: np.ndarray
    ends: np.ndarray
    strands: np.ndarray

    def __len__(self) -> int:
        return len(self.ctg_indices)

    def __getitem__(self, item: Any) -> "LocusPieces":
        if isinstance(item, (int, np.integer)):
            raise NotImplementedError("Single item access not implemented for LocusPieces")
        return LocusPieces(*(getattr(self, c)[item] for c, _ in _PIECE_COLS))

    @classmethod
    def empty(cls) -> "LocusPieces":
        return cls(*(np.empty(0, dtype=dt) for _, dt in _PIECE_COLS))

    @classmethod
    def concat(cls, batches: Iterable["LocusPieces"]) -> "LocusPieces":
        bs = list(batches)
        return cls(*(np.concatenate([getattr(b, c) for b in bs]) for c, _ in _PIECE_COLS)) if bs else cls.empty()

    @classmethod
    def from_dict(cls, data: dict[str, Any]) -> "LocusPieces":
        return cls(*(np.array(data[c], dtype=dt) for c, dt in _PIECE_COLS))

    def to_dict(self) -> dict[str, Any]:
        return {c: getattr(self, c) for c, _ in _PIECE_COLS}


_SCALARS = (
    "kaptive_version", "database_name", "database_version", "database_organism", "database_taxon", "genome",
    "best_locus_idx", "best_locus_name", "best_locus_score", "best_locus_completeness", "length_discrepancy",
    "percent_identity", "percent_coverage", "phenotype", "typeable",
)  # fmt: skip


@dataclass(slots=True, frozen=True)
class SerotypingResult:
    kaptive_version: str
    database_name: str
    database_version: str
    database_organism: str
    database_taxon: int
    genome: str
    best_locus_idx: int
    best_locus_name: str
    best_locus_score: float
    best_locus_completeness: float
    locus_pieces: LocusPieces
    length_discrepancy: float
    locus_seqs: Sequences
    gene_hits: GeneHits
    gene_states: np.ndarray  # int8
    gene_seqs: Sequences
    translations: Sequences
    percent_identity: float
    percent_coverage: float
    protein_identities: np.ndarray  # float32
    phenotype: str
    typeable: bool
    missing_expected_genes: tuple[str, ...]
    problems: SerotypingProblem = field(init=False)

    def __post_init__(self) -> None:
        """Derive the problem flags (reference: src/kaptive/serotyping/models.py:538-558)."""
        h, st = self.gene_hits, self.gene_states
        inside = h.is_inside
        p = SerotypingProblem.NONE
        if len(self.locus_pieces) > 1:
            p |= SerotypingProblem.FRAGMENTED
        if np.any(inside & ~h.is_expected & ~h.is_extra):
            p |= SerotypingProblem.UNEXPECTED_GENES
        if self.best_locus_completeness < 1.0 or np.any(~inside & h.is_expected):
            p |= SerotypingProblem.MISSING_GENES
        if np.any(inside & (st == GeneState.NOVEL)):
            p |= SerotypingProblem.NOVEL_GENES
        if np.any(inside & ((st == GeneState.TRUNCATED) | (st == GeneState.PARTIAL))):
            p |= SerotypingProblem.TRUNCATED_GENES
        object.__setattr__(self, "problems", p)

    def to_dict(self) -> dict[str, Any]:
        d: dict[str, Any] = {k: getattr(self, k) for k in _SCALARS}
        d["missing_expected_genes"] = self.missing_expected_genes
        d["problems"] = self.problems
        d["locus_pieces"] = self.locus_pieces.to_dict()
        d["gene_hits"] = self.gene_hits.to_dict()
        d["gene_states"] = self.gene_states
        d["protein_identities"] = self.protein_identities
        for k in ("locus_seqs", "gene_seqs", "translations"):
            d[k] = getattr(self, k).to_dict()
        return d

    @classmethod
    def from_dict(cls, data: dict[str, Any]) -> "SerotypingResult":
        scalars = {k: data[k] for k in _SCALARS}
        for k in ("best_locus_score", "best_locus_completeness", "length_discrepancy", "percent_identity", "percent_coverage"):
            if scalars[k] is None:  # (the JSON writer, like orjson, has no NaN: null stands for it)
                scalars[k] = float("nan")
        return cls(
            **scalars,
            missing_expected_genes=tuple(data.get("missing_expected_genes", [])),
            locus_pieces=LocusPieces.from_dict(data["locus_pieces"]),
            gene_hits=GeneHits.from_dict(data["gene_hits"]),
            gene_states=np.array(data["gene_states"], dtype=np.int8),
            protein_identities=np.array(data["protein_identities"], dtype=np.float32),
            locus_seqs=Sequences.from_dict(data["locus_seqs"]),
            gene_seqs=Sequences.from_dict(data["gene_seqs"]),
            translations=Sequences.from_dict(data["translations"]),
        )
