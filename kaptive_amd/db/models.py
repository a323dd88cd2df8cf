"""Database metadata and phenotype-rule SoA (reference: src/kaptive/db/models.py:27-307)."""

from __future__ import annotations

import re
from dataclasses import dataclass, fields
from typing import Any, Iterable

import numpy as np


class DatabaseError(Exception):
    """Raised for unusable database files or metadata; the CLI maps it to exit code 1."""


_REQUIRED = ("name", "keyword", "genbank", "organism", "taxon", "antigen", "pathway", "version", "id_threshold",
             "doi", "owner", "repo", "branch", "contact")  # fmt: skip


@dataclass(frozen=True, slots=True)
class DatabaseMetadata:
    name: str
    keyword: str
    genbank: str
    organism: str
    taxon: int
    antigen: str
    pathway: str
    version: str
    id_threshold: float
    doi: list
    owner: str
    repo: str
    branch: str
    contact: dict
    phenotype_logic: dict
    antigenic_units: dict

    @property
    def parsed_version(self) -> tuple[int, ...]:
        return tuple(int(x) for x in re.findall(r"\d+", str(self.version)))

    @classmethod
    def from_dict(cls, data: dict) -> "DatabaseMetadata":
        if not isinstance(data, dict):
            raise DatabaseError("Metadata must be a dictionary.")
        missing = [k for k in _REQUIRED if k not in data]
        if missing:
            raise DatabaseError(f"Metadata is missing required field: {missing[0]!r}")
        kw = {k: data[k] for k in _REQUIRED}
        try:
            kw["taxon"] = int(kw["taxon"])
            kw["id_threshold"] = float(kw["id_threshold"])
        except ValueError as e:
            raise DatabaseError(f"Metadata has an invalid value type: {e}")
        kw["phenotype_logic"] = data.get("phenotype_logic", data.get("logic", {}))
        kw["antigenic_units"] = data.get("antigenic_units", data.get("units", {}))
        return cls(**kw)

    def to_dict(self) -> dict:
        return {f.name: getattr(self, f.name) for f in fields(self)}


@dataclass(frozen=True, slots=True)
class Phenotype:
    id: str
    loci: set
    extra_genes: set
    inactive_genes: set
    priority: int = 50
    as_suffix: bool = False


_PHENO_COLS = ("ids", "locus_masks", "extra_masks", "inactive_masks", "extra_counts", "priorities", "as_suffix")


@dataclass(frozen=True, slots=True)
class Phenotypes:
    """Rule table: row p applies to loci ``locus_masks[p]``, needs all clusters in ``extra_masks[p]`` active and (if
    it names any) at least one expected cluster of ``inactive_masks[p]`` inactive."""

    ids: np.ndarray  # S32
    locus_masks: np.ndarray  # bool [n_rules, n_loci]
    extra_masks: np.ndarray  # int8 [n_rules, n_clusters]
    inactive_masks: np.ndarray  # int8 [n_rules, n_clusters]
    extra_counts: np.ndarray  # int8
    priorities: np.ndarray  # int8
    as_suffix: np.ndarray  # bool

    def __len__(self) -> int:
        return len(self.ids)

    def __getitem__(self, item: Any) -> "Phenotypes":
        if isinstance(item, (int, np.integer)):
            raise NotImplementedError("Single item access not implemented for Phenotypes")
        return Phenotypes(*(getattr(self, c)[item] for c in _PHENO_COLS))

    @classmethod
    def empty(cls) -> "Phenotypes":
        return cls(
            np.empty(0, "S32"), np.empty((0, 0), bool), np.empty((0, 0), np.int8), np.empty((0, 0), np.int8),
            np.empty(0, np.int8), np.empty(0, np.int8), np.empty(0, bool),
        )  # fmt: skip

    @classmethod
    def concat(cls, batches: Iterable["Phenotypes"]) -> "Phenotypes":
        bs = list(batches)
        if not bs:
            return cls.empty()
        return cls(*(np.concatenate([getattr(b, c) for b in bs]) for c in _PHENO_COLS))

    @classmethod
    def from_rules(cls, rules: list[Phenotype], locus_ids: tuple[str, ...], cluster_keys: tuple[str, ...]):
        """Rule objects -> mask table (reference: src/kaptive/db/core.py:456-505)."""
        loc = {n: i for i, n in enumerate(locus_ids)}
        clu = {n: i for i, n in enumerate(cluster_keys)}
        lm = np.zeros((len(rules), len(locus_ids)), bool)
        em = np.zeros((len(rules), len(cluster_keys)), np.int8)
        im = np.zeros((len(rules), len(cluster_keys)), np.int8)
        for r, p in enumerate(rules):
            lm[r, [loc[x] for x in p.loci]] = True
            em[r, [clu[x] for x in p.extra_genes]] = 1
            im[r, [clu[x] for x in p.inactive_genes]] = 1
        return cls(
            np.array([p.id.encode() for p in rules], dtype="S32"), lm, em, im, em.sum(axis=1, dtype=np.int8),
            np.array([p.priority for p in rules], dtype=np.int8), np.array([p.as_suffix for p in rules], dtype=bool),
        )  # fmt: skip

    def to_dict(self) -> dict:
        d = {c: getattr(self, c) for c in _PHENO_COLS}
        d["ids"] = np.char.decode(self.ids, "utf-8").tolist() if len(self) else []
        return d
