"""Database: the flat SoA the typing path reads (reference: src/kaptive/db/core.py:32-152).

Field names and dtypes are the reference's (src/kaptive/db/core.py:82-98). What differs is how one is made: the
reference compiles GenBank + TOML with the ``gb-io`` Rust wheel and caches a pickle; here a database is assembled
from already-parsed parts (``from_parts``) and stored as a versioned ``.npz`` blob (``save``/``load``) that holds only
arrays and JSON -- no pickled code objects. GenBank compilation is SURVEY.md section 8 row (f2), not built yet.
"""

from __future__ import annotations

import json
from dataclasses import dataclass
from fnmatch import filter as fnmatch_filter
from pathlib import Path
from typing import Any, Iterable

import numpy as np

from kaptive_amd.core.interval import Intervals
from kaptive_amd.core.seq import SeqRecord, Sequences
from kaptive_amd.db.models import DatabaseError, DatabaseMetadata, Phenotype, Phenotypes

BLOB_VERSION = 1


@dataclass(frozen=True, slots=True)
class Database:
    metadata: DatabaseMetadata
    loci: Sequences
    serotypes: tuple[str, ...]
    locus_gene_offsets: np.ndarray  # uint32
    locus_gene_lengths: np.ndarray  # uint32
    gene_intervals: Intervals
    genes: Sequences
    translations: Sequences
    extra_genes: np.ndarray  # bool
    gene_locus_indices: np.ndarray  # uint16
    cluster_keys: tuple[str, ...]
    gene_cluster_ids: np.ndarray  # uint16
    description_keys: tuple[str, ...]
    gene_description_ids: np.ndarray  # uint16
    gene_positions: np.ndarray  # uint16
    phenotypes: Phenotypes
    loci_sketches: Any = None  # never read on the typing path

    @property
    def max_locus_length(self) -> int:
        return int(np.max(self.loci.lengths)) if len(self.loci) > 0 else 0

    @property
    def cluster_vocab(self) -> dict[str, int]:
        return {k: i for i, k in enumerate(self.cluster_keys)}

    @property
    def description_vocab(self) -> dict[str, int]:
        return {k: i for i, k in enumerate(self.description_keys)}

    # -- building -----------------------------------------------------------------------------------------------
    @staticmethod
    def _parse_phenotype(id_: str, data: dict, locus_iterable: Iterable[str], cluster_iterable: Iterable[str]):
        """One ``[phenotype_logic.<id>]`` TOML table -> Phenotype; tokens with ``*`` are fnmatch patterns, unknown
        names are dropped silently (reference: src/kaptive/db/core.py:183-219)."""
        picked: dict[str, list[str]] = {}
        for token, universe in (("loci", locus_iterable), ("extra_genes", cluster_iterable),
                                ("inactive_genes", cluster_iterable)):  # fmt: skip
            names: list[str] = []
            for t in data.get(token, []):
                if "*" in t:
                    names += fnmatch_filter(universe, t)
                elif t in universe:
                    names.append(t)
            picked[token] = names
        return Phenotype(id_, set(picked["loci"]), set(picked["extra_genes"]), set(picked["inactive_genes"]),
                         data.get("priority", 50))  # fmt: skip

    @classmethod
    def from_parts(cls, metadata: DatabaseMetadata | dict, loci: list[dict]) -> "Database":
        """Assemble from parsed locus records.

        Each locus dict: ``name``, ``type`` (serotype, may be empty), ``extra`` (bool: an "Extra genes" record),
        ``seq`` (bytes) and ``genes`` -- a list of dicts ``start, end, strand, gene, product`` in feature order.
        Naming, vocabularies and derived arrays follow src/kaptive/db/core.py:386-505.
        """
        if isinstance(metadata, dict):
            metadata = DatabaseMetadata.from_dict(metadata)
        records, serotypes, offs, lens, ivs = [], [], [], [], []
        gene_ids, extra_flags, clu_ids, desc_ids, positions = [], [], [], [], []
        clu_vocab: dict[str, int] = {}
        desc_vocab: dict[str, int] = {}
        for loc in loci:
            feats = loc["genes"]
            if not feats:
                continue
            extra = bool(loc.get("extra", False))
            offs.append(len(gene_ids))
            lens.append(len(feats))
            for k, f in enumerate(feats, start=1):
                cluster, product = f.get("gene", ""), f.get("product", "")
                gene_ids.append(cluster if extra else f"{loc['name']}_{k:02}_{cluster}")
                clu_ids.append(clu_vocab.setdefault(cluster, len(clu_vocab)))
                desc_ids.append(desc_vocab.setdefault(product, len(desc_vocab)))
                positions.append(0 if extra else k)
                extra_flags.append(extra)
            lo_hi = [sorted((f["start"], f["end"])) for f in feats]
            ivs.append(
                Intervals(
                    np.array([a for a, _ in lo_hi], np.int32),
                    np.array([b for _, b in lo_hi], np.int32),
                    np.array([-1 if f["strand"] in (-1, "-") else 1 for f in feats], np.int8),
                )
            )
            records.append(SeqRecord(loc["name"], bytes(loc["seq"]).upper()))
            serotypes.append(loc.get("type") or "")
        if not records:
            raise DatabaseError("Database holds no locus with CDS features")
        gene_locus = np.repeat(np.arange(len(lens), dtype=np.uint16), lens)
        loci_seqs = Sequences.from_records(records)
        cluster_keys = tuple(clu_vocab)
        rules = [
            cls._parse_phenotype(k, v, loci_seqs.ids, cluster_keys) for k, v in metadata.phenotype_logic.items()
        ]
        all_ivs = Intervals.concat(ivs)
        genes = loci_seqs.extract_intervals(gene_locus, all_ivs, new_ids=tuple(gene_ids))
        return cls(
            metadata=metadata,
            loci=loci_seqs,
            serotypes=tuple(serotypes),
            locus_gene_offsets=np.array(offs, dtype=np.uint32),
            locus_gene_lengths=np.array(lens, dtype=np.uint32),
            gene_intervals=all_ivs,
            genes=genes,
            translations=genes.translate(),
            extra_genes=np.array(extra_flags, dtype=bool),
            gene_locus_indices=gene_locus,
            cluster_keys=cluster_keys,
            gene_cluster_ids=np.array(clu_ids, dtype=np.uint16),
            description_keys=tuple(desc_vocab),
            gene_description_ids=np.array(desc_ids, dtype=np.uint16),
            gene_positions=np.array(positions, dtype=np.uint16),
            phenotypes=Phenotypes.from_rules(rules, loci_seqs.ids, cluster_keys),
        )

    # -- storage ------------------------------------------------------------------------------------------------
    def save(self, file: str | Path) -> Path:
        file = Path(file)
        text = {
            "blob_version": BLOB_VERSION,
            "metadata": self.metadata.to_dict(),
            "serotypes": self.serotypes,
            "cluster_keys": self.cluster_keys,
            "description_keys": self.description_keys,
            "locus_ids": self.loci.ids,
            "gene_ids": self.genes.ids,
        }
        arrays = {
            "loci_seqs": self.loci.seqs, "loci_offsets": self.loci.offsets, "loci_lengths": self.loci.lengths,
            "gene_seqs": self.genes.seqs, "gene_offsets": self.genes.offsets, "gene_lengths": self.genes.lengths,
            "prot_seqs": self.translations.seqs, "prot_offsets": self.translations.offsets,
            "prot_lengths": self.translations.lengths,
            "iv_starts": self.gene_intervals.starts, "iv_ends": self.gene_intervals.ends,
            "iv_strands": self.gene_intervals.strands,
            "locus_gene_offsets": self.locus_gene_offsets, "locus_gene_lengths": self.locus_gene_lengths,
            "extra_genes": self.extra_genes, "gene_locus_indices": self.gene_locus_indices,
            "gene_cluster_ids": self.gene_cluster_ids, "gene_description_ids": self.gene_description_ids,
            "gene_positions": self.gene_positions,
        }  # fmt: skip
        for k, v in self.phenotypes.to_dict().items():
            if k != "ids":
                arrays["pheno_" + k] = v
        text["pheno_ids"] = self.phenotypes.to_dict()["ids"]
        with file.open("wb") as fh:
            np.savez_compressed(fh, text=np.frombuffer(json.dumps(text).encode(), np.uint8), **arrays)
        return file

    @classmethod
    def load(cls, file: str | Path) -> "Database":
        file = Path(file)
        if not (file.is_file() and file.stat().st_size > 0):
            raise FileNotFoundError(file)
        if file.suffix == ".gbk":
            raise DatabaseError("GenBank compilation is not built yet (SURVEY.md section 8 row f2); load a .npz blob")
        if file.suffix != ".npz":
            raise DatabaseError(f"File {file} not supported")
        with np.load(file, allow_pickle=False) as z:
            text = json.loads(z["text"].tobytes().decode())
            if text.get("blob_version") != BLOB_VERSION:
                raise DatabaseError(f"Unsupported database blob version: {text.get('blob_version')!r}")
            a = {k: z[k] for k in z.files if k != "text"}
        genes = Sequences(tuple(text["gene_ids"]), a["gene_seqs"], a["gene_offsets"], a["gene_lengths"])
        return cls(
            metadata=DatabaseMetadata.from_dict(text["metadata"]),
            loci=Sequences(tuple(text["locus_ids"]), a["loci_seqs"], a["loci_offsets"], a["loci_lengths"]),
            serotypes=tuple(text["serotypes"]),
            locus_gene_offsets=a["locus_gene_offsets"],
            locus_gene_lengths=a["locus_gene_lengths"],
            gene_intervals=Intervals(a["iv_starts"], a["iv_ends"], a["iv_strands"]),
            genes=genes,
            translations=Sequences(genes.ids, a["prot_seqs"], a["prot_offsets"], a["prot_lengths"]),
            extra_genes=a["extra_genes"],
            gene_locus_indices=a["gene_locus_indices"],
            cluster_keys=tuple(text["cluster_keys"]),
            gene_cluster_ids=a["gene_cluster_ids"],
            description_keys=tuple(text["description_keys"]),
            gene_description_ids=a["gene_description_ids"],
            gene_positions=a["gene_positions"],
            phenotypes=Phenotypes(
                np.array([s.encode() for s in text["pheno_ids"]], dtype="S32"), a["pheno_locus_masks"],
                a["pheno_extra_masks"], a["pheno_inactive_masks"], a["pheno_extra_counts"], a["pheno_priorities"],
                a["pheno_as_suffix"],
            ),  # fmt: skip
        )
