"""TSV report rows (reference: src/kaptive/serotyping/io.py:19-382). Bytes in, bytes out; the header and every
column are byte-for-byte what the reference writes for the same SerotypingResult."""

from __future__ import annotations

from dataclasses import dataclass, fields
from typing import Iterable, Iterator

import numpy as np

from kaptive_amd.serotyping.models import GeneState, SerotypingProblem, SerotypingResult


@dataclass(slots=True, frozen=True)
class ReportRow:
    @classmethod
    def header(cls) -> bytes:
        return ("\t".join(f.name for f in fields(cls)) + "\n").encode("utf-8")

    def __bytes__(self) -> bytes:
        return b"\t".join(getattr(self, f.name) for f in fields(self)) + b"\n"

    @classmethod
    def _parse_header_line(cls, header_line: bytes) -> list[str]:
        return header_line.rstrip(b"\r\n").decode("utf-8").split("\t")

    @classmethod
    def read_tsv(cls, lines: Iterable[bytes]) -> Iterator["ReportRow"]:
        it = iter(lines)
        first = next(it, None)
        if first is None:
            return
        names = cls._parse_header_line(first)
        known = {f.name for f in fields(cls)}
        for line in it:
            line = line.rstrip(b"\r\n")
            if line:
                yield cls(**{n: v for n, v in zip(names, line.split(b"\t")) if n in known})


_STATE_TAG = {GeneState.PARTIAL.value: b"partial", GeneState.TRUNCATED.value: b"truncated",
              GeneState.NOVEL.value: b"below_id_threshold"}  # fmt: skip


def _gene_details(result: SerotypingResult, mask: np.ndarray) -> bytes:
    """``id,ident%,cov%[,state]`` per selected hit, ``;``-joined."""
    out = []
    for i in np.flatnonzero(mask):
        parts = [result.gene_seqs.ids[i].encode("utf-8"), b"%.2f%%" % result.protein_identities[i],
                 b"%.2f%%" % result.gene_hits.coverages[i]]  # fmt: skip
        tag = _STATE_TAG.get(int(result.gene_states[i]))
        if tag:
            parts.append(tag)
        out.append(b",".join(parts))
    return b";".join(out)


@dataclass(slots=True, frozen=True)
class KaptiveRow(ReportRow):
    Kaptive_version: bytes
    Database_name: bytes
    Database_version: bytes
    Assembly: bytes
    Best_match_locus: bytes
    Best_match_type: bytes
    Match_confidence: bytes
    Problems: bytes
    Identity: bytes
    Coverage: bytes
    Length_discrepancy: bytes
    Expected_genes_in_locus: bytes
    Expected_genes_in_locus_details: bytes
    Missing_expected_genes: bytes
    Other_genes_in_locus: bytes
    Other_genes_in_locus_details: bytes
    Expected_genes_outside_locus: bytes
    Expected_genes_outside_locus_details: bytes
    Other_genes_outside_locus: bytes
    Other_genes_outside_locus_details: bytes
    Truncated_genes_details: bytes
    Extra_genes_details: bytes

    @classmethod
    def header(cls) -> bytes:
        cols = [f.name.encode().replace(b"_details", b", details").replace(b"_", b" ") for f in fields(cls)]
        return b"\t".join(cols) + b"\n"

    @classmethod
    def _parse_header_line(cls, header_line: bytes) -> list[str]:
        cols = header_line.rstrip(b"\r\n").split(b"\t")
        return [c.replace(b", details", b"_details").replace(b" ", b"_").decode("utf-8") for c in cols]

    @classmethod
    def from_result(cls, result: SerotypingResult) -> "KaptiveRow":
        h, st = result.gene_hits, result.gene_states
        inside, exp = h.is_inside, h.is_expected
        other = ~exp & ~h.is_extra

        def n_genes(mask: np.ndarray) -> int:
            return len(np.unique(h.gene_indices[mask]))

        n_in, n_out = n_genes(inside & exp), n_genes(~inside & exp)
        total = n_in + n_out + len(result.missing_expected_genes)

        def share(n: int) -> bytes:
            return b"%d / %d (%.2f%%)" % (n, total, n / total * 100.0) if total else b"0 / 0 (0.00%)"

        ld = result.length_discrepancy
        return cls(
            Kaptive_version=result.kaptive_version.encode(),
            Database_name=result.database_name.encode(),
            Database_version=result.database_version.encode(),
            Assembly=result.genome.encode(),
            Best_match_locus=result.best_locus_name.encode(),
            Best_match_type=result.phenotype.encode(),
            Match_confidence=b"Typeable" if result.typeable else b"Untypeable",
            Problems=result.problems.to_symbols(),
            Identity=b"%.2f%%" % result.percent_identity,
            Coverage=b"%.2f%%" % result.percent_coverage,
            Length_discrepancy=b"n/a" if (ld is None or np.isnan(ld)) else b"%d" % int(ld),
            Expected_genes_in_locus=share(n_in),
            Expected_genes_in_locus_details=_gene_details(result, inside & exp),
            Missing_expected_genes=b";".join(g.encode("utf-8") for g in result.missing_expected_genes),
            Other_genes_in_locus=b"%d" % n_genes(inside & other),
            Other_genes_in_locus_details=_gene_details(result, inside & other),
            Expected_genes_outside_locus=share(n_out),
            Expected_genes_outside_locus_details=_gene_details(result, ~inside & exp),
            Other_genes_outside_locus=b"%d" % n_genes(~inside & other),
            Other_genes_outside_locus_details=_gene_details(result, ~inside & other),
            Truncated_genes_details=_gene_details(
                result, (st == GeneState.TRUNCATED.value) | (st == GeneState.PARTIAL.value)
            ),
            Extra_genes_details=_gene_details(result, h.is_extra),
        )


_PHA4GE_PROBLEMS = (
    (SerotypingProblem.TRUNCATED_GENES, b"truncated gene/s in locus"),
    (SerotypingProblem.NOVEL_GENES, b"low identity gene/s"),
    (SerotypingProblem.FRAGMENTED, None),
    (SerotypingProblem.MISSING_GENES, b"missing expected gene/s"),
    (SerotypingProblem.UNEXPECTED_GENES, b"unexpected gene/s in locus"),
)


@dataclass(slots=True, frozen=True, kw_only=True)
class Pha4geRow(ReportRow):
    sample: bytes
    genotyping_method: bytes = b"In silico serotyping"
    genotyping_schema_taxon: bytes
    genotyping_database_name: bytes
    genotyping_database_version: bytes
    genotyping_schema_name: bytes = b"Kaptive"
    genotyping_software_name: bytes = b"Kaptive"
    genotyping_software_version: bytes
    genotype: bytes
    genotype_predicted_phenotype: bytes
    genotype_confidence_value: bytes
    genotyping_details: bytes
    genotyping_method_url: bytes = b"https://github.com/klebgenomics/Kaptive"

    @classmethod
    def from_result(cls, result: SerotypingResult) -> "Pha4geRow":
        locus = result.best_locus_name.encode()
        if result.problems:
            notes = [
                (b"match broken into %d pieces" % len(result.locus_pieces)) if text is None else text
                for flag, text in _PHA4GE_PROBLEMS
                if flag in result.problems
            ]
            details = b"Best locus match: %b. Problems: %b" % (locus, b", ".join(notes))
        else:
            details = b"Best locus match: %b." % locus
        return cls(
            sample=result.genome.encode(),
            genotyping_schema_taxon=b"%s [NCBITaxon:%d]" % (result.database_organism.encode(), result.database_taxon),
            genotyping_database_name=result.database_name.encode(),
            genotyping_database_version=result.database_version.encode(),
            genotyping_software_version=result.kaptive_version.encode(),
            genotype=locus,
            genotype_confidence_value=b"Typeable" if result.typeable else b"Untypeable",
            genotype_predicted_phenotype=result.phenotype.encode(),
            genotyping_details=details,
        )
