#!/usr/bin/env python3
"""Write every golden typing case as files the real ``kaptive`` accepts, so that a machine with network access (and
therefore Kaptive with its rammappy wheel) can produce TRUE end-to-end goldens for this repository.

    python tools/export_golden_inputs.py OUT_DIR

Writes ``OUT_DIR/db_<key>.gbk`` + ``.toml`` (one synthetic database per key of tests/golden/db_*.npz) and
``OUT_DIR/<case>.fasta`` for every case of tests/golden/typing_index.json, plus ``OUT_DIR/RUN.sh``, which runs

    kaptive assembly OUT_DIR/db_<key>.gbk OUT_DIR/<case>.fasta -o OUT_DIR/real_<case>.tsv

for each case.  Copy the resulting ``real_<case>.tsv`` files into tests/golden/: tests/test_real_kaptive_goldens.py then
compares the product's TSV row for each of them byte for byte (today the aligner stage has no such golden -- the
reference's aligner is a closed wheel that is absent from the build image).
"""
from __future__ import annotations

import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))


def main(out_dir: str) -> None:
    from kaptive_amd.db.genbank import write_genbank
    from tests.golden_util import case_names, load_case, load_db

    out = Path(out_dir)
    out.mkdir(parents=True, exist_ok=True)
    written, lines = set(), ["#!/bin/sh", "# run on a machine where `kaptive` (>= 3.3) is installed", "set -e", 'cd "$(dirname "$0")"']
    for name in case_names():
        key, genome, _hits, _exp, _scalars, kwargs = load_case(name)
        if name.startswith("random_hits"):
            continue  # hit tables without an assembly behind them: reduction-only cases
        if key not in written:
            db = load_db(key)
            write_genbank(db, out / f"db_{key}.gbk", antigen_word=db.metadata.antigen or "K")
            written.add(key)
        (out / f"{name}.fasta").write_bytes(genome.contigs.to_fasta())
        flags = []
        if "max_other_genes" in kwargs:
            flags += ["--max-other-genes", str(kwargs["max_other_genes"])]
        if "min_completeness" in kwargs:
            flags += ["--min-completeness", str(kwargs["min_completeness"])]
        if kwargs.get("allow_below_threshold"):
            flags.append("--below-threshold")
        lines.append(f"kaptive assembly db_{key}.gbk {name}.fasta {' '.join(flags)} -o real_{name}.tsv".replace("  ", " "))
    (out / "RUN.sh").write_text("\n".join(lines) + "\n")
    print(f"wrote {len(lines) - 4} cases and {len(written)} databases to {out}")


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "golden_inputs")
