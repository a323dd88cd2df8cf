"""True end-to-end goldens: TSV rows written by the real ``kaptive assembly`` (with its rammappy aligner) for the golden
cases, when someone has produced them (tools/export_golden_inputs.py explains how; none can be made in the build image,
which has neither rammappy nor network access).  Each ``tests/golden/real_<case>.tsv`` present is compared byte for byte
with the product's row for the same FASTA and database; with none present the test reports that aligner parity is
still unpinned instead of passing silently."""

import pytest

from tests.golden_util import GOLDEN, case_names, load_case, load_db

REAL = sorted(p.stem[len("real_") :] for p in GOLDEN.glob("real_*.tsv"))


def test_export_tool_writes_inputs_kaptive_accepts(tmp_path):
    """The exported GenBank + TOML compile back into the same database and the FASTA files into the same assemblies."""
    import numpy as np

    from kaptive_amd.core.genome import GenomeAssembly
    from kaptive_amd.db.genbank import database_from_genbank
    from tools.export_golden_inputs import main

    main(str(tmp_path))
    run = (tmp_path / "RUN.sh").read_text()
    for name in case_names():
        if name.startswith("random_hits"):
            continue
        key, genome, *_ = load_case(name)
        assert f"db_{key}.gbk {name}.fasta" in run
        back = GenomeAssembly.from_file(tmp_path / f"{name}.fasta")
        assert back.contigs.ids == genome.contigs.ids and np.array_equal(back.contigs.seqs, genome.contigs.seqs)
    for key in ("k", "o"):
        db, back = load_db(key), database_from_genbank(tmp_path / f"db_{key}.gbk")
        assert back.genes.ids == db.genes.ids and np.array_equal(back.genes.seqs, db.genes.seqs)
        assert back.loci.ids == db.loci.ids and back.serotypes == db.serotypes
        assert np.array_equal(back.extra_genes, db.extra_genes) and len(back.phenotypes) == len(db.phenotypes)


@pytest.mark.gpu
@pytest.mark.parametrize("name", REAL or ["<none>"])
def test_row_equals_real_kaptive(name):
    if name == "<none>":
        pytest.skip("no tests/golden/real_*.tsv: aligner parity against real Kaptive is unpinned (see DESIGN.md section 2)")
    from kaptive_amd.serotyping.core import Serotyper
    from kaptive_amd.serotyping.io import KaptiveRow

    key, genome, _hits, _exp, _scalars, kwargs = load_case(name)
    want = (GOLDEN / f"real_{name}.tsv").read_bytes().splitlines(keepends=True)
    assert want[0] == KaptiveRow.header()
    typer = Serotyper(load_db(key), **kwargs)
    got = bytes(KaptiveRow.from_result(typer(genome)))
    typer.engine.close()
    # the version column is whatever Kaptive release wrote the file
    assert got.split(b"\t", 1)[1] == want[1].split(b"\t", 1)[1]
