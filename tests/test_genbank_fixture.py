"""GenBank + TOML compile pinned to the reference.

``tests/golden/handwritten_db.gbk`` / ``.toml`` are written by hand after the reference's curation guide
(docs/db/curation.md:40-76 for the flat file, 85-158 for the metadata and phenotype logic) and exercise the rules of
``Database.from_genbank`` (src/kaptive/db/core.py:322-324 note patterns, 386-393 gene ids and vocabularies, 402-404
coordinates and strands, 415/437 extra-gene records, records without CDS skipped).  ``db_genbank_expected.npz`` holds what
the REFERENCE compiled from those two files (oracle/make_golden.py::gen_genbank).  The flat-file reader itself is
checked against the values typed into the fixture by hand (first test)."""

import numpy as np

from kaptive_amd.db.genbank import database_from_genbank, read_genbank
from tests.golden_util import GOLDEN

GBK = GOLDEN / "handwritten_db.gbk"


def test_reader_sees_what_was_typed_into_the_fixture():
    recs = read_genbank(GBK)
    assert [r["name"] for r in recs] == ["OL2a_1_record", "second", "nocds", "extras"]
    assert [len(r["seq"]) for r in recs] == [1500, 900, 300, 1300]
    first = recs[0]["features"]
    assert [f["kind"] for f in first] == ["source", "gene", "CDS", "CDS", "CDS", "CDS"]
    assert [f["location"] for f in first[2:]] == ["10..309", "complement(400..699)", "join(750..800,850..1049)",
                                                   "complement(<1100..>1399)"]  # fmt: skip
    notes = [q["value"] for q in first[0]["quals"] if q["key"] == "note"]
    assert notes == ["O locus: OL2α.1", "O type: O2α"]
    product = next(q["value"] for q in first[2]["quals"] if q["key"] == "product")
    assert product == ("ABC transporter permease with a description that is long enough to wrap over two lines of the "
                       "flat file")
    assert recs[1]["seq"][:450].isupper() or recs[1]["seq"].isupper()  # sequences come back upper-cased or as written


def test_compile_equals_reference_from_genbank():
    z = np.load(GOLDEN / "db_genbank_expected.npz")
    db = database_from_genbank(GBK)
    assert db.loci.ids == tuple(z["locus_ids"].tolist()) == ("OL2α.1", "OL101", "Extra_genes_O")
    assert db.serotypes == tuple(z["serotypes"].tolist())
    assert db.genes.ids == tuple(z["gene_ids"].tolist())
    assert db.genes.ids[:4] == ("OL2α.1_01_wzm", "OL2α.1_02_wzt", "OL2α.1_03_", "OL2α.1_04_wbbM") and db.genes.ids[-2:] == ("gmlA", "wbbY")
    assert db.cluster_keys == tuple(z["cluster_keys"].tolist()) and db.description_keys == tuple(z["description_keys"].tolist())
    for ours, theirs in (
        (db.loci.seqs, "loci_seqs"), (db.loci.offsets, "loci_offsets"), (db.loci.lengths, "loci_lengths"),
        (db.genes.seqs, "gene_seqs"), (db.genes.offsets, "gene_offsets"), (db.genes.lengths, "gene_lengths"),
        (db.translations.seqs, "prot_seqs"), (db.translations.offsets, "prot_offsets"),
        (db.translations.lengths, "prot_lengths"), (db.gene_intervals.starts, "gene_starts"),
        (db.gene_intervals.ends, "gene_ends"), (db.gene_intervals.strands, "gene_strands"),
        (db.gene_positions, "gene_positions"), (db.extra_genes, "extra_genes"),
        (db.gene_locus_indices, "gene_locus_indices"), (db.locus_gene_offsets, "locus_gene_offsets"),
        (db.locus_gene_lengths, "locus_gene_lengths"), (db.gene_cluster_ids, "gene_cluster_ids"),
        (db.gene_description_ids, "gene_description_ids"),
    ):  # fmt: skip
        assert np.array_equal(np.asarray(ours), z[theirs]), theirs
        assert np.asarray(ours).dtype == z[theirs].dtype, (theirs, np.asarray(ours).dtype, z[theirs].dtype)
    assert db.max_locus_length == int(z["max_locus_length"]) and db.metadata.id_threshold == float(z["id_threshold"])
    ph = db.phenotypes
    assert [bytes(i) for i in ph.ids] == [bytes(i) for i in z["pheno_ids"]]
    for ours, theirs in ((ph.locus_masks, "pheno_locus_masks"), (ph.extra_masks, "pheno_extra_masks"),
                         (ph.inactive_masks, "pheno_inactive_masks"), (ph.extra_counts, "pheno_extra_counts"),
                         (ph.priorities, "pheno_priorities"), (ph.as_suffix, "pheno_as_suffix")):  # fmt: skip
        assert np.array_equal(ours, z[theirs]) and ours.dtype == z[theirs].dtype, theirs
