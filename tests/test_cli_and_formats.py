"""FASTA ingest (native and Python), GenBank round trip, JSONL wire format and the CLI parser (no GPU needed)."""

import gzip
import json

import numpy as np
import pytest

from kaptive_amd import _native
from kaptive_amd.cli import build_parser, result_to_json, run_convert
from kaptive_amd.core.genome import GenomeAssembly, parse_fasta_bytes
from kaptive_amd.core.seq import SeqRecord, Sequences
from kaptive_amd.db import Database
from kaptive_amd.db.genbank import database_from_genbank, write_genbank
from kaptive_amd.serotyping.io import KaptiveRow
from kaptive_amd.serotyping.models import SerotypingResult
from kaptive_amd.synth import make_assembly, make_db
from tests.golden_util import load_db


def _messy_fasta(genome) -> bytes:
    lines = [b"; comment before the first record", b""]
    for i, r in enumerate(genome.contigs):
        lines.append(b">" + r.id.encode() + (b" len=%d extra words" % len(r.seq) if i % 2 else b""))
        width = 60 if i % 3 else 71
        lines += [r.seq[j : j + width] for j in range(0, len(r.seq), width)]
        if i % 4 == 0:
            lines.append(b"")
    return (b"\r\n" if len(genome.contigs) % 2 else b"\n").join(lines) + b"\n"


def test_fasta_parsers_and_native_pack_agree(tmp_path):
    db = load_db("k")
    genome = make_assembly(db, seed=19, n_run=40, length=90_000, median_contigs=5, min_contig=200)
    text = _messy_fasta(genome)
    recs = parse_fasta_bytes(text)
    assert [n for n, _ in recs] == list(genome.contigs.ids)
    assert [s for _, s in recs] == [r.seq for r in genome.contigs]
    packed, names = _native.fasta_pack(text)
    ref = genome.packed()
    assert names == genome.contigs.ids and packed.padded_len == ref.padded_len
    for f in ("words", "ctg_start", "ctg_len", "n_runs"):
        assert np.array_equal(getattr(packed, f), getattr(ref, f)), f
    # files: plain and gzip, id = file name without the fasta suffix
    p = tmp_path / "sample_7.fna.gz"
    p.write_bytes(gzip.compress(text))
    loaded = GenomeAssembly.from_file(p)
    assert loaded.id == "sample_7" and loaded.contigs.ids == genome.contigs.ids
    assert np.array_equal(loaded.contigs.seqs, genome.contigs.seqs)
    with pytest.raises(NotImplementedError):
        GenomeAssembly.from_file(tmp_path / "reads.fastq")
    # degenerate inputs
    for blob in (b"", b"no header at all\n", b">only_name\n", b">a\n\n>b\nACGT\n"):
        pa, nm = _native.fasta_pack(blob)
        assert len(nm) == len(parse_fasta_bytes(blob)) and pa.padded_len % 64 == 0


@pytest.mark.parametrize("level", [0, 1, 2])
def test_native_pack_fuzz_against_the_python_packer(level):
    """Random FASTA text -- line widths around the packer's 16-symbol blocks and 64-byte chunks, lower case, U, IUPAC / N
    runs that cross line ends, '>' inside a line, bytes above 0x7F, blanks and tabs inside lines, CR LF, empty records, no
    final newline -- packs to what the numpy path (parse_fasta_bytes + pack_contigs) gives, on every text path of the
    ingest (kp_fasta_simd: table look-ups, AVX2, AVX-512), with and without the sequence text kept."""
    from kaptive_amd.core.seq import SeqRecord, Sequences
    from kaptive_amd.pack import pack_contigs

    h = _native.lib()
    if h.kp_fasta_simd(level) != level:
        h.kp_fasta_simd(99)
        pytest.skip(f"this host's CPU has no level-{level} path")
    try:
        rng = np.random.default_rng(99)
        alphabet = np.frombuffer(b"ACGTacgtUuNnRYKM-*>\x80\xff\x00\x7f@`", np.uint8)
        for trial in range(90):
            lines, eol = [], (b"\r\n" if trial % 3 == 0 else b"\n")
            if trial % 4 == 0:
                lines.append(b"leading junk line")
            n_rec = int(rng.integers(0, 6)) if trial % 10 else 700  # (700 short records: the word block has to grow)
            for r in range(n_rec):
                n = int(rng.choice([0, 1, 15, 16, 17, 31, 33, 63, 64, 65, 127, 200, 1000, 5000]))
                p_odd = float(rng.choice([0.0, 0.02, 0.3, 0.95]))
                seq = np.where(rng.random(n) < p_odd, alphabet[rng.integers(8, len(alphabet), n)], alphabet[rng.integers(0, 8, n)])
                seq = seq.astype(np.uint8).tobytes()
                lines.append(b">rec%d_%d %s" % (trial, r, b"desc here" if r % 2 else b""))
                width = int(rng.choice([1, 7, 16, 17, 32, 60, 61, 63, 64, 65, 80, 127, 128, 4000]))
                for j in range(0, n, width):
                    chunk = seq[j : j + width]
                    if chunk[:1] == b">":  # (at a line start that would be a header: the reference parser's reading too)
                        chunk = b"N" + chunk[1:]
                    if rng.random() < 0.1 and len(chunk) > 2:
                        k = int(rng.integers(1, len(chunk)))
                        chunk = chunk[:k] + (b" " if rng.random() < 0.5 else b"\t") + chunk[k:]
                    lines.append(chunk)
                if rng.random() < 0.3:
                    lines.append(b"")
            text = eol.join(lines) + (eol if trial % 5 else b"")
            recs = parse_fasta_bytes(text)
            want = pack_contigs(Sequences.from_records([SeqRecord(nm, sq) for nm, sq in recs]))
            got, names = _native.fasta_pack(text)
            seqs = Sequences.from_records([SeqRecord(nm, sq) for nm, sq in recs])
            direct = _native.pack_contigs(seqs.seqs, seqs.offsets, seqs.lengths)  # the same contigs, already in memory
            assert direct.padded_len == want.padded_len and all(
                np.array_equal(getattr(direct, f), getattr(want, f)) for f in ("words", "ctg_start", "ctg_len", "n_runs")), trial
            assert list(names) == [nm for nm, _ in recs], trial
            assert got.padded_len == want.padded_len, trial
            for f in ("words", "ctg_start", "ctg_len", "n_runs"):
                assert np.array_equal(getattr(got, f), getattr(want, f)), (trial, f)
            pa, names2, kept, lengths = _native.fasta_ingest(text, keep_text=True)
            assert list(names2) == list(names) and np.array_equal(pa.words, want.words) and np.array_equal(pa.n_runs, want.n_runs)
            assert kept.tobytes() == b"".join(sq for _, sq in recs) and lengths.tolist() == [len(sq) for _, sq in recs], trial
    finally:
        h.kp_fasta_simd(99)


def test_shard_ingest_equals_file_by_file(tmp_path):
    """kp_fasta_ingest_shard + kp_shard_words_into (a chunk of files -> the tables and words of one batch, on the
    library's threads) against kp_fasta_ingest of every file: plain, gzip, an empty file, one with N runs, many contigs;
    a missing file is reported, not fatal; the words can be handed over only once."""
    rng = np.random.default_rng(5)
    paths, comps = [], []
    for k in range(9):
        lines = []
        for c in range(int(rng.integers(0, 40)) if k != 3 else 0):
            n = int(rng.integers(0, 3000))
            seq = np.frombuffer(b"ACGTN", np.uint8)[rng.choice(5, n, p=[0.245, 0.245, 0.245, 0.245, 0.02])].tobytes()
            lines.append(b">c%d_%d some words" % (k, c))
            lines += [seq[j : j + 70] for j in range(0, n, 70)]
        text = b"\n".join(lines) + b"\n" if lines else b""
        if k % 3 == 1:
            path, comp = tmp_path / f"asm{k}.fna.gz", "gz"
            path.write_bytes(gzip.compress(text))
        else:
            path, comp = tmp_path / f"asm{k}.fasta", None
            path.write_bytes(text)
        paths.append(str(path))
        comps.append(comp)
    shard = _native.FastaShard(paths, comps, threads=3)
    assert shard.n_asm == 9 and not shard.failed
    word_off, ctg_start, ctg_len, first_ctg, n_runs, first_run = [a.copy() for a in shard.tables()]
    dst = np.full(shard.total_words + 5, 0xDEADBEEF, np.uint32)
    shard.words_into(dst, threads=2)
    with pytest.raises(RuntimeError):
        shard.words_into(dst)
    assert (dst[shard.total_words :] == 0xDEADBEEF).all()
    for i, (path, comp) in enumerate(zip(paths, comps)):
        pa, _, _, _ = _native.fasta_ingest_file(path, gzipped=comp, keep_text=False)
        assert word_off[i + 1] - word_off[i] == pa.padded_len // 16
        assert np.array_equal(dst[word_off[i] : word_off[i + 1]], pa.words), i
        assert np.array_equal(ctg_start[first_ctg[i] : first_ctg[i + 1]], pa.ctg_start)
        assert np.array_equal(ctg_len[first_ctg[i] : first_ctg[i + 1]], pa.ctg_len)
        assert np.array_equal(n_runs[2 * first_run[i] : 2 * first_run[i + 1]].reshape(-1, 2), pa.n_runs)
    shard.close()
    bad = _native.FastaShard([paths[0], str(tmp_path / "missing.fasta"), str(tmp_path / "asm1.fna.gz")], [None, None, None])
    assert [i for i, _ in bad.failed] == [1] and bad.n_asm == 3  # (a gzip file read as text parses -- as junk -- like any bytes)
    bad.close()
    empty = _native.FastaShard([], [])
    assert empty.total_words == 0 and len(empty.tables()[0]) == 1
    empty.words_into(np.empty(0, np.uint32))


@pytest.mark.parametrize("kind", ["kpsc_k", "kpsc_o"])
def test_genbank_round_trip(kind, tmp_path):
    db = make_db(kind, seed=5, n_loci=6 if kind == "kpsc_k" else None)
    path = write_genbank(db, tmp_path / f"{kind}.gbk", antigen_word="K" if kind == "kpsc_k" else "O")
    back = database_from_genbank(path)
    assert back.loci.ids == db.loci.ids and back.genes.ids == db.genes.ids and back.serotypes == db.serotypes
    for f in ("locus_gene_offsets", "locus_gene_lengths", "extra_genes", "gene_locus_indices", "gene_cluster_ids",
              "gene_description_ids", "gene_positions"):  # fmt: skip
        assert np.array_equal(getattr(back, f), getattr(db, f)), f
    assert np.array_equal(back.genes.seqs, db.genes.seqs) and np.array_equal(back.translations.seqs, db.translations.seqs)
    assert back.cluster_keys == db.cluster_keys and back.metadata.id_threshold == db.metadata.id_threshold
    assert np.array_equal(back.phenotypes.locus_masks, db.phenotypes.locus_masks)
    assert np.array_equal(back.phenotypes.extra_masks, db.phenotypes.extra_masks)
    # blob round trip of the compiled database
    again = Database.load(back.save(tmp_path / "db.npz"))
    assert again.genes.ids == db.genes.ids and np.array_equal(again.loci.seqs, db.loci.seqs)


def test_jsonl_round_trip_and_convert(tmp_path, oracle):
    from tests.test_sharding_gloo import _rows_for
    from kaptive_amd.core.pairwise import PairwiseAlignments
    from kaptive_amd.serotyping.core import Serotyper
    from tests.golden_util import hits_to_alignments, load_case

    key, genome, hits, exp, scalars, _ = load_case("o_extra2")
    db = load_db(key)
    typer = Serotyper(
        db, aligner=lambda g: hits_to_alignments(db, g, hits),
        protein_aligner=lambda q, t: PairwiseAlignments.from_table(
            oracle.protein_align(q.seqs, q.offsets, q.lengths, t.seqs, t.offsets, t.lengths)),
    )  # fmt: skip
    res = typer(genome)
    line = result_to_json(res)
    back = SerotypingResult.from_dict(json.loads(line))
    assert bytes(KaptiveRow.from_result(back)) == bytes(KaptiveRow.from_result(res))
    assert back.problems == res.problems and back.translations.ids == res.translations.ids
    # NaN length discrepancy survives (fragmented locus)
    key2, genome2, hits2, *_ = load_case("k_split")
    db2 = load_db(key2)
    res2 = Serotyper(db2, aligner=lambda g: hits_to_alignments(db2, g, hits2), protein_aligner=typer._protein_aligner)(genome2)
    jl = tmp_path / "r.jsonl"
    jl.write_bytes(line + result_to_json(res2))
    args = build_parser().parse_args(["convert", str(jl), "-t", str(tmp_path / "o.tsv")])
    assert run_convert(args) == 0
    rows = (tmp_path / "o.tsv").read_bytes().splitlines(keepends=True)
    assert rows[0] == KaptiveRow.header() and rows[1] == bytes(KaptiveRow.from_result(res))
    assert rows[2] == bytes(KaptiveRow.from_result(res2)) and b"\tn/a\t" in rows[2]


def test_json_lines_follow_orjson_conventions():
    """serotyping/jsonl.py restates what orjson.dumps(..., OPT_SERIALIZE_NUMPY | OPT_APPEND_NEWLINE) writes (the wheel is
    absent here): compact, insertion order, raw UTF-8, numpy values in their own precision, Ryu's number layout, NaN as
    null.  The digits must read back as the same float (float32: as the same float32)."""
    from kaptive_amd.serotyping.jsonl import dumps_line, format_f32, format_f64

    assert [format_f64(x) for x in (1.0, 100.0, 99.5, 0.1, 1e-5, 1e-6, 1.5e-7, 1e15, 1e16, 1.234e20, -2.5, 5e-324, 0.0, -0.0)] == [
        "1.0", "100.0", "99.5", "0.1", "0.00001", "1e-6", "1.5e-7", "1000000000000000.0", "1e16", "1.234e20", "-2.5", "5e-324",
        "0.0", "-0.0"]  # fmt: skip
    assert [format_f32(x) for x in (0.1, 99.5, 1e-6, 1e-7, 16777216.0, 1e13, 3.4028235e38)] == [
        "0.1", "99.5", "0.000001", "1e-7", "16777216.0", "1e13", "3.4028235e38"]
    assert format_f64(float("nan")) == format_f64(float("inf")) == format_f32(np.float32("nan")) == "null"
    rng = np.random.default_rng(4)
    for x in np.concatenate([rng.random(300) * 100, 10.0 ** rng.uniform(-30, 30, 300), rng.standard_normal(100) * 1e-3]):
        assert float(format_f64(float(x))) == float(x)
        assert np.float32(format_f32(np.float32(x))) == np.float32(x)
        assert len(format_f32(np.float32(x))) <= len(format_f64(float(np.float32(x))))  # never the widened digits
    line = dumps_line({"a": np.float32(0.1), "b": np.array([1, -2], np.int8), "c": np.array([True, False]), "d": 'h\u00e9 "q"\n',
                       "e": (1, 2.0), "f": np.array([[0.1, 2.5]], np.float32), "g": float("nan"), "h": None, "i": np.uint32(7)})
    assert line == '{"a":0.1,"b":[1,-2],"c":[true,false],"d":"h\u00e9 \\"q\\"\\n","e":[1,2.0],"f":[[0.1,2.5]],"g":null,"h":null,"i":7}\n'.encode()
    assert json.loads(line)["d"] == 'h\u00e9 "q"\n'


def test_cli_parser_matches_reference_flags():
    ap = build_parser()
    a = ap.parse_args(["assembly", "db.npz", "a.fasta", "b.fna.gz", "-o", "out.tsv", "-j", "out.jsonl", "--max-other-genes",
                       "2", "--min-completeness", "0.7", "--below-threshold", "-t", "4", "--partial-edge-tolerance", "9",
                       "-l", "loci", "-g", "genes", "-p", "prot", "--pha4ge", "p.tsv", "-V"])  # fmt: skip
    assert a.genomes == ["a.fasta", "b.fna.gz"] and a.max_other_genes == 2 and a.min_completeness == 0.7
    assert a.below_threshold and a.threads == 4 and a.partial_edge_tolerance == 9 and a.out == "out.tsv"
    b = ap.parse_args(["type", "db.npz", "x.fa"])
    assert b.func is a.func and b.max_other_genes == 1 and b.min_completeness == 0.5 and not b.below_threshold
    # defaults, nargs and const of the output flags as the reference declares them (src/kaptive/cli.py:424-504):
    # -o defaults to stdout; the others are off unless given, and take their const when given without a value
    assert b.out == "stdout" and b.json is None and b.pha4ge is None and b.loci is None and b.genes is None and b.proteins is None
    from pathlib import Path

    c = ap.parse_args(["type", "db.npz", "x.fa", "-l", "-g", "-p", "-j", "--pha4ge"])
    assert c.loci == Path("./") and c.genes == Path("./") and c.proteins == Path("./")
    assert c.json == "kaptive_results.jsonl" and c.pha4ge == Path("kaptive_results.pha4ge") and c.out == "stdout"
    d = ap.parse_args(["convert", "r.jsonl"])
    assert d.tsv is None and d.loci is None
    e = ap.parse_args(["convert", "r.jsonl", "-t"])
    assert e.tsv == "stdout"
    f = ap.parse_args(["convert", "r.jsonl", "-t", "x.tsv", "-g", "genes"])
    assert f.tsv == "x.tsv" and f.genes == Path("genes")


def test_from_file_goes_through_the_native_ingest(tmp_path):
    """GenomeAssembly.from_file = one native pass (kp_fasta_ingest: zlib for .gz, record split, whitespace strip, 2-bit
    pack).  Contigs and packed form equal what the numpy parser + kp_pack_contigs give, for every suffix the reference
    opens (src/kaptive/core/genome.py:194-214), odd formatting, several gzip members; a truncated stream raises."""
    import bz2
    import gzip
    import lzma

    db = make_db("kpsc_k", seed=7, n_loci=3)
    genome = make_assembly(db, seed=3, length=300_000, median_contigs=9, n_run=40)
    fa = genome.contigs.to_fasta()
    messy = (b"junk before\r\n" + fa[:20000].replace(b"\n", b"\r\n") + b">empty\n>sp ace desc\nAC GT\tNNRY acgu\n" + fa[20000:])
    for stem, data in (("a.fasta", fa), ("b.fna", messy)):
        recs = parse_fasta_bytes(data)
        want = Sequences.from_records([SeqRecord(n, s) for n, s in recs])
        want_packed = _native.pack_contigs(want.seqs, want.offsets, want.lengths)
        for ext, squeeze in (("", None), (".gz", gzip.compress), (".bz2", bz2.compress), (".xz", lzma.compress)):
            path = tmp_path / (stem + ext)
            path.write_bytes(squeeze(data) if squeeze else data)
            got = GenomeAssembly.from_file(path)
            assert got.id == stem.rsplit(".", 1)[0] and got.contigs.ids == want.ids, path
            assert np.array_equal(got.contigs.seqs, want.seqs) and np.array_equal(got.contigs.offsets, want.offsets)
            assert np.array_equal(got.contigs.lengths, want.lengths)
            pg = got.packed()
            assert np.array_equal(pg.words, want_packed.words) and np.array_equal(pg.n_runs, want_packed.n_runs)
            assert np.array_equal(pg.ctg_start, want_packed.ctg_start) and pg.padded_len == want_packed.padded_len
    two = tmp_path / "two.fa.gz"
    two.write_bytes(gzip.compress(fa[:10000]) + gzip.compress(fa[10000:]))
    assert np.array_equal(GenomeAssembly.from_file(two).contigs.seqs, genome.contigs.seqs)
    cut = tmp_path / "cut.fa.gz"
    cut.write_bytes(gzip.compress(fa)[:3000])
    with pytest.raises(ValueError):
        GenomeAssembly.from_file(cut)
    with pytest.raises(NotImplementedError):
        GenomeAssembly.from_file(tmp_path / "x.txt")
    # bz2 and xz are inflated inside the library too (libbz2 / liblzma of the host, looked up at first use): several
    # streams in one file are read through, a truncated stream raises
    assert _native.compression_is_native(), "libbz2.so.1 / liblzma.so.5 are part of the image"
    for ext, squeeze in ((".bz2", bz2.compress), (".xz", lzma.compress)):
        pa, names, seqs, _ = _native.fasta_ingest(squeeze(fa[:10000]) + squeeze(fa[10000:]), gzipped=ext[1:])
        assert np.array_equal(seqs, genome.contigs.seqs) and tuple(names) == genome.contigs.ids
        with pytest.raises(ValueError):
            _native.fasta_ingest(squeeze(fa)[:3000], gzipped=ext[1:])
        with pytest.raises(ValueError):
            _native.fasta_ingest(b"not a compressed stream at all", gzipped=ext[1:])


def test_ingest_views_outlive_the_call_and_buffers_are_recycled():
    """fasta_ingest hands out views of the native buffers (no copy with the GIL held); the record behind them lives as
    long as any view does, its blocks go back to the library's pool afterwards, and a pool of reader threads gives the
    same bytes as one."""
    import gc
    from concurrent.futures import ThreadPoolExecutor

    db = make_db("kpsc_k", seed=7, n_loci=3)
    texts = [make_assembly(db, seed=40 + i, length=120_000 + 7_000 * i, median_contigs=5, n_run=10).contigs.to_fasta()
             for i in range(6)]
    want = []
    for t in texts:
        pa, names, seqs, lengths = _native.fasta_ingest(t)
        want.append((pa.words.copy(), seqs.copy(), names))
        words_view, seqs_view = pa.words, seqs
        del pa, seqs
        gc.collect()  # the views alone keep the native record alive
        assert np.array_equal(words_view, want[-1][0]) and np.array_equal(seqs_view, want[-1][1])
        assert words_view.flags.writeable is not None and not words_view.flags.owndata
    del words_view, seqs_view
    gc.collect()
    with ThreadPoolExecutor(4) as pool:  # blocks of freed records are reused by whichever thread asks next
        for rnd in range(3):
            got = list(pool.map(_native.fasta_ingest, texts))
            for (pa, names, seqs, _), (w, s, n) in zip(got, want):
                assert np.array_equal(pa.words, w) and np.array_equal(seqs, s) and names == n
            del got


def test_from_files_equals_from_file(tmp_path):
    """GenomeAssembly.from_files (one native call, the library's own threads) gives what from_file gives file by file,
    for plain, gzip, bz2 and xz files in one list; an unreadable file raises."""
    import bz2
    import gzip
    import lzma

    db = make_db("kpsc_k", seed=7, n_loci=3)
    paths = []
    for i, (ext, squeeze) in enumerate((("", None), (".gz", gzip.compress), (".bz2", bz2.compress), (".xz", lzma.compress),
                                        ("", None), (".gz", gzip.compress))):
        g = make_assembly(db, seed=60 + i, length=90_000 + 5_000 * i, median_contigs=4 + i, n_run=5 * i)
        data = g.contigs.to_fasta()
        p = tmp_path / f"asm{i}.fasta{ext}"
        p.write_bytes(squeeze(data) if squeeze else data)
        paths.append(p)
    many = GenomeAssembly.from_files(paths, threads=3)
    for p, got in zip(paths, many):
        want = GenomeAssembly.from_file(p)
        assert got.id == want.id and got.contigs.ids == want.contigs.ids
        assert np.array_equal(got.contigs.seqs, want.contigs.seqs) and np.array_equal(got.contigs.lengths, want.contigs.lengths)
        assert np.array_equal(got.packed().words, want.packed().words) and np.array_equal(got.packed().n_runs, want.packed().n_runs)
    assert GenomeAssembly.from_files([]) == []
    bad = tmp_path / "cut.fasta.gz"
    bad.write_bytes(gzip.compress(b">x\nACGT\n")[:12])
    with pytest.raises(ValueError):
        GenomeAssembly.from_files([paths[0], bad])


def test_failing_pipeline_start_cancels_its_queued_reads(tmp_path, monkeypatch):
    """A bad database (or no device) after the early reads were queued: the constructor shuts its reader pools down and
    cancels what is queued, so that the error is reported at once instead of after every queued FASTA was read by the
    interpreter's exit hook (advisor finding, round 5)."""
    import threading

    from kaptive_amd import cli as C

    gate, started = threading.Event(), []

    def slow_shard(self, paths):
        started.append(paths)
        gate.wait(5)
        raise RuntimeError("reader let go")

    monkeypatch.setattr(C._TypingPipeline, "_load_shard", slow_shard)
    monkeypatch.setattr(C._TypingPipeline, "PREFETCH", 0)  # one shard reader: every chunk but the first stays queued
    files = []
    for i in range(6):
        f = tmp_path / f"g{i}.fasta"
        f.write_text(">c\nACGT\n")
        files.append(str(f))
    args = C.build_parser().parse_args(["assembly", str(tmp_path / "missing_db.npz"), *files, "-o", str(tmp_path / "o.tsv")])
    monkeypatch.setenv("KAPTIVE_AMD_READ_AHEAD_GB", "1")
    pipe = C._TypingPipeline.__new__(C._TypingPipeline)
    chunks = [(k, [p]) for k, p in enumerate(files)]
    with pytest.raises((FileNotFoundError, OSError)):
        pipe.__init__(args, 0, chunks=chunks)
    gate.set()
    for pool in (pipe.readers, pipe.shard_readers, pipe.copiers, pipe.formatters, pipe.janitor):
        assert pool._shutdown
        pool.shutdown(wait=True)  # returns at once: nothing is left to drain
    assert len(started) <= 1 and pipe._early == []  # the five queued chunks were cancelled, not read
