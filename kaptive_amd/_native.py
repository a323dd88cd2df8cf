"""ctypes binding of libkaptive_amd.so (include/kaptive_amd.h).  There is no fallback: if the library or a GPU is
missing, every entry point raises."""

from __future__ import annotations

import atexit
import ctypes as C
import os
import threading
import weakref
from pathlib import Path

import numpy as np

LIB_PATH = Path(__file__).resolve().parent / "libkaptive_amd.so"
MAX_GENE_LEN = 65535  # KP_MAX_GENE_LEN of include/kp_spec.h (tests/test_native_abi.py compares the two)
FILL16_MAX_GENE_LEN = 15800  # KP_FILL16_MAX_GENE_LEN: longer genes are filled with 32-bit scores
WORK_SLOTS = 3  # KP_WORK_SLOTS of include/kaptive_amd.h: alignment results a context keeps resident

HIT_DTYPE = np.dtype(
    [("gene", "<i4"), ("contig", "<i4"), ("q_start", "<i4"), ("q_end", "<i4"), ("t_start", "<i4"), ("t_end", "<i4"),
     ("score", "<i4"), ("matches", "<i4"), ("block_len", "<i4"), ("strand", "i1"), ("mapq", "u1"), ("n_seeds", "u1"), ("pad", "u1")]
)  # fmt: skip
TASK_DTYPE = np.dtype(
    [("gs", "<i4"), ("contig", "<i4"), ("lo", "<i4"), ("width", "<i4"), ("n_anchors", "<i4"), ("qmin", "<i4"),
     ("qmax", "<i4"), ("chain_score", "<i4")]
)  # fmt: skip

JOIN_MAX_PIECES = 8  # KP_JOIN_MAX_PIECES
JOIN_DTYPE = np.dtype(
    [("gs", "<i4"), ("contig", "<i4"), ("n_pieces", "<i4"), ("n_anchors", "<i4"), ("chain_score", "<i4"), ("width", "<i4"),
     ("lo", "<i4", (JOIN_MAX_PIECES,)), ("piece", "<i4", (JOIN_MAX_PIECES, 11))]
)  # fmt: skip (kp_batch_joins: per piece state, visited mask, cell score, q_start, q_end, t_start, t_end, matches, block_len, score, bonus)

EXPORTS = (
    "kp_ctx_create", "kp_ctx_destroy", "kp_last_error", "kp_ctx_stream", "kp_ctx_set_option", "kp_host_alloc",
    "kp_host_free", "kp_host_reserve", "kp_host_lock", "kp_host_pinned_bytes", "kp_device_allocations", "kp_db_load", "kp_db_n_postings", "kp_batch_create", "kp_batch_create_async",
    "kp_batch_upload_wait", "kp_batch_depends_on", "kp_batch_create_device", "kp_batch_device_words", "kp_batch_destroy", "kp_batch_align", "kp_batch_wait",
    "kp_batch_hit_offsets", "kp_batch_hits", "kp_batch_set_hits", "kp_batch_stats", "kp_batch_profile", "kp_batch_anchors",
    "kp_batch_tasks", "kp_batch_task_results", "kp_batch_joins", "kp_db_load_typing", "kp_db_load_typing_group", "kp_batch_use_group", "kp_batch_score", "kp_batch_reduce", "kp_batch_typing_caps",
    "kp_device_count", "kp_device_numa_node", "kp_batch_typing", "kp_batch_proteins", "kp_protein_align", "kp_fasta_pack", "kp_fasta_ingest", "kp_fasta_ingest_many", "kp_fasta_ingest_file", "kp_fasta_ingest_shard", "kp_shard_words_into", "kp_shard_free", "kp_fasta_simd", "kp_pack_contigs",
    "kp_fasta_free", "kp_format_rows", "kp_format_json", "kp_format_fasta", "kp_protein_align_seeded", "kp_randstrobes", "kp_randstrobe_top_hits",
)  # fmt: skip


class PackedFasta(C.Structure):  # kp_packed_fasta
    _fields_ = [("padded_len", C.c_int64), ("n_contigs", C.c_int32), ("n_runs", C.c_int32),
                ("words", C.POINTER(C.c_uint32)), ("ctg_start", C.POINTER(C.c_int32)), ("ctg_len", C.POINTER(C.c_int32)),
                ("n_run_pairs", C.POINTER(C.c_int32)), ("names", C.POINTER(C.c_char)), ("name_off", C.POINTER(C.c_int32)),
                ("seqs", C.POINTER(C.c_uint8)), ("n_seq_bytes", C.c_int64)]  # fmt: skip


class _FastaRecord:
    """Keeps a kp_packed_fasta alive for as long as an array that views its large buffers (packed words, sequence text)
    is: the arrays are handed out without a copy -- a copy is a few milliseconds with the GIL held per 5 Mbp assembly,
    which is what stopped a thread pool of readers from scaling -- and the buffers return to the library's block pool
    when the last view dies."""

    def __init__(self, out) -> None:
        self.out = out

    def __del__(self) -> None:
        try:
            h = lib()
            h.kp_fasta_free.restype = None
            h.kp_fasta_free(self.out)
        except Exception:  # interpreter shutdown
            pass

    def view(self, ptr, n: int, ctype, dtype) -> np.ndarray:
        if not n:
            return np.empty(0, dtype)
        buf = (ctype * n).from_address(C.addressof(ptr.contents))
        buf._record = self  # numpy keeps `buf` as the array's base, `buf` keeps the record
        return np.frombuffer(buf, dtype=dtype)


def _packed_from(out, want_names: bool, record: "_FastaRecord | None" = None):
    """kp_packed_fasta -> (PackedAssembly, names).  With ``record`` the packed words are a view of the library's buffer
    (the record frees it when the views are gone); without, everything is copied and the caller frees."""
    from kaptive_amd.pack import PackedAssembly

    p = out.contents
    nc, nr = p.n_contigs, p.n_runs
    arr = lambda ptr, n, dt: np.ctypeslib.as_array(ptr, shape=(n,)).astype(dt, copy=True) if n else np.empty(0, dt)  # noqa: E731
    if record is not None:
        words = record.view(p.words, p.padded_len // 16, C.c_uint32, np.uint32)
    else:
        words = arr(p.words, p.padded_len // 16, np.uint32)
    names = ()
    if want_names:
        off = arr(p.name_off, nc + 1, np.int32)
        blob = C.string_at(p.names, int(off[-1])) if nc else b""
        names = tuple(blob[off[i] : off[i + 1]].decode("utf-8", "replace") for i in range(nc))
    pa = PackedAssembly(words, int(p.padded_len), arr(p.ctg_start, nc, np.int32), arr(p.ctg_len, nc, np.int32),
                        arr(p.n_run_pairs, 2 * nr, np.int32).reshape(-1, 2))  # fmt: skip
    return pa, names


def pack_contigs(seqs: np.ndarray, offsets: np.ndarray, lengths: np.ndarray):
    """Contigs in memory (dense bytes + offsets + lengths) -> PackedAssembly, by the native packer (kp_pack_contigs;
    about ten times the numpy route)."""
    h = lib()
    h.kp_fasta_free.restype = None
    seqs, offsets, lengths = _c(seqs, np.uint8), _c(offsets, np.int64), _c(lengths, np.int32)
    out = C.POINTER(PackedFasta)()
    rc = h.kp_pack_contigs(_p(seqs), _p(offsets), _p(lengths), C.c_int32(len(lengths)), C.byref(out))
    if rc != 0:
        raise ValueError("assembly too long for the packed layout (KP_MAX_ASM_LEN)")
    try:
        return _packed_from(out, False)[0]
    finally:
        h.kp_fasta_free(out)


FASTA_FLAGS = {None: 0, "": 0, False: 0, True: 1, "gz": 1, "bz2": 4, "xz": 8}  # KP_FASTA_GZIP / _BZ2 / _XZ
ENOTSUP = -6
EIO = -7


def _host_inflate(data: bytes, compression) -> bytes:
    """bz2 / xz through Python's own modules: only for hosts whose libbz2 / liblzma the library could not load."""
    import bz2
    import lzma

    return (bz2.decompress if compression == "bz2" else lzma.decompress)(data)


def fasta_ingest(data: bytes, gzipped: "bool | str | None" = False, keep_text: bool = True):
    """A FASTA file's bytes -> (PackedAssembly, contig names, sequence text uint8, contig lengths int32) in one native
    pass (kp_fasta_ingest: inflate -- `gzipped` is True / "gz", "bz2" or "xz" --, split records, strip whitespace, pack;
    no GPU needed).  ``keep_text=False`` leaves the sequence text out (empty array): a run that only writes the TSV
    report never looks at it."""
    h = lib()
    h.kp_fasta_free.restype = None
    out = C.POINTER(PackedFasta)()
    text = 2 if keep_text else 0
    rc = h.kp_fasta_ingest(data, C.c_int64(len(data)), C.c_int32(FASTA_FLAGS[gzipped] | text), C.byref(out))
    if rc == ENOTSUP:
        data = _host_inflate(data, gzipped)
        rc = h.kp_fasta_ingest(data, C.c_int64(len(data)), C.c_int32(text), C.byref(out))
    if rc != 0:
        raise ValueError(f"kp_fasta_ingest failed ({rc}): not a readable FASTA / gzip stream, or longer than KP_MAX_ASM_LEN")
    record = _FastaRecord(out)  # frees the native record when the arrays below are gone
    pa, names = _packed_from(out, True, record)
    p = out.contents
    seqs = record.view(p.seqs, int(p.n_seq_bytes), C.c_uint8, np.uint8)
    return pa, names, seqs, pa.ctg_len.copy()


def fasta_ingest_file(path, gzipped: "bool | str | None" = False, keep_text: bool = True):
    """``fasta_ingest`` of a file by its path (kp_fasta_ingest_file): the library reads the file into a recycled buffer of its
    own, so its text never becomes a Python bytes object.  Falls back to reading the file here when the path is not a regular file or the
    host lacks the decompression library."""
    import os

    h = lib()
    h.kp_fasta_free.restype = None
    out = C.POINTER(PackedFasta)()
    text = 2 if keep_text else 0
    rc = h.kp_fasta_ingest_file(os.fsencode(path), C.c_int32(FASTA_FLAGS[gzipped] | text), C.byref(out))
    if rc in (ENOTSUP, EIO):
        with open(path, "rb") as f:
            return fasta_ingest(f.read(), gzipped, keep_text)
    if rc != 0:
        raise ValueError(f"kp_fasta_ingest_file failed ({rc}) for {path}: not a readable FASTA / compressed stream, or longer "
                         "than KP_MAX_ASM_LEN")
    record = _FastaRecord(out)
    pa, names = _packed_from(out, True, record)
    p = out.contents
    seqs = record.view(p.seqs, int(p.n_seq_bytes), C.c_uint8, np.uint8)
    return pa, names, seqs, pa.ctg_len.copy()


class _PackedShard(C.Structure):  # kp_packed_shard
    _fields_ = [("n_asm", C.c_int32), ("n_failed", C.c_int32), ("first_failed", C.c_int32), ("total_words", C.c_int64),
                ("asm_word_off", C.POINTER(C.c_int64)), ("ctg_start", C.POINTER(C.c_int32)), ("ctg_len", C.POINTER(C.c_int32)),
                ("asm_first_ctg", C.POINTER(C.c_int32)), ("n_runs", C.POINTER(C.c_int32)), ("asm_first_nrun", C.POINTER(C.c_int32)),
                ("rc", C.POINTER(C.c_int32))]  # fmt: skip


class FastaShard:
    """A chunk of FASTA files parsed and packed by the library's own threads (kp_fasta_ingest_shard), held as the tables
    of one batch: no per-file Python work at all -- with a Python call per file, names and arrays included, the interpreter
    lock capped a reader pool at ~6.5 k files/s whatever the parser did.  ``words_into`` copies the packed words into
    page-locked memory (kp_shard_words_into); ``Batch(ctx, tables=shard.tables(), pinned_words=...)`` uploads them."""

    def __init__(self, paths, compressions, threads: int = 0) -> None:
        import os

        h = lib()
        h.kp_shard_free.restype = None
        n = len(paths)
        self._paths = [os.fsencode(p) for p in paths]
        arr = (C.c_char_p * max(n, 1))(*self._paths)
        fl = (C.c_int32 * max(n, 1))(*[FASTA_FLAGS[c] for c in compressions])
        self._h = C.POINTER(_PackedShard)()
        rc = h.kp_fasta_ingest_shard(arr, fl, C.c_int32(n), C.c_int32(threads), C.byref(self._h))
        if rc != 0:
            raise MemoryError(f"kp_fasta_ingest_shard failed ({rc})")
        sh = self._h.contents
        self.n_asm, self.total_words = n, int(sh.total_words)
        self.failed = []
        if sh.n_failed:
            self.failed = [(i, int(sh.rc[i])) for i in range(n) if sh.rc[i] != 0]

    def tables(self) -> tuple:
        """(asm_word_off, ctg_start, ctg_len, asm_first_ctg, n_runs, asm_first_nrun): views of the library's arrays, valid
        while this object lives."""
        sh, n = self._h.contents, self.n_asm

        def view(ptr, k, dt):
            return np.ctypeslib.as_array(ptr, shape=(k,)) if k else np.empty(0, dt)

        first_ctg = view(sh.asm_first_ctg, n + 1, np.int32)
        first_run = view(sh.asm_first_nrun, n + 1, np.int32)
        nc, nr = (int(first_ctg[-1]), int(first_run[-1])) if n else (0, 0)
        return (view(sh.asm_word_off, n + 1, np.int64), view(sh.ctg_start, nc, np.int32), view(sh.ctg_len, nc, np.int32),
                first_ctg, view(sh.n_runs, 2 * nr, np.int32), first_run)

    def words_into(self, dst: np.ndarray, threads: int = 0) -> None:
        if dst.dtype != np.uint32 or not dst.flags.c_contiguous or len(dst) < self.total_words:
            raise ValueError("words_into needs a contiguous uint32 array of at least total_words")
        rc = lib().kp_shard_words_into(self._h, _p(dst), C.c_int64(len(dst)), C.c_int32(threads))
        if rc != 0:
            raise RuntimeError(f"kp_shard_words_into failed ({rc})")

    def close(self) -> None:
        if getattr(self, "_h", None):
            try:
                lib().kp_shard_free(self._h)
            except Exception:  # interpreter shutdown
                pass
            self._h = None

    __del__ = close


def device_count() -> int:
    """GPUs this process sees (kp_device_count)."""
    n = lib().kp_device_count()
    if n < 0:
        raise NativeError(f"kp_device_count failed ({n}): {lib().kp_last_error(None).decode()}")
    return int(n)


def compression_is_native() -> bool:
    """Whether the library found libbz2 and liblzma on this host (both tried with an empty stream)."""
    h = lib()
    out = C.POINTER(PackedFasta)()
    h.kp_fasta_free.restype = None
    ok = True
    for f in (4, 8):
        rc = h.kp_fasta_ingest(b"", C.c_int64(0), C.c_int32(f), C.byref(out))
        if rc == 0:
            h.kp_fasta_free(out)
        ok = ok and rc != ENOTSUP
    return ok


def fasta_ingest_many(datas: "list[bytes]", gzipped: "list[bool] | bool" = False, threads: int = 0) -> list:
    """``fasta_ingest`` of many files in one native call on the library's own threads (kp_fasta_ingest_many; 0 = one per
    core): a list of (PackedAssembly, names, sequence text, contig lengths), in input order.  Raises for the first file
    that could not be read."""
    h = lib()
    n = len(datas)
    if n == 0:
        return []
    flags = [gzipped] * n if isinstance(gzipped, (bool, str)) or gzipped is None else list(gzipped)
    if any(FASTA_FLAGS[g] & 12 for g in flags) and not compression_is_native():
        datas = [_host_inflate(d, g) if FASTA_FLAGS[g] & 12 else d for d, g in zip(datas, flags)]
        flags = [None if FASTA_FLAGS[g] & 12 else g for g in flags]
    ptrs = (C.c_char_p * n)(*datas)
    lens = (C.c_int64 * n)(*[len(d) for d in datas])
    fl = (C.c_int32 * n)(*[FASTA_FLAGS[g] | 2 for g in flags])
    outs = (C.POINTER(PackedFasta) * n)()
    rcs = (C.c_int32 * n)()
    rc = h.kp_fasta_ingest_many(ptrs, lens, fl, C.c_int32(n), C.c_int32(threads), outs, rcs)
    if rc != 0:
        raise ValueError(f"kp_fasta_ingest_many failed ({rc})")
    records = [_FastaRecord(C.cast(outs[i], C.POINTER(PackedFasta))) if rcs[i] == 0 else None for i in range(n)]
    result = []
    for i, rec in enumerate(records):
        if rec is None:
            raise ValueError(f"kp_fasta_ingest failed ({rcs[i]}) for file {i}: not a readable FASTA / gzip stream, or longer "
                             "than KP_MAX_ASM_LEN")
        pa, names = _packed_from(rec.out, True, rec)
        p = rec.out.contents
        result.append((pa, names, rec.view(p.seqs, int(p.n_seq_bytes), C.c_uint8, np.uint8), pa.ctg_len.copy()))
    return result


def fasta_pack(data: bytes):
    """FASTA text -> (PackedAssembly, contig names), parsed and packed by the native library (no GPU needed)."""
    h = lib()
    h.kp_fasta_free.restype = None
    out = C.POINTER(PackedFasta)()
    rc = h.kp_fasta_pack(data, C.c_int64(len(data)), C.byref(out))
    if rc != 0:
        raise ValueError(f"kp_fasta_pack failed ({rc}): sequence too long for the packed layout or bad arguments")
    try:
        return _packed_from(out, True)
    finally:
        h.kp_fasta_free(out)


class TypingTables(C.Structure):  # kp_typing_tables
    _fields_ = [("gene_locus", C.c_void_p), ("gene_extra", C.c_void_p), ("gene_pos", C.c_void_p),
                ("gene_strand", C.c_void_p), ("locus_gene_off", C.c_void_p), ("locus_gene_len", C.c_void_p),
                ("n_loci", C.c_int32), ("prot", C.c_void_p), ("prot_off", C.c_void_p), ("prot_len", C.c_void_p)]  # fmt: skip


class RowTables(C.Structure):  # kp_row_tables
    _fields_ = [("prefix", C.c_char_p), ("prefix_len", C.c_int32), ("gene_ids", C.c_void_p), ("gene_id_off", C.c_void_p),
                ("locus_names", C.c_void_p), ("locus_name_off", C.c_void_p), ("locus_gene_off", C.c_void_p),
                ("locus_gene_len", C.c_void_p)]  # fmt: skip


class RowColumns(C.Structure):  # kp_row_columns
    _fields_ = [("asm_ids", C.c_void_p), ("asm_id_off", C.c_void_p), ("phenotypes", C.c_void_p),
                ("phenotype_off", C.c_void_p), ("best_locus", C.c_void_p), ("typeable", C.c_void_p),
                ("problems", C.c_void_p), ("identity", C.c_void_p), ("coverage", C.c_void_p),
                ("length_discrepancy", C.c_void_p)]  # fmt: skip


def _blob(strings) -> tuple[np.ndarray, np.ndarray]:
    """utf-8 strings -> (bytes back to back, n + 1 offsets)."""
    enc = [s if isinstance(s, bytes) else str(s).encode("utf-8") for s in strings]
    off = np.zeros(len(enc) + 1, np.int32)
    if enc:
        np.cumsum([len(e) for e in enc], out=off[1:])
    return np.frombuffer(b"".join(enc) or b"\0", np.uint8), off


class RowFormatter:
    """KaptiveRow bytes for whole batches (kp_format_rows); the database's string tables are prepared once."""

    def __init__(self, db, kaptive_version: str) -> None:
        meta = db.metadata
        self._prefix = f"{kaptive_version}\t{meta.name}\t{meta.version}\t".encode("utf-8")
        self._keep = dict(
            gene_ids=_blob(db.genes.ids), locus_names=_blob(db.loci.ids),
            locus_gene_off=_c(db.locus_gene_offsets, np.int32), locus_gene_len=_c(db.locus_gene_lengths, np.int32),
        )  # fmt: skip
        k = self._keep
        self._tables = RowTables(
            prefix=self._prefix, prefix_len=len(self._prefix), gene_ids=_p(k["gene_ids"][0]).value,
            gene_id_off=_p(k["gene_ids"][1]).value, locus_names=_p(k["locus_names"][0]).value,
            locus_name_off=_p(k["locus_names"][1]).value, locus_gene_off=_p(k["locus_gene_off"]).value,
            locus_gene_len=_p(k["locus_gene_len"]).value,
        )  # fmt: skip

    def format(self, ids, phenotypes, sums, kept, best_locus, typeable, problems, identity, coverage, discrepancy) -> bytes:
        n = len(sums)
        if n == 0:
            return b""
        ids_b, ids_o = _blob(ids)
        ph_b, ph_o = _blob(phenotypes)
        cols = dict(best=_c(best_locus, np.int32), typeable=_c(typeable, np.uint8), problems=_c(problems, np.int32),
                    identity=_c(identity, np.float64), coverage=_c(coverage, np.float64),
                    discrepancy=_c(discrepancy, np.float64))  # fmt: skip
        c = RowColumns(asm_ids=_p(ids_b).value, asm_id_off=_p(ids_o).value, phenotypes=_p(ph_b).value,
                       phenotype_off=_p(ph_o).value, best_locus=_p(cols["best"]).value, typeable=_p(cols["typeable"]).value,
                       problems=_p(cols["problems"]).value, identity=_p(cols["identity"]).value,
                       coverage=_p(cols["coverage"]).value, length_discrepancy=_p(cols["discrepancy"]).value)  # fmt: skip
        sums, kept = np.ascontiguousarray(sums), np.ascontiguousarray(kept)
        stride = kept.shape[1] if kept.ndim == 2 else 0
        h = lib()
        h.kp_format_rows.restype = C.c_int64
        out = np.empty(max(4096, 1200 * n), np.uint8)
        for _ in range(2):
            need = h.kp_format_rows(C.byref(self._tables), C.c_int32(n), _p(sums), _p(kept), C.c_int32(stride), C.byref(c),
                                    _p(out), C.c_int64(len(out)))
            if need < 0:
                raise ValueError(f"kp_format_rows failed ({need})")
            if need <= len(out):
                return out[:need].tobytes()
            out = np.empty(int(need), np.uint8)
        raise NativeError("kp_format_rows: size kept changing")


class JsonTables(C.Structure):  # kp_json_tables
    _fields_ = [("head", C.c_char_p), ("head_len", C.c_int32), ("gene_names", C.c_void_p), ("gene_name_off", C.c_void_p),
                ("gene_ids", C.c_void_p), ("cluster_names", C.c_void_p), ("products", C.c_void_p), ("gene_id_off", C.c_void_p),
                ("cluster_name_off", C.c_void_p), ("product_off", C.c_void_p), ("locus_names", C.c_void_p),
                ("locus_name_off", C.c_void_p), ("locus_gene_off", C.c_void_p), ("locus_gene_len", C.c_void_p),
                ("gene_position", C.c_void_p), ("gene_strand", C.c_void_p), ("comp_map", C.c_void_p), ("char_map", C.c_void_p),
                ("codon_map", C.c_void_p), ("n_loci", C.c_int32), ("n_genes", C.c_int32)]  # fmt: skip


class JsonColumns(C.Structure):  # kp_json_columns
    _fields_ = [("asm_ids", C.c_void_p), ("asm_id_off", C.c_void_p), ("phenotypes", C.c_void_p), ("phenotype_off", C.c_void_p),
                ("best_locus", C.c_void_p), ("typeable", C.c_void_p), ("problems", C.c_void_p), ("best_score", C.c_void_p),
                ("completeness", C.c_void_p), ("identity", C.c_void_p), ("coverage", C.c_void_p), ("length_discrepancy", C.c_void_p),
                ("piece_order", C.c_void_p), ("piece_ctg_names", C.c_void_p), ("piece_ctg_name_off", C.c_void_p),
                ("ctg_seqs", C.c_void_p), ("ctg_off", C.c_void_p), ("n_ctg", C.c_void_p), ("ctg_text_len", C.c_void_p)]  # fmt: skip


def _blob64(strings) -> tuple[np.ndarray, np.ndarray]:
    """utf-8 strings -> (bytes back to back, n + 1 int64 offsets)."""
    enc = [s if isinstance(s, bytes) else str(s).encode("utf-8") for s in strings]
    off = np.zeros(len(enc) + 1, np.int64)
    if enc:
        np.cumsum([len(e) for e in enc], out=off[1:])
    return np.frombuffer(b"".join(enc) or b"\0", np.uint8), off


class JsonFormatter:
    """The JSON lines of whole batches (kp_format_json): what ``dumps_line(result.to_dict())`` writes per assembly
    (kaptive_amd/serotyping/jsonl.py), without an object per assembly.  The database's strings are escaped once."""

    def __init__(self, typer, kaptive_version: str) -> None:
        from kaptive_amd.core.seq import CHAR_MAP, CODON_MAP, COMP_MAP
        from kaptive_amd.serotyping.jsonl import _str

        db = typer._db
        self._typer_gene_ids = tuple(db.genes.ids)
        meta = db.metadata
        head = ("{" + f'"kaptive_version":{_str(kaptive_version)},"database_name":{_str(meta.name)},"database_version":{_str(meta.version)},'
                f'"database_organism":{_str(meta.organism)},"database_taxon":{int(meta.taxon)},"genome":').encode("utf-8")
        text = lambda col: [_str(v.decode("utf-8", "replace")) for v in col.tolist()]  # noqa: E731  (GeneHits.to_dict decodes the truncated bytes)
        self._keep = dict(
            head=head, gene_names=_blob64([_str(x) for x in db.genes.ids]), gene_ids=_blob64(text(typer._gene_ids_s32)),
            cluster=_blob64(text(typer._cluster_s10)), product=_blob64(text(typer._product_s64)),
            locus=_blob64([_str(x) for x in db.loci.ids]),
            locus_gene_off=_c(db.locus_gene_offsets, np.int32), locus_gene_len=_c(db.locus_gene_lengths, np.int32),
            gene_position=_c(db.gene_positions, np.int32), gene_strand=_c(db.gene_intervals.strands, np.int8),
            comp=_c(COMP_MAP, np.uint8), char=_c(CHAR_MAP, np.uint8), codon=_c(CODON_MAP, np.uint8),
        )  # fmt: skip
        k = self._keep
        self._tables = JsonTables(
            head=head, head_len=len(head), gene_names=_p(k["gene_names"][0]).value, gene_name_off=_p(k["gene_names"][1]).value,
            gene_ids=_p(k["gene_ids"][0]).value, cluster_names=_p(k["cluster"][0]).value, products=_p(k["product"][0]).value,
            gene_id_off=_p(k["gene_ids"][1]).value, cluster_name_off=_p(k["cluster"][1]).value, product_off=_p(k["product"][1]).value,
            locus_names=_p(k["locus"][0]).value, locus_name_off=_p(k["locus"][1]).value,
            locus_gene_off=_p(k["locus_gene_off"]).value, locus_gene_len=_p(k["locus_gene_len"]).value,
            gene_position=_p(k["gene_position"]).value, gene_strand=_p(k["gene_strand"]).value, comp_map=_p(k["comp"]).value,
            char_map=_p(k["char"]).value, codon_map=_p(k["codon"]).value, n_loci=len(db.loci.ids), n_genes=len(db.genes.ids),
        )  # fmt: skip

    def _prepare(self, ids, phenotypes, sums, kept, pieces, best_locus, best_score, completeness, typeable, problems, identity,
                 coverage, discrepancy, genomes) -> dict:
        """The batch's columns as kp_json_columns (and everything that must stay alive while the library reads them)."""
        from kaptive_amd.serotyping.jsonl import _str

        n = len(sums)
        sums, kept, pieces = np.ascontiguousarray(sums), np.ascontiguousarray(kept), np.ascontiguousarray(pieces)
        kstride = kept.shape[1] if kept.ndim == 2 else 0
        pstride = pieces.shape[1] if pieces.ndim == 2 else 0
        # order of the locus pieces: numpy's argsort of their mean positions (core.py:281), per assembly; names of their contigs
        order = np.zeros((n, max(pstride, 1)), np.int32)
        order[:] = np.arange(max(pstride, 1), dtype=np.int32)[None, :]
        names = [b""] * (n * pstride)
        raw_names = [b""] * (n * pstride)
        n_pieces = sums["n_pieces"]
        if n and (int(n_pieces.min()) < 0 or int(n_pieces.max()) > pstride):
            raise ValueError("kp_format_json: a summary counts more locus pieces than the piece table holds")
        for a in np.flatnonzero(n_pieces > 0):
            m = int(n_pieces[a])
            if m > 1:
                order[a, :m] = np.argsort(np.ascontiguousarray(pieces["mean_pos"][a, :m]))
            cids = genomes[a].contigs.ids
            for p_ in range(m):
                ci = int(pieces["contig"][a, p_])
                if not 0 <= ci < len(cids):
                    raise ValueError("kp_format_json: a locus piece names a contig the assembly does not have")
                cid = cids[ci]
                names[a * pstride + p_] = _str(cid)[1:-1].encode("utf-8")
                raw_names[a * pstride + p_] = cid.encode()
        ids_b, ids_o = _blob64([_str(x) for x in ids])
        ph_b, ph_o = _blob64([_str(x) for x in phenotypes])
        nm_b, nm_o = _blob64(names)
        seq_arrays = [np.ascontiguousarray(g.contigs.seqs, dtype=np.uint8) for g in genomes]
        off_arrays = [np.ascontiguousarray(g.contigs.offsets, dtype=np.int32) for g in genomes]
        seq_ptrs = (C.c_void_p * max(n, 1))(*[a.ctypes.data for a in seq_arrays])
        off_ptrs = (C.c_void_p * max(n, 1))(*[a.ctypes.data for a in off_arrays])
        n_ctg = np.array([len(o) for o in off_arrays], np.int32)
        text_len = np.array([len(a) for a in seq_arrays], np.int64)
        cols = dict(n_ctg=n_ctg, text_len=text_len, best=_c(best_locus, np.int32), typeable=_c(typeable, np.uint8), problems=_c(problems, np.int32),
                    score=_c(best_score, np.float64), compl=_c(completeness, np.float64), identity=_c(identity, np.float64),
                    coverage=_c(coverage, np.float64), discrepancy=_c(discrepancy, np.float64), order=order)  # fmt: skip
        c = JsonColumns(asm_ids=_p(ids_b).value, asm_id_off=_p(ids_o).value, phenotypes=_p(ph_b).value, phenotype_off=_p(ph_o).value,
                        best_locus=_p(cols["best"]).value, typeable=_p(cols["typeable"]).value, problems=_p(cols["problems"]).value,
                        best_score=_p(cols["score"]).value, completeness=_p(cols["compl"]).value, identity=_p(cols["identity"]).value,
                        coverage=_p(cols["coverage"]).value, length_discrepancy=_p(cols["discrepancy"]).value,
                        piece_order=_p(order).value, piece_ctg_names=_p(nm_b).value, piece_ctg_name_off=_p(nm_o).value,
                        ctg_seqs=C.cast(seq_ptrs, C.c_void_p).value, ctg_off=C.cast(off_ptrs, C.c_void_p).value,
                        n_ctg=_p(n_ctg).value, ctg_text_len=_p(text_len).value)  # fmt: skip
        return dict(n=n, sums=sums, kept=kept, pieces=pieces, kstride=kstride, pstride=pstride, c=c, raw_piece_names=raw_names,
                    keep=(ids_b, ids_o, ph_b, ph_o, nm_b, nm_o, seq_arrays, off_arrays, seq_ptrs, off_ptrs, cols))

    def format(self, *columns) -> bytes:
        """``columns``: ids, phenotypes, sums, kept, pieces, best_locus, best_score, completeness, typeable, problems, identity,
        coverage, length discrepancy, genomes (the attributes of a ``BatchTyping``)."""
        if len(columns[2]) == 0:
            return b""
        q = self._prepare(*columns)
        h = lib()
        h.kp_format_json.restype = C.c_int64
        out = np.empty(max(1 << 16, 80_000 * q["n"]), np.uint8)
        for _ in range(2):
            need = h.kp_format_json(C.byref(self._tables), C.c_int32(q["n"]), _p(q["sums"]), _p(q["kept"]), C.c_int32(q["kstride"]),
                                    _p(q["pieces"]), C.c_int32(q["pstride"]), C.byref(q["c"]), _p(out), C.c_int64(len(out)))
            if need < 0:
                raise ValueError(f"kp_format_json failed ({need})")
            if need <= len(out):
                return out[:need].tobytes()
            out = np.empty(int(need), np.uint8)
        raise NativeError("kp_format_json: size kept changing")

    def fasta(self, kinds, *columns) -> dict:
        """Per-assembly FASTA bytes (kp_format_fasta) for every kind asked for -- "loci", "genes", "proteins": what
        ``result.locus_seqs / gene_seqs / translations .to_fasta()`` give per result -- as ``{kind: [bytes per assembly]}``."""
        n = len(columns[2])
        if n == 0:
            return {kind: [] for kind in kinds}
        q = self._prepare(*columns)
        if "gene_names_raw" not in self._keep:
            self._keep["gene_names_raw"] = _blob64([x.encode() for x in self._typer_gene_ids])
        piece_b, piece_o = _blob64(q["raw_piece_names"])
        h = lib()
        h.kp_format_fasta.restype = C.c_int64
        out = {}
        for kind in kinds:
            code = {"loci": 0, "genes": 1, "proteins": 2}[kind]
            nb, no = (piece_b, piece_o) if code == 0 else self._keep["gene_names_raw"]
            ends = np.zeros(n, np.int64)
            buf = np.empty(max(1 << 16, (40_000 if code < 2 else 12_000) * n), np.uint8)
            for _ in range(2):
                need = h.kp_format_fasta(C.byref(self._tables), C.c_int32(n), _p(q["sums"]), _p(q["kept"]), C.c_int32(q["kstride"]),
                                         _p(q["pieces"]), C.c_int32(q["pstride"]), C.byref(q["c"]), C.c_int32(code), _p(nb), _p(no),
                                         _p(buf), C.c_int64(len(buf)), _p(ends))
                if need < 0:
                    raise ValueError(f"kp_format_fasta failed ({need})")
                if need <= len(buf):
                    break
                buf = np.empty(int(need), np.uint8)
            else:
                raise NativeError("kp_format_fasta: size kept changing")
            blob = buf[:need].tobytes()
            starts = np.concatenate([[0], ends[:-1]])
            out[kind] = [blob[int(a):int(b)] for a, b in zip(starts, ends)]
        return out


class TypingParams(C.Structure):  # kp_typing_params
    _fields_ = [("min_gene_coverage", C.c_double), ("id_threshold", C.c_float), ("max_locus_length", C.c_int32),
                ("edge_tolerance", C.c_int32)]  # fmt: skip

_lib = None
_lock = threading.Lock()
_live: "weakref.WeakSet" = weakref.WeakSet()  # contexts and batches still holding device memory


@atexit.register
def _close_all() -> None:
    """Release device objects before the HIP runtime's own exit handlers run (batches first, then contexts)."""
    objs = list(_live)
    for o in sorted(objs, key=lambda x: isinstance(x, Context)):
        try:
            o.close()
        except Exception:
            pass


class NativeError(RuntimeError):
    pass


def is_loaded() -> bool:
    """Whether the shared library (and with it the HIP runtime) has been loaded into this process."""
    return _lib is not None


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        with _lock:
            if _lib is None:
                if not LIB_PATH.exists():
                    raise NativeError(
                        f"{LIB_PATH} is missing: build it with `python -m kaptive_amd.build` (hipcc, gfx950). "
                        "kaptive_amd has no CPU fallback."
                    )
                # The HIP runtime reads GPU_MAX_HW_QUEUES when it initialises, i.e. when the library below pulls it in: a
                # context drives a dozen streams and the default of 4 hardware queues costs a host-fed stream a fifth of
                # its throughput (kaptive_amd.tune_runtime).  Library users who never call an entry point get it here;
                # an explicit setting in the environment wins.
                os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
                h = C.CDLL(str(LIB_PATH))
                h.kp_last_error.restype = C.c_char_p
                h.kp_ctx_stream.restype = C.c_void_p
                for f in ("kp_db_n_postings", "kp_batch_anchors", "kp_batch_tasks", "kp_batch_task_results", "kp_batch_joins"):
                    getattr(h, f).restype = C.c_int64
                h.kp_ctx_destroy.restype = None
                h.kp_batch_destroy.restype = None
                h.kp_batch_device_words.restype = C.c_void_p
                h.kp_host_free.restype = None
                h.kp_host_free.argtypes = [C.c_void_p]
                h.kp_host_alloc.argtypes = [C.c_size_t, C.POINTER(C.c_void_p)]
                h.kp_host_reserve.argtypes = [C.c_size_t, C.POINTER(C.c_void_p)]
                h.kp_host_lock.argtypes = [C.c_void_p]
                _lib = h
    return _lib


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _c(a, dt):
    return np.ascontiguousarray(a, dtype=dt)


class Context:
    """One GPU + stream + resident database (kp_ctx)."""

    def __init__(self, device: int = 0) -> None:
        self._h = C.c_void_p()
        rc = lib().kp_ctx_create(C.c_int(device), C.byref(self._h))
        if rc != 0:
            raise NativeError(f"kp_ctx_create failed ({rc}): {lib().kp_last_error(None).decode()}")
        self.device = device
        self.group_loci: dict[int, int] = {}  # loci of every typing group's database (shapes of the score arrays)
        self._batches: "weakref.WeakSet" = weakref.WeakSet()
        _live.add(self)

    def close(self) -> None:
        if getattr(self, "_h", None):
            for b in list(self._batches):  # a batch must not outlive its context
                b.close()
            lib().kp_ctx_destroy(self._h)
            self._h = None

    __del__ = close

    def _check(self, rc: int, what: str) -> None:
        if rc < 0:
            msg = lib().kp_last_error(self._h).decode()
            raise (ValueError if rc == -1 else NativeError)(f"{what} failed ({rc}): {msg}")

    @property
    def stream(self) -> int:
        return int(lib().kp_ctx_stream(self._h) or 0)

    def set_option(self, name: str, value: int) -> None:
        """Tuning knob of the context (kp_ctx_set_option; names in include/kaptive_amd.h)."""
        self._check(lib().kp_ctx_set_option(self._h, name.encode(), C.c_int64(int(value))), "kp_ctx_set_option")

    def load_genes(self, gene_codes: np.ndarray, gene_off: np.ndarray) -> None:
        codes, off = _c(gene_codes, np.uint8), _c(gene_off, np.int32)
        self._check(lib().kp_db_load(self._h, _p(codes), _p(off), C.c_int32(len(off) - 1)), "kp_db_load")

    def load_typing(self, db, group: int = 0, gene_lo: int = 0, gene_hi: int | None = None) -> None:
        """Upload the Database columns the batched reduction reads (kp_db_load_typing / kp_db_load_typing_group).  With
        several databases in one context, ``db``'s genes are ``[gene_lo, gene_hi)`` of the genes given to
        ``load_genes``."""
        prot = db.translations
        keep = dict(
            gene_locus=_c(db.gene_locus_indices, np.uint16), gene_extra=_c(db.extra_genes, np.uint8),
            gene_pos=_c(db.gene_positions, np.uint16), gene_strand=_c(db.gene_intervals.strands, np.int8),
            locus_gene_off=_c(db.locus_gene_offsets, np.int32), locus_gene_len=_c(db.locus_gene_lengths, np.int32),
            prot=_c(prot.seqs, np.uint8), prot_off=_c(prot.offsets, np.int32), prot_len=_c(prot.lengths, np.int32),
        )  # fmt: skip
        t = TypingTables(n_loci=len(db.loci), **{k: _p(v).value for k, v in keep.items()})
        gene_hi = gene_lo + len(db.genes) if gene_hi is None else gene_hi
        self._check(
            lib().kp_db_load_typing_group(self._h, C.c_int32(group), C.c_int32(gene_lo), C.c_int32(gene_hi), C.byref(t)),
            "kp_db_load_typing_group",
        )
        self.group_loci[group] = len(db.loci)
        if group == 0:
            self.n_loci = len(db.loci)

    @property
    def n_postings(self) -> int:
        return int(lib().kp_db_n_postings(self._h))

    def protein_align(self, q, q_off, q_len, t, t_off, t_len) -> np.ndarray:
        n = len(q_off)
        out = np.zeros((n, 8), np.int32)
        if n:
            q, t = _c(q, np.uint8), _c(t, np.uint8)
            self._check(
                lib().kp_protein_align(self._h, _p(q), _p(_c(q_off, np.int32)), _p(_c(q_len, np.int32)), _p(t),
                                       _p(_c(t_off, np.int32)), _p(_c(t_len, np.int32)), C.c_int32(n), _p(out)),
                "kp_protein_align",
            )  # fmt: skip
        return out

    def protein_align_seeded(self, q, q_off, q_len, t, t_off, t_len, offsets, k: int) -> np.ndarray:
        """Seeded mode (kp_protein_align_seeded): band ``k`` around the diagonal ``offsets[p]`` of every pair."""
        n = len(q_off)
        out = np.zeros((n, 8), np.int32)
        if n:
            q, t = _c(q, np.uint8), _c(t, np.uint8)
            self._check(
                lib().kp_protein_align_seeded(self._h, _p(q), _p(_c(q_off, np.int32)), _p(_c(q_len, np.int32)), _p(t),
                                              _p(_c(t_off, np.int32)), _p(_c(t_len, np.int32)), C.c_int32(n),
                                              _p(_c(offsets, np.int32)), C.c_int32(int(k)), _p(out)),
                "kp_protein_align_seeded",
            )  # fmt: skip
        return out

    def batch(self, packed: list, device_words: int | None = None, pinned_words: "np.ndarray | None" = None,
              after: "Batch | None" = None, tables: "tuple | None" = None) -> "Batch":
        return Batch(self, packed, device_words, pinned_words, after, tables)


class Batch:
    """Packed assemblies resident on the device (kp_batch). ``packed`` is a list of PackedAssembly; with
    ``device_words`` (a device pointer to the concatenated words) nothing but the small tables is copied."""

    def __init__(self, ctx: Context, packed: "list | None", device_words: int | None = None,
                 pinned_words: "np.ndarray | None" = None, after: "Batch | None" = None, tables: "tuple | None" = None) -> None:
        """``after``: the batch (of another context on the same GPU) whose device words this one adopts while their
        upload may still be in flight.  ``pinned_words``: the concatenated words of ``packed`` in page-locked memory (``pinned_array``); the upload
        is then enqueued asynchronously (kp_batch_create_async) and the array must stay alive until ``upload_wait`` or
        the first ``wait``/``score`` of the batch.  ``tables`` (instead of ``packed``): the batch's tables as they go to
        the library -- ``FastaShard.tables()`` -- next to ``pinned_words`` or ``device_words``."""
        self.ctx = ctx

        def cat(xs, dt):
            return np.ascontiguousarray(np.concatenate(xs), dtype=dt) if xs else np.empty(0, dt)

        if tables is not None:
            if packed is not None or (pinned_words is None and device_words is None):
                raise ValueError("tables come with pinned_words or device_words, not with packed assemblies")
            word_off, ctg_start, ctg_len, first_ctg, n_runs, first_run = tables
            self.n_asm = len(word_off) - 1
        else:
            self.n_asm = len(packed)
            word_off = np.zeros(self.n_asm + 1, np.int64)
            first_ctg = np.zeros(self.n_asm + 1, np.int32)
            first_run = np.zeros(self.n_asm + 1, np.int32)
            for i, pa in enumerate(packed):
                word_off[i + 1] = word_off[i] + pa.padded_len // 16
                first_ctg[i + 1] = first_ctg[i] + len(pa.ctg_start)
                first_run[i + 1] = first_run[i] + len(pa.n_runs)
            ctg_start = cat([pa.ctg_start for pa in packed], np.int32)
            ctg_len = cat([pa.ctg_len for pa in packed], np.int32)
            n_runs = cat([pa.n_runs.reshape(-1) for pa in packed], np.int32)
        self._h = C.c_void_p()
        self._pinned = pinned_words
        if pinned_words is not None:
            if pinned_words.dtype != np.uint32 or len(pinned_words) != int(word_off[-1]):
                raise ValueError("pinned_words must be the uint32 concatenation of the packed assemblies' words")
            rc = lib().kp_batch_create_async(ctx._h, C.c_int32(self.n_asm), _p(pinned_words), _p(word_off), _p(ctg_start),
                                             _p(ctg_len), _p(first_ctg), _p(n_runs), _p(first_run), C.byref(self._h))  # fmt: skip
        elif device_words is None:
            words = cat([pa.words for pa in packed], np.uint32)
            rc = lib().kp_batch_create(ctx._h, C.c_int32(self.n_asm), _p(words), _p(word_off), _p(ctg_start),
                                       _p(ctg_len), _p(first_ctg), _p(n_runs), _p(first_run), C.byref(self._h))  # fmt: skip
        else:
            rc = lib().kp_batch_create_device(ctx._h, C.c_int32(self.n_asm), C.c_void_p(device_words), _p(word_off),
                                              _p(ctg_start), _p(ctg_len), _p(first_ctg), _p(n_runs), _p(first_run),
                                              C.byref(self._h))  # fmt: skip
        ctx._check(rc, "kp_batch_create")
        self._after = after
        if after is not None:
            ctx._check(lib().kp_batch_depends_on(ctx._h, self._h, after._h), "kp_batch_depends_on")
        self.total_words = int(word_off[-1])
        ctx._batches.add(self)
        _live.add(self)

    def close(self) -> None:
        if getattr(self, "_h", None):
            if getattr(self.ctx, "_h", None):  # the context owns the device; without it there is nothing to free
                lib().kp_batch_destroy(self._h)
            self._h = None

    __del__ = close

    @property
    def device_words(self) -> int:
        """Device address of the packed words (kp_batch_device_words); another context's batch can adopt them."""
        return int(lib().kp_batch_device_words(self._h) or 0)

    def upload_wait(self) -> None:
        self.ctx._check(lib().kp_batch_upload_wait(self.ctx._h, self._h), "kp_batch_upload_wait")

    def align_async(self) -> None:
        self.ctx._check(lib().kp_batch_align(self.ctx._h, self._h), "kp_batch_align")

    def wait(self) -> None:
        self.ctx._check(lib().kp_batch_wait(self.ctx._h, self._h), "kp_batch_wait")

    def align(self) -> tuple[np.ndarray, np.ndarray]:
        """Run the aligner; returns (hits [HIT_DTYPE], hit_off [n_asm + 1])."""
        self.align_async()
        self.wait()
        return self.hits()

    def hits(self) -> tuple[np.ndarray, np.ndarray]:
        off = np.zeros(self.n_asm + 1, np.int64)
        self.ctx._check(lib().kp_batch_hit_offsets(self.ctx._h, self._h, _p(off)), "kp_batch_hit_offsets")
        out = np.zeros(int(off[-1]), HIT_DTYPE)
        self.ctx._check(lib().kp_batch_hits(self.ctx._h, self._h, _p(out), C.c_int64(len(out))), "kp_batch_hits")
        return out, off

    def set_hits(self, hits: np.ndarray, off: np.ndarray) -> None:
        """Replace the finished hit table by the caller's (kp_batch_set_hits): rows ``off[a]:off[a + 1]`` are assembly a's."""
        hits = np.ascontiguousarray(hits, dtype=HIT_DTYPE)
        off = np.ascontiguousarray(off, dtype=np.int64)
        if len(off) != self.n_asm + 1 or int(off[-1]) != len(hits):
            raise ValueError("offsets do not describe the hit table")
        self.ctx._check(lib().kp_batch_set_hits(self.ctx._h, self._h, _p(hits), _p(off)), "kp_batch_set_hits")

    def stats(self) -> dict[str, int]:
        s = np.zeros(5, np.int64)
        self.ctx._check(lib().kp_batch_stats(self.ctx._h, self._h, _p(s)), "kp_batch_stats")
        return dict(zip(("anchors", "tasks", "dp_cells", "hits", "retries"), s.tolist()))

    # -- batched reduction ------------------------------------------------------------------------------------------
    def score(self, min_gene_coverage: float, group: int = 0) -> tuple[np.ndarray, np.ndarray]:
        """(locus_scores f64, locus_counts i32), both [n_asm, n_loci]; finalises the hit tables on the device."""
        self.use_group(group)
        n_loci = self.ctx.group_loci[group]
        scores = np.zeros((self.n_asm, n_loci), np.float64)
        counts = np.zeros((self.n_asm, n_loci), np.int32)
        self.ctx._check(
            lib().kp_batch_score(self.ctx._h, self._h, C.c_double(min_gene_coverage), _p(scores), _p(counts)),
            "kp_batch_score",
        )
        return scores, counts

    def use_group(self, group: int) -> None:
        """The typing group (database) the score / reduce / typing / proteins calls address (kp_batch_use_group)."""
        self.ctx._check(lib().kp_batch_use_group(self.ctx._h, self._h, C.c_int32(group)), "kp_batch_use_group")

    def reduce_async(self, best_locus: np.ndarray, params: "TypingParams", group: int = 0) -> None:
        self.use_group(group)
        best = _c(best_locus, np.int32)
        self.ctx._check(lib().kp_batch_reduce(self.ctx._h, self._h, _p(best), C.byref(params)), "kp_batch_reduce")

    def typing(self, group: int = 0):
        """(summaries [n_asm], kept [n_asm, kept_cap], pieces [n_asm, piece_cap]) as structured arrays."""
        from kaptive_amd.serotyping.batch import KEPT_DTYPE, PIECE_DTYPE, SUMMARY_DTYPE

        self.use_group(group)
        for _ in range(2):  # capacities may grow inside kp_batch_typing when a buffer overflowed
            kc, pc = C.c_int32(0), C.c_int32(0)
            self.ctx._check(lib().kp_batch_typing_caps(self.ctx._h, self._h, C.byref(kc), C.byref(pc)), "typing_caps")
            sums = np.zeros(self.n_asm, SUMMARY_DTYPE)
            kept = np.zeros((self.n_asm, kc.value), KEPT_DTYPE)
            pieces = np.zeros((self.n_asm, pc.value), PIECE_DTYPE)
            rc = lib().kp_batch_typing(self.ctx._h, self._h, _p(sums), _p(kept), kc, _p(pieces), pc)
            if rc == -1 and b"strides too small" in lib().kp_last_error(self.ctx._h):
                continue
            self.ctx._check(rc, "kp_batch_typing")
            return sums, kept, pieces
        raise NativeError("kp_batch_typing: capacities kept changing")

    def proteins(self, asm_index: int, nbytes: int, group: int = 0) -> np.ndarray:
        self.use_group(group)
        out = np.zeros(nbytes, np.uint8)
        self.ctx._check(lib().kp_batch_proteins(self.ctx._h, self._h, C.c_int32(asm_index), _p(out), C.c_int64(nbytes)),
                        "kp_batch_proteins")  # fmt: skip
        return out

    def profile(self) -> dict[str, float]:
        """Stage milliseconds of the most recent alignment pass (HIP events recorded on the context's stream)."""
        ms = np.zeros(7, np.float32)
        nbytes = C.c_int64(0)
        self.ctx._check(lib().kp_batch_profile(self.ctx._h, self._h, _p(ms), C.byref(nbytes)), "kp_batch_profile")
        d = dict(zip(("scan", "sort", "chain", "sw16", "sw32", "sw64", "sw128"), ms.tolist()))
        d["bytes_scanned"] = int(nbytes.value)
        return d

    def anchors(self, asm_index: int) -> np.ndarray:
        n = lib().kp_batch_anchors(self.ctx._h, self._h, C.c_int32(asm_index), None, C.c_int64(0))
        self.ctx._check(n, "kp_batch_anchors")
        out = np.zeros(n, np.uint64)
        lib().kp_batch_anchors(self.ctx._h, self._h, C.c_int32(asm_index), _p(out), C.c_int64(n))
        return out

    def tasks(self, asm_index: int) -> np.ndarray:
        n = lib().kp_batch_tasks(self.ctx._h, self._h, C.c_int32(asm_index), None, C.c_int64(0))
        self.ctx._check(n, "kp_batch_tasks")
        out = np.zeros(n, TASK_DTYPE)
        lib().kp_batch_tasks(self.ctx._h, self._h, C.c_int32(asm_index), _p(out), C.c_int64(n))
        return out


    def task_results(self, asm_index: int) -> np.ndarray:
        """[n_tasks, 7] int32 -- score, q_start, q_end, t_start, t_end, matches, block_len -- row for row with ``tasks``;
        tasks below the score cut-off carry their score and zeros (kp_batch_task_results)."""
        n = lib().kp_batch_task_results(self.ctx._h, self._h, C.c_int32(asm_index), None, C.c_int64(0))
        self.ctx._check(n, "kp_batch_task_results")
        out = np.zeros((n, 7), np.int32)
        lib().kp_batch_task_results(self.ctx._h, self._h, C.c_int32(asm_index), _p(out), C.c_int64(n))
        return out


    def joins(self, asm_index: int) -> np.ndarray:
        """The joins of one assembly (kp_spec.h, kp-align v5) as JOIN_DTYPE rows, in device order (kp_batch_joins)."""
        n = lib().kp_batch_joins(self.ctx._h, self._h, C.c_int32(asm_index), None, C.c_int64(0))
        self.ctx._check(n, "kp_batch_joins")
        out = np.zeros(n, JOIN_DTYPE)
        lib().kp_batch_joins(self.ctx._h, self._h, C.c_int32(asm_index), _p(out), C.c_int64(n))
        return out


RANDSTROBE_DTYPE = np.dtype([("hash", "<u8"), ("seq_idx", "<u4"), ("pos1", "<u4"), ("pos2", "<u4")])


def randstrobes(seqs, offsets, lengths, lut, k: int, s: int, w_min: int, w_max: int, sort_by_hash: bool) -> np.ndarray:
    """Randstrobe records of a batch of sequences (kp_randstrobes; host only)."""
    h = lib()
    h.kp_randstrobes.restype = C.c_int64
    seqs, offsets, lengths, lut = _c(seqs, np.uint8), _c(offsets, np.int32), _c(lengths, np.int32), _c(lut, np.uint8)
    args = (_p(seqs), _p(offsets), _p(lengths), C.c_int32(len(offsets)), _p(lut), C.c_int32(k), C.c_int32(s),
            C.c_int32(w_min), C.c_int32(w_max), C.c_int32(int(sort_by_hash)))  # fmt: skip
    n = h.kp_randstrobes(*args, None, C.c_int64(0))
    if n < 0:
        raise ValueError(f"kp_randstrobes failed ({n})")
    out = np.empty(int(n), RANDSTROBE_DTYPE)
    if n:
        h.kp_randstrobes(*args, _p(out), C.c_int64(n))
    return out


def randstrobe_top_hits(q_records: np.ndarray, n_queries: int, t_records: np.ndarray, n_targets: int):
    """(best target u32, score u32, diagonal offset i32) per query sequence (kp_randstrobe_top_hits; host only)."""
    q_records, t_records = np.ascontiguousarray(q_records), np.ascontiguousarray(t_records)
    best_t, score = np.zeros(n_queries, np.uint32), np.zeros(n_queries, np.uint32)
    off = np.zeros(n_queries, np.int32)
    rc = lib().kp_randstrobe_top_hits(_p(q_records), C.c_int64(len(q_records)), C.c_int32(n_queries), _p(t_records),
                                      C.c_int64(len(t_records)), C.c_int32(n_targets), _p(best_t), _p(score), _p(off))  # fmt: skip
    if rc != 0:
        raise ValueError(f"kp_randstrobe_top_hits failed ({rc})")
    return best_t, score, off


def pinned_bytes() -> int:
    """Page-locked host bytes this process holds through the library (kp_host_pinned_bytes)."""
    f = lib().kp_host_pinned_bytes
    f.restype = C.c_int64
    return int(f())


def device_allocations() -> int:
    """How many device buffers the library has re-allocated so far (kp_device_allocations): each one stalls every pass
    in flight, so a settled stream of batches must not add to it."""
    f = lib().kp_device_allocations
    f.restype = C.c_int64
    return int(f())


class PinnedBuffer:
    """Page-locked host memory (kp_host_alloc) exposed as a numpy array; freed by ``close`` / garbage collection.
    ``lazy``: plain huge-page memory for now (kp_host_reserve: no call into the device runtime), page-locked by ``lock()``
    once the caller's threads have filled it -- the first touch is then theirs, spread over as many threads as they are,
    and the lock itself takes 2 ms per GB."""

    def __init__(self, n_items: int, dtype=np.uint32, lazy: bool = False) -> None:
        self._p = C.c_void_p()
        dt = np.dtype(dtype)
        nbytes = max(1, int(n_items) * dt.itemsize)
        self.locked = not lazy
        rc = (lib().kp_host_reserve if lazy else lib().kp_host_alloc)(C.c_size_t(nbytes), C.byref(self._p))
        if rc != 0:
            raise NativeError(f"kp_host_alloc failed ({rc}): {lib().kp_last_error(None).decode()}")
        buf = (C.c_uint8 * nbytes).from_address(self._p.value)
        self.array = np.frombuffer(buf, dtype=dt, count=int(n_items))

    def lock(self) -> None:
        if not self.locked:
            rc = lib().kp_host_lock(self._p)
            if rc != 0:
                raise NativeError(f"kp_host_lock failed ({rc}): {lib().kp_last_error(None).decode()}")
            self.locked = True

    def close(self) -> None:
        if getattr(self, "_p", None) and self._p.value:
            self.array = None
            lib().kp_host_free(self._p)
            self._p = C.c_void_p()

    __del__ = close


_default_ctx: dict[int, Context] = {}


def default_context(device: int = 0) -> Context:
    with _lock:
        if device not in _default_ctx:
            _default_ctx[device] = Context(device)
        return _default_ctx[device]


def protein_align(device, q, q_off, q_len, t, t_off, t_len) -> np.ndarray:
    return default_context(device).protein_align(q, q_off, q_len, t, t_off, t_len)
