"""ctypes view of tests/native_harness (g++ build of kaptive_amd/csrc/kp_reduce_core.h) -- test infrastructure."""

from __future__ import annotations

import ctypes as C
import subprocess
from pathlib import Path

import numpy as np

from kaptive_amd.serotyping.batch import KEPT_DTYPE, PIECE_DTYPE, SUMMARY_DTYPE

HERE = Path(__file__).resolve().parent / "native_harness"
ROOT = HERE.parent.parent
_LIB = None


class TypingDb(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("gene_locus", "gene_extra", "gene_pos", "gene_strand", "gene_len",
                                           "locus_gene_off", "locus_gene_len", "prot", "prot_off", "prot_len")] + [
        ("n_genes", C.c_int32), ("n_loci", C.c_int32)]  # fmt: skip


class TypingParams(C.Structure):
    _fields_ = [("min_gene_coverage", C.c_double), ("id_threshold", C.c_float), ("max_locus_length", C.c_int32),
                ("edge_tolerance", C.c_int32)]  # fmt: skip


def lib() -> C.CDLL:
    global _LIB
    if _LIB is None:
        so, src = HERE / "libreduce_harness.so", HERE / "reduce_harness.cpp"
        core = ROOT / "kaptive_amd" / "csrc" / "kp_reduce_core.h"
        if not so.exists() or so.stat().st_mtime < max(src.stat().st_mtime, core.stat().st_mtime):
            subprocess.run(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-Wall", "-o", str(so), str(src)], check=True)
        _LIB = C.CDLL(str(so))
    return _LIB


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


class HarnessDb:
    """Keeps the arrays alive and exposes the KpTypingDb struct."""

    def __init__(self, db):
        self.arrays = dict(
            gene_locus=np.ascontiguousarray(db.gene_locus_indices, np.uint16),
            gene_extra=np.ascontiguousarray(db.extra_genes, np.uint8),
            gene_pos=np.ascontiguousarray(db.gene_positions, np.uint16),
            gene_strand=np.ascontiguousarray(db.gene_intervals.strands, np.int8),
            gene_len=np.ascontiguousarray(db.genes.lengths, np.int32),
            locus_gene_off=np.ascontiguousarray(db.locus_gene_offsets, np.int32),
            locus_gene_len=np.ascontiguousarray(db.locus_gene_lengths, np.int32),
            prot=np.ascontiguousarray(db.translations.seqs, np.uint8),
            prot_off=np.ascontiguousarray(db.translations.offsets, np.int32),
            prot_len=np.ascontiguousarray(db.translations.lengths, np.int32),
        )
        self.struct = TypingDb(**{k: _p(v).value for k, v in self.arrays.items()}, n_genes=len(db.genes),
                               n_loci=len(db.loci))  # fmt: skip


def params(db, typer) -> TypingParams:
    return TypingParams(typer.min_gene_coverage, float(np.float32(db.metadata.id_threshold)), db.max_locus_length,
                        typer.partial_edge_tolerance)  # fmt: skip


def finalise_hits(hits: np.ndarray) -> np.ndarray:
    h = np.ascontiguousarray(hits).copy()
    n = lib().kph_finalise_hits(_p(h), C.c_int(len(h)))
    return h[:n]


def locus_scores(hits: np.ndarray, hdb: HarnessDb, min_cov: float):
    n_loci = hdb.struct.n_loci
    scores, counts = np.zeros(n_loci, np.float64), np.zeros(n_loci, np.int32)
    h = np.ascontiguousarray(hits)
    lib().kph_locus_scores(_p(h), C.c_int(len(h)), C.byref(hdb.struct), C.c_double(min_cov), _p(scores), _p(counts))
    return scores, counts


def reduce(hits, hdb, prm, best, pa, kept_cap=2048, piece_cap=64, prot_cap=1 << 20):
    h = np.ascontiguousarray(hits)
    kept, pieces = np.zeros(kept_cap, KEPT_DTYPE), np.zeros(piece_cap, PIECE_DTYPE)
    summary = np.zeros(1, SUMMARY_DTYPE)
    prot = np.zeros(prot_cap, np.uint8)
    words = np.ascontiguousarray(pa.words, np.uint32)
    cs = np.ascontiguousarray(pa.ctg_start, np.int32)
    runs = np.ascontiguousarray(pa.n_runs, np.int32).reshape(-1)
    nk = lib().kph_reduce(_p(h), C.c_int(len(h)), C.byref(hdb.struct), C.byref(prm), C.c_int(int(best)), _p(words),
                          _p(cs), _p(runs), C.c_int(len(runs) // 2), _p(kept), C.c_int(kept_cap), _p(pieces),
                          C.c_int(piece_cap), _p(summary), _p(prot), C.c_int(prot_cap))  # fmt: skip
    assert nk >= 0, f"harness overflow {summary['overflow']}"
    return kept[:nk], pieces[: int(summary["n_pieces"][0])], summary[0], prot


def states(kept, hdb, prm, ctg_len, dp8, summary=None):
    cl = np.ascontiguousarray(ctg_len, np.int32)
    dp = np.ascontiguousarray(dp8, np.int32)
    s = np.zeros(1, SUMMARY_DTYPE) if summary is None else np.array([summary], SUMMARY_DTYPE)
    lib().kph_states(_p(kept), C.c_int(len(kept)), C.byref(hdb.struct), C.byref(prm), _p(cl), _p(dp), _p(s))
    if summary is not None:
        summary["ident_sum"], summary["n_normal"] = s[0]["ident_sum"], s[0]["n_normal"]
    return kept


def np_sum_f32(a: np.ndarray) -> np.float32:
    a = np.ascontiguousarray(a, np.float32)
    lib().kph_np_sum_f32.restype = C.c_float
    return np.float32(lib().kph_np_sum_f32(_p(a), C.c_int(len(a))))
