"""ctypes front end of oracle/mm2_model.c -- the independent model of the published minimap2 pipeline.

TEST INFRASTRUCTURE ONLY (tests/ and tools/concordance.py).  PARITY UNPINNED: see the header of mm2_model.c.
"""

from __future__ import annotations

import ctypes as C
import subprocess
from pathlib import Path

import numpy as np

from oracle.oracle import HIT_DTYPE

_HERE = Path(__file__).resolve().parent
_LIB = None


def build(force: bool = False) -> Path:
    so, src = _HERE / "libmm2_model.so", _HERE / "mm2_model.c"
    if force or not so.exists() or so.stat().st_mtime < src.stat().st_mtime:
        subprocess.run(["make", "-C", str(_HERE), "-B", "libmm2_model.so"], check=True, capture_output=True)
    return so


def lib() -> C.CDLL:
    global _LIB
    if _LIB is None:
        _LIB = C.CDLL(str(build()))
        _LIB.mm2_index_build.restype = C.c_void_p
        _LIB.mm2_map.restype = C.c_int64
        _LIB.mm2_sketch.restype = C.c_int64
        _LIB.mm2_index_n_minimizers.restype = C.c_int64
    return _LIB


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


class Mm2Index:
    """Minimizer index over one assembly's contigs (the reference indexes the assembly: core/genome.py:177-191)."""

    def __init__(self, seqs: np.ndarray, offsets, lengths):
        self._seqs = np.ascontiguousarray(seqs, np.uint8)
        self._off = np.ascontiguousarray(offsets, np.int64)
        self._len = np.ascontiguousarray(lengths, np.int32)
        self._h = C.c_void_p(lib().mm2_index_build(_p(self._seqs), _p(self._off), _p(self._len), C.c_int(len(self._len))))

    @classmethod
    def from_contigs(cls, contigs) -> "Mm2Index":
        return cls(contigs.seqs, contigs.offsets, contigs.lengths)

    @property
    def mid_occ(self) -> int:
        return int(lib().mm2_index_mid_occ(self._h))

    @property
    def n_minimizers(self) -> int:
        return int(lib().mm2_index_n_minimizers(self._h))

    def map(self, genes, score_kind: int = 0) -> np.ndarray:
        """All genes (a Sequences-like with seqs/offsets/lengths) against the index -> HIT_DTYPE rows, genes in order,
        each gene's hits in minimap2's output order."""
        g = np.ascontiguousarray(genes.seqs, np.uint8)
        off = np.ascontiguousarray(genes.offsets, np.int64)
        ln = np.ascontiguousarray(genes.lengths, np.int32)
        cap = 1 << 14
        while True:
            out = np.zeros(cap, HIT_DTYPE)
            n = int(lib().mm2_map(self._h, _p(g), _p(off), _p(ln), C.c_int(len(ln)), _p(out), C.c_int64(cap),
                                  C.c_int(score_kind)))  # fmt: skip
            if n <= cap:
                return out[:n]
            cap = n

    def __del__(self):
        if getattr(self, "_h", None) and _LIB is not None:
            _LIB.mm2_index_free(self._h)
            self._h = None


def sketch(seq: bytes) -> tuple[np.ndarray, np.ndarray]:
    s = np.frombuffer(seq, np.uint8)
    cap = len(s) + 8
    x, y = np.zeros(cap, np.uint64), np.zeros(cap, np.uint64)
    n = int(lib().mm2_sketch(_p(np.ascontiguousarray(s)), C.c_int(len(s)), _p(x), _p(y), C.c_int64(cap)))
    return x[:n], y[:n]


_CODE = np.full(256, 4, np.uint8)
for _i, _c in enumerate(b"ACGT"):
    _CODE[_c] = _i
    _CODE[_c + 32] = _i


def ksw(q: bytes, t: bytes, w: int = 751, zdrop: int = -1, ext_only: bool = False, right: bool = False):
    """(score, max, max_t, max_q, zdropped, cigar string) of the model's ksw2_extd2 on two ASCII sequences."""
    qc = np.ascontiguousarray(_CODE[np.frombuffer(q, np.uint8)])
    tc = np.ascontiguousarray(_CODE[np.frombuffer(t, np.uint8)])
    res = np.zeros(5, np.int32)
    cig = np.zeros(len(q) + len(t) + 4, np.uint32)
    n = lib().mm2_ksw(_p(qc), C.c_int(len(qc)), _p(tc), C.c_int(len(tc)), C.c_int(w), C.c_int(zdrop),
                      C.c_int(int(ext_only)), C.c_int(int(right)), _p(res), _p(cig), C.c_int(len(cig)))  # fmt: skip
    s = "".join(f"{int(c) >> 4}{'MID'[int(c) & 0xF]}" for c in cig[:n])
    return int(res[0]), int(res[1]), int(res[2]), int(res[3]), int(res[4]), s
