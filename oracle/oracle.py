"""ctypes front end of the CPU oracle (oracle/kp_oracle.c).  TEST INFRASTRUCTURE ONLY.

Only tests/, ``__graft_entry__.smoke()`` and ``bench.py``'s cpu_baseline leg may import this module; the product
package (kaptive_amd/) never does.  See kp_oracle.c for what each entry point restates and how it is pinned.
"""

from __future__ import annotations

import ctypes as C
import os
import subprocess
from pathlib import Path

import numpy as np

_HERE = Path(__file__).resolve().parent
_LIB = None

HIT_DTYPE = np.dtype(
    [("gene", "<i4"), ("contig", "<i4"), ("q_start", "<i4"), ("q_end", "<i4"), ("t_start", "<i4"), ("t_end", "<i4"),
     ("score", "<i4"), ("matches", "<i4"), ("block_len", "<i4"), ("strand", "i1"), ("mapq", "u1"), ("n_seeds", "u1"), ("pad", "u1")]
)  # fmt: skip
TASK_DTYPE = np.dtype(
    [("gs", "<i4"), ("contig", "<i4"), ("lo", "<i4"), ("width", "<i4"), ("n_anchors", "<i4"), ("qmin", "<i4"),
     ("qmax", "<i4"), ("chain_score", "<i4")]
)  # fmt: skip


JOIN_MAX_PIECES = 8
JOIN_DTYPE = np.dtype(
    [("gs", "<i4"), ("contig", "<i4"), ("n_pieces", "<i4"), ("n_anchors", "<i4"), ("chain_score", "<i4"), ("width", "<i4"),
     ("lo", "<i4", (JOIN_MAX_PIECES,)),
     # per piece: state (0 nothing, 1 hit, 2 rejected by the drop test), mask of visited pieces, then the cell score,
     # q_start, q_end, t_start, t_end, matches, block_len, the reported score of its joined path and its order-score bonus
     ("piece", "<i4", (JOIN_MAX_PIECES, 11))]
)  # fmt: skip


def build(force: bool = False) -> Path:
    so = _HERE / "libkp_oracle.so"
    src = _HERE / "kp_oracle.c"
    spec = _HERE.parent / "include" / "kp_spec.h"
    mapq = _HERE.parent / "include" / "kp_mapq.h"
    if force or not so.exists() or so.stat().st_mtime < max(src.stat().st_mtime, spec.stat().st_mtime, mapq.stat().st_mtime):
        subprocess.run(["make", "-C", str(_HERE), "-B", "libkp_oracle.so"], check=True, capture_output=True)
    return so


def build_native() -> str | None:
    """-O3 -march=native build for the box this runs on (bench.py's cpu_baseline leg; the committed recipe's portable
    build travels with the tree, a -march=native one must not).  Written to the temp dir; None without a compiler."""
    import shutil
    import tempfile

    cc = shutil.which("gcc") or shutil.which("cc")
    if not cc:
        return None
    out = Path(tempfile.gettempdir()) / f"libkp_oracle_native_{os.getpid()}.so"
    r = subprocess.run([cc, "-O3", "-march=native", "-std=c11", "-fPIC", "-shared", "-fvisibility=hidden",
                        f"-I{_HERE.parent / 'include'}", "-o", str(out), str(_HERE / "kp_oracle.c"), "-lm"], capture_output=True)
    return str(out) if r.returncode == 0 else None


def use_library(path: str) -> None:
    """Load the oracle from ``path`` instead of the committed recipe's build (before the first call into it)."""
    global _LIB
    _LIB = None
    _load(Path(path))


def _load(path: Path) -> None:
    global _LIB
    _LIB = C.CDLL(str(path))
    _LIB.kpo_db_create.restype = C.c_void_p
    _LIB.kpo_db_n_postings.restype = C.c_int64
    for f in ("kpo_anchors", "kpo_tasks", "kpo_sw", "kpo_align", "kpo_translate", "kpo_extract", "kpo_seeds", "kpo_joins"):
        getattr(_LIB, f).restype = C.c_int64


def lib() -> C.CDLL:
    global _LIB
    if _LIB is None:
        _load(build())
    return _LIB


def _p(a: np.ndarray | None):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _c(a, dt):
    return np.ascontiguousarray(a, dtype=dt)


# ---- reference-pinned kernels ----------------------------------------------------------------------------------
def blosum62() -> np.ndarray:
    out = np.empty((256, 256), np.int8)
    lib().kpo_blosum62(_p(out))
    return out


def protein_align(q, q_off, q_len, t, t_off, t_len) -> np.ndarray:
    """Returns int32 [n, 8]: score, matches, mismatches, gaps, q_start, q_end, t_start, t_end."""
    n = len(q_off)
    out = np.zeros((n, 8), np.int32)
    if n:
        q, t = _c(q, np.uint8), _c(t, np.uint8)
        lib().kpo_protein_align(_p(q), _p(_c(q_off, np.int32)), _p(_c(q_len, np.int32)), _p(t),
                                _p(_c(t_off, np.int32)), _p(_c(t_len, np.int32)), C.c_int(n), _p(out))  # fmt: skip
    return out


def protein_align_seeded(q, q_off, q_len, t, t_off, t_len, offsets, k=20) -> np.ndarray:
    """Seeded mode: band ``k`` around the diagonal ``offsets[p]`` of every pair.  Same columns as protein_align."""
    n = len(q_off)
    out = np.zeros((n, 8), np.int32)
    if n:
        q, t = _c(q, np.uint8), _c(t, np.uint8)
        lib().kpo_protein_align_seeded(_p(q), _p(_c(q_off, np.int32)), _p(_c(q_len, np.int32)), _p(t),
                                       _p(_c(t_off, np.int32)), _p(_c(t_len, np.int32)), C.c_int(n),
                                       _p(_c(offsets, np.int32)), C.c_int(int(k)), _p(out))  # fmt: skip
    return out


def cull_overlaps(order, g1, g2, starts, ends, max_frac=0.1) -> np.ndarray:
    n = len(starts)
    kept = np.zeros(n, np.uint8)
    lib().kpo_cull_overlaps(_p(_c(order, np.int32)), _p(_c(g1, np.int32)), _p(_c(g2, np.int32)),
                            _p(_c(starts, np.int32)), _p(_c(ends, np.int32)), C.c_double(max_frac), C.c_int(n),
                            _p(kept))  # fmt: skip
    return kept.astype(bool)


def cluster(starts, ends, groups, tolerance, order) -> np.ndarray:
    n = len(starts)
    ids = np.zeros(n, np.int32)
    lib().kpo_cluster(_p(_c(starts, np.int32)), _p(_c(ends, np.int32)), _p(_c(groups, np.int32)),
                      C.c_int64(int(tolerance)), _p(_c(order, np.int32)), C.c_int(n), _p(ids))  # fmt: skip
    return ids


def translate(seqs, off, length, frames, to_stop):
    n = len(off)
    seqs, off, length = _c(seqs, np.uint8), _c(off, np.int32), _c(length, np.int32)
    frames = _c(frames, np.int8)
    o_off, o_len = np.zeros(n, np.int32), np.zeros(n, np.int32)
    out = np.zeros(int(length.sum()) // 3 + n + 1, np.uint8)
    total = lib().kpo_translate(_p(seqs), _p(off), _p(length), _p(frames), C.c_int(n), C.c_int(int(to_stop)), _p(out),
                                _p(o_off), _p(o_len))  # fmt: skip
    return out[:total], o_off, o_len


def extract(seqs, off, idx, starts, ends, strands):
    n = len(idx)
    starts, ends = _c(starts, np.int32), _c(ends, np.int32)
    o_off, o_len = np.zeros(n, np.int32), np.zeros(n, np.int32)
    out = np.zeros(int(np.maximum(ends - starts, 0).sum()) + 1, np.uint8)
    total = lib().kpo_extract(_p(_c(seqs, np.uint8)), _p(_c(off, np.int32)), _p(_c(idx, np.int32)), _p(starts),
                              _p(ends), _p(_c(strands, np.int8)), C.c_int(n), _p(out), _p(o_off), _p(o_len))  # fmt: skip
    return out[:total], o_off, o_len


# ---- aligner (spec: include/kp_spec.h; parity vs the reference's rammappy stage is unpinned) ----------------------
def seeds(codes: np.ndarray):
    """Seeds of one sequence of codes (0..3, else ambiguous) in emission order: (start, strand bit, x) arrays."""
    codes = _c(codes, np.uint8)
    cap = len(codes) + 8
    start, z, x = np.zeros(cap, np.int32), np.zeros(cap, np.uint8), np.zeros(cap, np.uint32)
    n = int(lib().kpo_seeds(_p(codes), C.c_int64(len(codes)), _p(start), _p(z), _p(x), C.c_int64(cap)))
    return start[:n], z[:n], x[:n]


class OracleDB:
    """Seed index over gene codes (one byte per base, 0..4) with int32 offsets of length n_genes+1."""

    def __init__(self, gene_codes: np.ndarray, gene_off: np.ndarray) -> None:
        self.codes = _c(gene_codes, np.uint8)
        self.off = _c(gene_off, np.int32)
        self.n_genes = len(self.off) - 1
        self._h = C.c_void_p(lib().kpo_db_create(_p(self.codes), _p(self.off), C.c_int(self.n_genes)))

    def __del__(self) -> None:
        if getattr(self, "_h", None) and _LIB is not None:  # (module globals are gone at interpreter shutdown)
            _LIB.kpo_db_free(self._h)
            self._h = None

    @property
    def n_postings(self) -> int:
        return int(lib().kpo_db_n_postings(self._h))

    def _asm_args(self, pa):
        words = _c(pa.words, np.uint32)
        cs, cl = _c(pa.ctg_start, np.int32), _c(pa.ctg_len, np.int32)
        nr = _c(pa.n_runs, np.int32).reshape(-1)
        keep = (words, cs, cl, nr)
        return keep, (self._h, _p(words), C.c_int64(pa.padded_len), _p(cs), _p(cl), C.c_int(len(cs)), _p(nr),
                      C.c_int(len(nr) // 2))  # fmt: skip

    def anchors(self, pa) -> np.ndarray:
        keep, args = self._asm_args(pa)
        n = lib().kpo_anchors(*args, None, C.c_int64(0))
        out = np.zeros(n, np.uint64)
        lib().kpo_anchors(*args, _p(out), C.c_int64(n))
        return out

    def tasks(self, pa) -> np.ndarray:
        keep, args = self._asm_args(pa)
        n = lib().kpo_tasks(*args, None, C.c_int64(0))
        out = np.zeros(n, TASK_DTYPE)
        lib().kpo_tasks(*args, _p(out), C.c_int64(n))
        return out

    def sw(self, pa, tasks: np.ndarray) -> np.ndarray:
        keep, args = self._asm_args(pa)
        tasks = np.ascontiguousarray(tasks, dtype=TASK_DTYPE)
        out = np.zeros((len(tasks), 7), np.int32)
        lib().kpo_sw(*args, _p(tasks), C.c_int64(len(tasks)), _p(out))
        return out

    def joins(self, pa) -> np.ndarray:
        """Joins of the assembly (kp_spec.h, v5), one JOIN_DTYPE row each, in the order the oracle finds them."""
        keep, args = self._asm_args(pa)
        n = lib().kpo_joins(*args, None, C.c_int64(0))
        out = np.zeros(n, JOIN_DTYPE)
        lib().kpo_joins(*args, _p(out), C.c_int64(n))
        return out

    def align(self, pa, with_stats: bool = False, with_chain: bool = False):
        """Hits in emission order; ``with_chain`` adds the chain score behind every hit, with the order-score bonus of a
        joined hit (kp_spec.h) above bit 16 -- the inputs of its mapq and rank that the hit record does not keep."""
        keep, args = self._asm_args(pa)
        stats = np.zeros(3, np.int64)
        cap = 1 << 14
        while True:
            out, chain = np.zeros(cap, HIT_DTYPE), np.zeros(cap, np.int32)
            n = lib().kpo_align(*args, _p(out), C.c_int64(cap), _p(stats), _p(chain))
            if n <= cap:
                break
            cap = int(n)
        res = (out[:n],) + ((stats,) if with_stats else ()) + ((chain[:n],) if with_chain else ())
        return res if len(res) > 1 else res[0]
