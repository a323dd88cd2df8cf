#!/usr/bin/env python3
"""Time the protein alignment kernels (GPU box) on pair shapes like the ones the typing path produces.

    python tools/protein_bench.py
"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from kaptive_amd import _native  # noqa: E402
from kaptive_amd.core.seq import Sequences  # noqa: E402


def pairs(rng, n, len_t, len_q):
    aa = np.frombuffer(b"ARNDCQEGHILKMFPSTWYV", np.uint8)
    qs, ts = [], []
    for _ in range(n):
        lt = int(rng.integers(len_t[0], len_t[1]))
        lq = min(lt, int(rng.integers(len_q[0], len_q[1])))
        t = aa[rng.integers(0, 20, size=lt)]
        q = t[:lq].copy()
        hit = rng.random(lq) < 0.1
        q[hit] = aa[rng.integers(0, 20, size=int(hit.sum()))]
        qs.append(q.tobytes())
        ts.append(t.tobytes() + b"*")
    return Sequences.from_bytes(qs), Sequences.from_bytes(ts)


def main():
    ctx = _native.Context(0)
    rng = np.random.default_rng(1)
    for name, n, lt, lq in (("full-length 20000", 20000, (200, 600), (200, 600)),
                            ("truncated 1000", 1000, (200, 600), (20, 150)),
                            ("truncated 4000", 4000, (200, 600), (20, 150)),
                            ("one truncated", 1, (599, 600), (50, 51)),
                            ("64 x (600, 450)", 64, (599, 600), (450, 451)),
                            ("64 x (600, 50)", 64, (599, 600), (50, 51)),
                            ("2000 x (600, 450)", 2000, (599, 600), (450, 451))):
        q, t = pairs(rng, n, lt, lq)
        for _ in range(2):
            t0 = time.perf_counter()
            out = ctx.protein_align(q.seqs, q.offsets, q.lengths, t.seqs, t.offsets, t.lengths)
            dt = time.perf_counter() - t0
        print(f"{name:>20}: {dt * 1e3:8.2f} ms (incl. copies), mean score {out[:, 0].mean():.1f}", flush=True)


if __name__ == "__main__":
    main()
