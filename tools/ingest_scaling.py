#!/usr/bin/env python3
"""How the FASTA ingest scales with threads on this host: in-memory texts through kp_fasta_ingest_many, files on tmpfs
through kp_fasta_ingest_shard -- GB/s of FASTA text per thread count.  No GPU needed.

    python tools/ingest_scaling.py [--files 96] [--mbp 5] [--dir /dev/shm/kp_ingest]
"""
import argparse
import json
import os
import shutil
import time

import numpy as np

from kaptive_amd import _native, usable_cpus


def main() -> int:
    ap = argparse.ArgumentParser()
    ap.add_argument("--files", type=int, default=96)
    ap.add_argument("--mbp", type=float, default=5.0)
    ap.add_argument("--dir", default="/dev/shm/kp_ingest")
    ap.add_argument("--threads", default="1,2,4,8,16,32,64")
    args = ap.parse_args()
    rng = np.random.default_rng(3)
    os.makedirs(args.dir, exist_ok=True)
    texts, paths = [], []
    for k in range(args.files):
        n = int(args.mbp * 1e6)
        seq = np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, n)].tobytes()
        lines = []
        for c in range(60):
            part = seq[c * n // 60 : (c + 1) * n // 60]
            lines.append(b">contig_%d_%d len=%d" % (k, c, len(part)))
            lines += [part[j : j + 80] for j in range(0, len(part), 80)]
        text = b"\n".join(lines) + b"\n"
        texts.append(text)
        path = os.path.join(args.dir, f"asm{k}.fasta")
        with open(path, "wb") as f:
            f.write(text)
        paths.append(path)
    nbytes = sum(map(len, texts))
    out = {"files": args.files, "GB": round(nbytes / 1e9, 3), "usable_cpus": usable_cpus(), "cpu_count": os.cpu_count(),
           "text_path_level": _native.lib().kp_fasta_simd(-1), "in_memory_GBps": {}, "files_GBps": {}, "files_plus_copy_GBps": {}}
    dst = None
    for t in [int(x) for x in args.threads.split(",")]:
        _native.fasta_ingest_many(texts[:8], False, t)
        t0 = time.perf_counter()
        r = _native.fasta_ingest_many(texts, False, t)
        out["in_memory_GBps"][t] = round(nbytes / (time.perf_counter() - t0) / 1e9, 2)
        del r
        sh = _native.FastaShard(paths[:8], [None] * 8, t)
        sh.close()
        t0 = time.perf_counter()
        sh = _native.FastaShard(paths, [None] * len(paths), t)
        t1 = time.perf_counter()
        if dst is None:
            dst = np.empty(sh.total_words, np.uint32)
            dst[:] = 0
        sh.words_into(dst, t)
        t2 = time.perf_counter()
        sh.close()
        out["files_GBps"][t] = round(nbytes / (t1 - t0) / 1e9, 2)
        out["files_plus_copy_GBps"][t] = round(nbytes / (t2 - t0) / 1e9, 2)
    shutil.rmtree(args.dir, ignore_errors=True)
    print(json.dumps(out))
    return 0


if __name__ == "__main__":
    raise SystemExit(main())
