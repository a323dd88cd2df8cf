#!/usr/bin/env python3
"""Is the reader slow when it starts?  A fresh process parses 3 chunks of 512 files at once (6 threads each, as the command
line does), then the same again, then a third time; with and without a device context being created beside the first round.
python tools/experiments/reader_start.py"""
import os, sys, time, tempfile, shutil, subprocess
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT))

if len(sys.argv) > 1 and sys.argv[1] == "child":
    import threading
    import numpy as np
    from kaptive_amd import _native

    paths = sorted(str(p) for p in Path(sys.argv[2]).glob("asm*.fasta"))
    with_ctx = sys.argv[3] == "ctx"
    lazy = sys.argv[4] == "lazy"
    chunk = (paths * 8)[:512]
    t00 = time.perf_counter()

    def parse(out, k):
        t = time.perf_counter()
        sh = _native.FastaShard(chunk, [None] * len(chunk), 6)
        t1 = time.perf_counter()
        pb = _native.PinnedBuffer(sh.total_words + sh.total_words // 8, np.uint32, lazy=lazy)
        t2 = time.perf_counter()
        sh.words_into(pb.array, 6)
        t3 = time.perf_counter()
        sh.close()
        out[k] = (t1 - t, t2 - t1, t3 - t2, pb)

    for rnd in range(5):
        out = {}
        ths = [threading.Thread(target=parse, args=(out, k)) for k in range(3)]
        t = time.perf_counter()
        for th in ths:
            th.start()
        if with_ctx and rnd == 0:
            tc = time.perf_counter()
            ctx = _native.Context(0)
            tc = time.perf_counter() - tc
        for th in ths:
            th.join()
        dt = time.perf_counter() - t
        print(f"  round {rnd}: 1536 files in {dt:.3f} s = {1536 / dt:.0f} files/s; per chunk parse/alloc/copy: " +
              ", ".join(f"{a:.3f}/{b:.3f}/{c:.3f}" for a, b, c, _ in out.values()) + (f"; context took {tc:.3f}" if with_ctx and rnd == 0 else ""), flush=True)
        for *_, pb in out.values():
            pb.close()
    sys.exit(0)

from multiprocessing import Pool
from tools.cli_probe import one

root = Path(tempfile.mkdtemp(prefix="kp_rs_", dir="/dev/shm"))
try:
    with Pool(16) as pool:
        pool.map(one, [(i, str(root), False) for i in range(192)])
    time.sleep(0.5)
    for ctx in ("noctx",):
        for lazy in ("lazy",):
            print(f"{ctx}, {lazy} buffers:", flush=True)
            subprocess.run([sys.executable, __file__, "child", str(root), ctx, lazy], env=dict(os.environ, PYTHONPATH=str(ROOT)))
finally:
    shutil.rmtree(root, ignore_errors=True)
