#!/usr/bin/env python3
"""Scan-kernel ablation on the GPU box: time kp_scan_kernel with (0) everything, (1) no filter/table reads, (2) stream
only, on the same resident batch.  KAPTIVE_AMD_SCAN_ABLATE is read once per process, so each mode runs in a child."""
import json, os, subprocess, sys

CHILD = r'''
import sys, json, numpy as np
sys.path.insert(0, ".")
from kaptive_amd.synth import make_db, make_assembly
from kaptive_amd.engine import Engine
db = make_db(sys.argv[1], seed=100 if sys.argv[1] == "kpsc_k" else 101)
dbo = make_db("kpsc_o", seed=101)
gen = [make_assembly(make_db("kpsc_k", seed=100) if i == 0 else gen_db, seed=200 + i, also=(dbo,)) for i, gen_db in enumerate([None] * 0)]
dbk = make_db("kpsc_k", seed=100)
genomes = [make_assembly(dbk, seed=200 + i, also=(dbo,)) for i in range(int(sys.argv[2]))]
eng = Engine(db)
batch = eng.ctx.batch([g.packed() for g in genomes])
try:
    batch.align()
except Exception as e:
    pass
ms = []
for _ in range(5):
    try:
        ms.append(batch.profile()["scan"])
    except Exception:
        break
print(json.dumps({"scan_ms": ms, "bytes": batch.total_words * 4}))
'''
for db in ("kpsc_k", "kpsc_o"):
    for mode in (0, 1, 2):
        env = dict(os.environ, KAPTIVE_AMD_SCAN_ABLATE=str(mode))
        out = subprocess.run([sys.executable, "-c", CHILD, db, sys.argv[1] if len(sys.argv) > 1 else "64"], env=env,
                             capture_output=True, text=True)
        line = out.stdout.strip().splitlines()[-1] if out.stdout.strip() else out.stderr[-300:]
        try:
            d = json.loads(line)
            best = min(d["scan_ms"]) if d["scan_ms"] else float("nan")
            print(f"{db} mode {mode}: best {best:.3f} ms  -> {d['bytes'] / best / 1e6:.1f} GB/s   all={['%.3f' % x for x in d['scan_ms']]}")
        except Exception:
            print(db, mode, "failed:", line)
