#!/usr/bin/env python3
"""Where the time before the first rows goes (GPU box): context creation, the first alignment pass of a small batch, the same
batch again, a larger one.  python tools/first_pass_probe.py"""
import sys, time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import kaptive_amd

kaptive_amd.tune_runtime()
t0 = time.perf_counter()
from kaptive_amd.engine import Engine
from kaptive_amd.serotyping.core import Serotyper
from kaptive_amd.synth import make_assembly, make_db

t_imp = time.perf_counter() - t0
db = make_db("kpsc_k", seed=100)
asms = [make_assembly(db, seed=700 + i) for i in range(128)]
packed = [a.packed() for a in asms]
t = time.perf_counter()
eng = Engine(db)
typer = Serotyper(db)
typer._engine = eng
t_ctx = time.perf_counter() - t
out = {"imports_s": round(t_imp, 3), "context_s": round(t_ctx, 3)}
for tag, n in (("first_64", 64), ("again_64", 64), ("first_128", 128), ("again_128", 128), ("tiny_1", 1)):
    t = time.perf_counter()
    b = eng.ctx.batch(packed[:n])
    t1 = time.perf_counter()
    b.align_async(); b.wait()
    t2 = time.perf_counter()
    bt = eng.type_batch(typer, b, [a.id for a in asms[:n]], aligned=True)
    rows = bt.tsv()
    t3 = time.perf_counter()
    out[tag] = {"batch_create_s": round(t1 - t, 3), "align_wait_s": round(t2 - t1, 3), "type_rows_s": round(t3 - t2, 3), "retries": b.stats()["retries"]}
    b.close()
print(out)
