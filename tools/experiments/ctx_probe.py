#!/usr/bin/env python3
"""Where the 0.35-0.5 s until `context_ready` go (GPU box): library load, kp_ctx_create (runtime start-up, streams, first
buffers), kp_db_load (host-side seed index + uploads), kp_db_load_typing.  python tools/experiments/ctx_probe.py"""
import sys, time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
t0 = time.perf_counter()
import numpy as np
from kaptive_amd import _native
from kaptive_amd.pack import pack_sequences_flat
from kaptive_amd.synth import make_db

t_imp = time.perf_counter() - t0
db = make_db("kpsc_k", seed=100)
out = {"imports_s": round(t_imp, 3)}
t = time.perf_counter(); _native.lib(); out["dlopen_s"] = round(time.perf_counter() - t, 3)
t = time.perf_counter(); n = _native.device_count(); out["device_count_s (runtime start-up)"] = round(time.perf_counter() - t, 3)
t = time.perf_counter(); ctx = _native.Context(0); out["kp_ctx_create_s"] = round(time.perf_counter() - t, 3)
t = time.perf_counter(); codes, off = pack_sequences_flat(db.genes); out["pack_genes_s"] = round(time.perf_counter() - t, 3)
t = time.perf_counter(); ctx.load_genes(codes, off); out["kp_db_load_s"] = round(time.perf_counter() - t, 3)
t = time.perf_counter(); ctx.load_typing(db, group=0, gene_lo=0, gene_hi=len(db.genes)); out["kp_db_load_typing_s"] = round(time.perf_counter() - t, 3)
t = time.perf_counter(); c2 = _native.Context(0); out["second kp_ctx_create_s"] = round(time.perf_counter() - t, 3)
t = time.perf_counter(); c2.load_genes(codes, off); out["second kp_db_load_s"] = round(time.perf_counter() - t, 3)
print(out)
