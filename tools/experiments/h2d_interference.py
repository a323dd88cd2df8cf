#!/usr/bin/env python3
"""Does a host-to-device copy stream slow the typing kernels down?  Resident batches typed in a loop, alone and beside a
thread that keeps a 1.25 GB pinned->device copy in flight (torch, its own stream).  GPU box only."""
import sys, time, threading
sys.path.insert(0, ".")
import numpy as np
import bench
import kaptive_amd
kaptive_amd.tune_runtime()
bench._load_dbs("kpsc")
ids, packed = bench.build_workload(3000, 200, 5.0e6, 16)
import torch
from kaptive_amd.engine import Engine
from kaptive_amd.serotyping.core import Serotyper
dbs = [bench._DBS["main"], bench._DBS["also"]]
eng = Engine(dbs)
typers = [Serotyper(d, engine=eng.view(i)) if False else None for i, d in enumerate(dbs)]
from kaptive_amd.serotyping.core import Serotyper as S
typer = S(dbs[0]); typer._engine = eng
batches = [eng.ctx.batch(packed[i * 1000:(i + 1) * 1000]) for i in range(3)]
bids = [ids[i * 1000:(i + 1) * 1000] for i in range(3)]
def one_round(n=4):
    t = time.perf_counter()
    for _ in range(n):
        eng.type_batches(typer, batches, bids)
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / (3 * n) * 1e3
one_round(1); one_round(1)
print("alone: %.2f ms per batch" % one_round())
host = torch.empty(1_250_000_000 // 4, dtype=torch.int32).pin_memory()
dev = torch.empty_like(host, device="cuda")
stop = False
def feeder(mode):
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        while not stop:
            if mode == "h2d": dev.copy_(host, non_blocking=True)
            else: host.copy_(dev, non_blocking=True)
            s.synchronize()
for mode in ("h2d", "d2h"):
    stop = False
    th = threading.Thread(target=feeder, args=(mode,)); th.start()
    time.sleep(0.2)
    print("beside a continuous %s copy: %.2f ms per batch" % (mode, one_round()))
    stop = True; th.join()
print("alone again: %.2f ms per batch" % one_round())
