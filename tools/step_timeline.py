#!/usr/bin/env python3
"""Host-side timeline of one bench step (GPU box): when each database's alignment pass ends on the device and how long
every host-visible stage of the typing takes.  Diagnostic only; prints one line per step.

    python tools/step_timeline.py [--assemblies 1000] [--steps 3]
"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--assemblies", type=int, default=1000)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--workers", type=int, default=0)
    args = ap.parse_args()
    workers = args.workers or max(1, min(16, os.cpu_count() or 1))
    db_k, db_o, genomes, packed = bench.build_workload(args.assemblies, 200, 5.0e6, workers)
    import torch

    from kaptive_amd.engine import Engine
    from kaptive_amd.serotyping import batch as B
    from kaptive_amd.serotyping.core import Serotyper

    ids = [g.id for g in genomes]
    stages = []
    for name, db in (("K", db_k), ("O", db_o)):
        eng = Engine(db, device=0)
        typer = Serotyper(db, device=0)
        typer._engine = eng
        stages.append((name, eng, typer, eng.ctx.batch(packed)))
    for it in range(args.steps + 1):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        marks = []
        for _, _, _, b in stages:
            b.align_async()
        marks.append(("enqueue", time.perf_counter() - t0))
        for name, eng, typer, b in sorted(stages, key=lambda st: len(st[1].db.genes)):
            b.wait()
            marks.append((name + ".wait", time.perf_counter() - t0))
            scores, counts = b.score(typer.min_gene_coverage)
            marks.append((name + ".score", time.perf_counter() - t0))
            best, _, _ = B.choose_best_loci(scores, counts, typer._expected_genes_per_locus)
            marks.append((name + ".choose", time.perf_counter() - t0))
            b.reduce_async(best, eng.typing_params(typer))
            sums, kept, pieces = b.typing()
            marks.append((name + ".typing", time.perf_counter() - t0))
            B.BatchTyping(typer, ids, sums, kept, pieces, scores, best)
            marks.append((name + ".columns", time.perf_counter() - t0))
        torch.cuda.synchronize()
        marks.append(("end", time.perf_counter() - t0))
        if it:
            print(" ".join(f"{k}={v * 1e3:.1f}" for k, v in marks), {n: b.profile() for n, _, _, b in stages}, flush=True)


if __name__ == "__main__":
    main()
