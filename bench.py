#!/usr/bin/env python3
"""bench.py -- assemblies typed per second (K + O databases back to back) on N MI355X GPUs.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--assemblies A]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One "step" types every assembly of the rank's resident batch against the synthetic KpSC K-locus database and the
O-locus database: per database one alignment pass (seed scan -> anchor sort -> chaining -> banded Smith-Waterman), then
hit finalisation, locus scoring, overlap cull, pieces, translation, protein DP and gene states, all on the GPU; the
host does three small numpy float steps per database and builds the result objects.  The two databases' passes run on
their own contexts and streams (`--shared-pass`: one alignment pass over the genes of both databases, see --help).
Packed assemblies are resident in HBM before the timed region.  Ranks hold disjoint assemblies (weak scaling, no
collective on the data path; the only torch.distributed calls are the barrier and the max-over-ranks of the elapsed
time).  Rank 0 prints one JSON line.

Extra objects in the line:
  roofline      seed-scan kernel of the K database pass (the kernel that streams every base against the large
                database): algorithmic bytes = 4 * packed words per launch, duration = mean over its launches in the
                timed region, from HIP events the library records on the kernel's own stream (kp_batch_profile), peak =
                8 TB/s HBM3E; traffic = PMC bytes of the committed offline collection (profiles/scan_pmc.json).
  dp            banded Smith-Waterman kernels: DP cells per second (integer VALU work; no HBM or MFMA roofline applies).
  cpu_baseline  the CPU oracle (oracle/kp_oracle.c + the numpy reduction) typing a bounded sample of the same
                assemblies on one host core.
"""

from __future__ import annotations

import argparse
import json
import os
import sys
import time
from multiprocessing import get_context
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E spec (MI355X_MICROARCH.md)
L2_PEAK_BYTES_PER_S = 34.5e12  # aggregate L2 bandwidth (MI355X_MICROARCH.md), served in 128-byte lines


def _make_one(job):
    kind, seed, length = job
    from kaptive_amd.synth import make_assembly

    db = _DBS[kind]
    g = make_assembly(db, seed=seed, length=length, also=(_DBS["o"],))
    pa = g.packed()
    return g.id, g.contigs.ids, g.contigs.seqs, g.contigs.lengths, pa


_DBS: dict = {}


def build_workload(n_asm: int, seed0: int, length: float, workers: int):
    """Synthetic KpSC-shaped databases and assemblies (SURVEY.md section 8d, config 2 inputs), generated before any
    GPU state exists so that worker processes can be forked safely."""
    from kaptive_amd.core.genome import GenomeAssembly
    from kaptive_amd.core.seq import Sequences
    from kaptive_amd.synth import make_db

    _DBS["k"] = make_db("kpsc_k", seed=100)
    _DBS["o"] = make_db("kpsc_o", seed=101)
    jobs = [("k", seed0 + i, length) for i in range(n_asm)]
    if workers > 1 and n_asm > 4:
        pool = get_context("fork").Pool(workers)
        try:
            rows = pool.map(_make_one, jobs, chunksize=max(1, n_asm // (workers * 4)))
        finally:  # close + join, never terminate(): a SIGTERM to the workers wedges tools that hook signals (rocprofv3)
            pool.close()
            pool.join()
    else:
        rows = [_make_one(j) for j in jobs]
    genomes, packed = [], []
    for gid, ids, seqs, lengths, pa in rows:
        off = np.zeros(len(lengths), np.int32)
        if len(lengths) > 1:
            np.cumsum(lengths[:-1], out=off[1:])
        g = GenomeAssembly(gid, Sequences(ids, seqs, off, lengths))
        g._packed.append(pa)
        genomes.append(g)
        packed.append(pa)
    return _DBS["k"], _DBS["o"], genomes, packed


def pmc_traffic(args):
    """PMC figures of the scan kernel (HBM bytes and L2 requests per launch).  Counters cannot be read from inside the
    process, so these are the figures of the committed offline collection (profiles/scan_pmc.json: separate rocprofv3
    --pmc passes of this same command, gfx950 correction applied as MI355X_MICROARCH.md prescribes); they are reported
    only when the workload is the one that collection ran, otherwise null."""
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "scan_pmc.json")
    try:
        with open(path) as fh:
            pmc = json.load(fh)
    except OSError:
        return None
    if pmc["workload"] != {"assemblies": args.assemblies, "length": args.length}:
        return None
    return pmc


def cpu_baseline(dbs, genomes, budget_s: float = 20.0) -> dict:
    """Type a bounded sample with the CPU oracle (C aligner + C protein DP + numpy reduction), one core."""
    from kaptive_amd.core.pairwise import PairwiseAlignments
    from kaptive_amd.pack import pack_sequences_flat
    from kaptive_amd.serotyping.core import Serotyper
    from oracle import oracle as O
    from tests.golden_util import hits_to_alignments

    def oracle_proteins(q, t):
        return PairwiseAlignments.from_table(O.protein_align(q.seqs, q.offsets, q.lengths, t.seqs, t.offsets, t.lengths))

    stages = []
    for db in dbs:
        odb = O.OracleDB(*pack_sequences_flat(db.genes))
        stages.append((db, odb, Serotyper(db, aligner=lambda g: None, protein_aligner=oracle_proteins)))
    n = 0
    t0 = time.perf_counter()
    for g in genomes:
        for db, odb, typer in stages:
            hits = odb.align(g.packed())
            typer.reduce(g, hits_to_alignments(db, g, hits))
        n += 1
        if time.perf_counter() - t0 > budget_s:
            break
    dt = time.perf_counter() - t0
    return {
        "value": n / dt, "unit": "assemblies/s", "cores": 1, "kind": "port",
        "sample": f"first {n} assemblies of the same batch, K then O, CPU oracle (oracle/kp_oracle.c aligner + protein DP, "
                  f"numpy reduction), {dt:.1f} s wall on 1 of {os.cpu_count()} host cores",
    }  # fmt: skip


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--assemblies", type=int, default=1000, help="assemblies per GPU (resident batch)")
    ap.add_argument("--length", type=float, default=5.0e6)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--workers", type=int, default=0, help="processes for workload generation (0 = auto, 1 = inline)")
    ap.add_argument("--sub-batches", type=int, default=1,
                    help="resident device batches per database; they are software-pipelined so that host-side steps of "
                         "one overlap device work of the next")
    ap.add_argument("--shared-pass", action="store_true",
                    help="the genes of both databases share one seed index: every assembly is scanned, chained and "
                         "aligned once and each database's reduction takes its own run of the gene-sorted hit table "
                         "(default: one context and one alignment pass per database, as the reference runs them; the "
                         "shared pass does a tenth less device work, but the small database's pass and reduction "
                         "otherwise hide completely underneath the large one's alignment, so the step is not shorter)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")

    workers = args.workers or max(1, min(16, (os.cpu_count() or 1) // max(world, 1)))
    if workers == 1:
        import torch  # noqa: F401  (inline generation loads the native library; under rocprofv3 torch has to come first)
    t_gen = time.perf_counter()
    db_k, db_o, genomes, packed = build_workload(args.assemblies, 200 + rank * args.assemblies, args.length, workers)
    t_gen = time.perf_counter() - t_gen

    import torch
    import torch.distributed as dist

    from kaptive_amd.engine import Engine
    from kaptive_amd.serotyping.core import Serotyper

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: no GPU visible (kaptive_amd has no CPU path)")
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))

    def sync_all():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    from kaptive_amd.shard import shard_bounds

    n_sub = max(1, min(args.sub_batches, len(packed)))
    spans = [shard_bounds(len(packed), i, n_sub) for i in range(n_sub)]
    sub_ids = [[g.id for g in genomes[lo:hi]] for lo, hi in spans]
    # passes: (engine, resident batches) per alignment pass; stages: (engine as one database sees it, typer, batches)
    passes, stages = [], []
    if not args.shared_pass:
        for db in (db_k, db_o):
            eng = Engine(db, device=local_rank)
            batches = [eng.ctx.batch(packed[lo:hi]) for lo, hi in spans]
            passes.append((eng, batches))
            stages.append((eng, Serotyper(db, device=local_rank), batches))
    else:
        eng = Engine([db_k, db_o], device=local_rank)
        batches = [eng.ctx.batch(packed[lo:hi]) for lo, hi in spans]
        passes.append((eng, batches))
        stages = [(eng.view(i), Serotyper(db, device=local_rank), batches) for i, db in enumerate((db_k, db_o))]
    for view, typer, _ in stages:
        typer._engine = view

    def step():
        # every alignment pass is enqueued up front (contexts have their own streams); then per database: score ->
        # choice of best locus (numpy) -> reduction -> decisions as columns (BatchTyping)
        for _, batches in passes:
            for b in batches:
                b.align_async()
        # Reductions of every database are enqueued before any result is collected (typing groups have their own
        # streams), so the column-wise finishing of one database (host) runs while the device reduces the next.
        # separate passes: the smaller database first (its pass ends long before the other's); one shared pass: the
        # larger one first (its reduction is the longer chain and the other one runs beside it)
        order = sorted(stages, key=lambda st: len(st[0].db.genes), reverse=len(passes) == 1)
        if len(passes) == 1:  # all scores first (cheap), so that none queues up behind another database's reduction
            staged = [view.score_batches(typer, batches) for view, typer, batches in order]
            for (view, typer, batches), st in zip(order, staged):
                view.enqueue_reductions(typer, batches, st)
        else:  # a database's reduction is enqueued as soon as its own pass is through
            staged = [view.reduce_batches(typer, batches, aligned=True) for view, typer, batches in order]
        out = []
        done = list(zip(order, staged))
        if len(passes) == 1:
            done.reverse()  # the short chain is finished first: its columns are built while the long one still runs
        for (view, typer, batches), st in done:
            out += view.collect_batches(typer, batches, sub_ids, st)
        return out

    for _ in range(args.warmup):
        step()
    sync_all()
    prof = [[] for _ in passes]  # stage timings of every timed launch, per alignment pass
    t0 = time.perf_counter()
    for _ in range(args.steps):
        res = step()
        for plist, (_, batches) in zip(prof, passes):
            plist += [b.profile() for b in batches]  # events of the passes that just ran; no extra GPU work
    sync_all()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    if rank == 0:
        n_total = args.assemblies * world * args.steps
        stats = []
        for _, batches in passes:  # counters summed over the sub-batches of one alignment pass
            parts = [b.stats() for b in batches]
            stats.append({k: sum(p[k] for p in parts) for k in parts[0]})
        # roofline kernel: kp_scan_kernel<0, false>, the scan with the presence filter in L2 -- every launch of the first
        # alignment pass (K and O genes together, or the K database's pass) in the timed region.
        scan_all = [p["scan"] for p in prof[0]]
        scan_ms = float(np.mean(scan_all))
        scan_bytes = float(np.mean([p["bytes_scanned"] for p in prof[0]]))
        achieved = scan_bytes / (scan_ms * 1e-3) / 1e9
        pmc = pmc_traffic(args)
        l2_req = pmc["l2"]["K_l2"]["TCC_REQ_sum"] if pmc else None
        # per database: mean over the timed steps of the sum over sub-batches
        mean_ms = [{k: float(np.sum([p[k] for p in plist])) / args.steps for k in plist[0] if k != "bytes_scanned"}
                   for plist in prof]
        sw_ms = sum(m["sw16"] + m["sw32"] + m["sw64"] + m["sw128"] for m in mean_ms)
        cells = sum(s["dp_cells"] for s in stats)
        typed = int(sum(bt.typeable.sum() for bt in res))
        t_rows = time.perf_counter()
        rows = [r for bt in res for r in bt.rows()]  # TSV formatting of the last step, outside the timed region
        n_rows = len(rows)
        t_rows = time.perf_counter() - t_rows
        import hashlib

        rows_digest = hashlib.sha1(b"".join(sorted(rows))).hexdigest()  # same workload -> same digest, whatever the schedule
        line = {
            "metric": "assemblies typed/sec (K+O)",
            "value": n_total / elapsed,
            "unit": "assemblies/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "int32",
            "data": "synthetic",
            "config": {
                "workload": f"{args.assemblies} synthetic {args.length / 1e6:g} Mbp KpSC assemblies per GPU typed against "
                            "the synthetic K-locus and O-locus databases, packed batch resident in HBM; "
                            + ("one alignment pass per database" if not args.shared_pass else
                               "one alignment pass over the genes of both databases, one reduction per database"),
                "alignment_passes": len(passes),
                "assemblies_per_gpu": args.assemblies,
                "sub_batches": n_sub,
                "db_k": f"{len(db_k.loci)} loci / {len(db_k.genes)} genes",
                "db_o": f"{len(db_o.loci)} loci / {len(db_o.genes)} genes",
                "parallelism": f"{world} x independent shard, no collective",
                "typeable_in_last_step": typed,
                "tsv_rows_per_s_host": round(n_rows / t_rows, 1),
                "tsv_rows_sha1": rows_digest,
                "workload_generation_s": round(t_gen, 1),
            },
            "roofline": {
                "bound": "hbm", "kernel": "kp_scan_kernel<0, false>", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBPS, "traffic": pmc["traffic_bytes_per_launch_K_l2"] if pmc else None,
                "bytes_per_launch": scan_bytes, "ms_per_launch": scan_ms, "launches_timed": len(scan_all),
                # what actually bounds this kernel: one L2 request per presence-filter gather (PMC: TCC_REQ_sum of the
                # committed collection) against the L2's 34.5 TB/s in 128-byte lines
                "l2_requests_per_launch": l2_req,
                "l2_request_rate_frac": (l2_req / (scan_ms * 1e-3)) / (L2_PEAK_BYTES_PER_S / 128.0) if l2_req else None,
            },
            "dp": {
                "kernel": "kp_sw_kernel<8|16|32|64>", "cells_per_pass": [s["dp_cells"] for s in stats],
                "ms": sw_ms, "gcups": cells / (sw_ms * 1e-3) / 1e9 if sw_ms > 0 else None,
                "tasks": [s["tasks"] for s in stats], "anchors": [s["anchors"] for s in stats],
            },
            "kernel_ms": {k: [round(m[k], 3) for m in mean_ms] for k in ("scan", "sort", "chain", "sw16", "sw32", "sw64", "sw128")},
        }  # fmt: skip
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline((db_k, db_o), genomes)
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
