#!/usr/bin/env python3
"""bench.py -- assemblies typed per second (K + O databases back to back) on N MI355X GPUs.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--assemblies A] [--batch B] [--db kpsc|ab_k]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Workload (BASELINE.json config 3): A = 10 000 synthetic ~5 Mbp KpSC assemblies per GPU, typed against the synthetic
K-locus database and then the O-locus database.  One "step" is one pass of the whole hot path over all A assemblies:
they go through the context-owned work buffers as A / B batches of B = 1000, per database one alignment pass per batch
(seed scan -> anchor sort -> chaining -> banded Smith-Waterman), then hit finalisation, locus scoring, overlap cull,
pieces, translation, protein DP and gene states on the GPU; the host does three small numpy float steps per database
and batch and formats the TSV rows.  The two databases have their own contexts and streams and share one resident copy
of the packed assemblies.  `--db ab_k` runs config 4 instead (A. baumannii K database, 4 Mbp, ~1 500 contigs).

Three throughputs are reported in the one JSON line:
  value                      packed assemblies resident in HBM before the timed region (the contract's number)
  e2e.from_host_shards       every batch uploaded from pre-packed shards in pinned host memory inside the timed region
                             (H2D on a copy stream, two batches ahead of the alignment pass), then typed
  e2e.with_tsv               the same plus the TSV bytes of every row
Ranks hold disjoint assemblies (weak scaling, no collective on the data path; torch.distributed is used for the
barrier and the max-over-ranks of the elapsed time only).  Rank 0 prints the line.

Extra objects in the line:
  roofline      seed-scan kernel of the K database pass (the kernel that streams every base against the large
                database): algorithmic bytes = 4 * packed words per launch, duration = mean over its launches in the
                timed region, from HIP events the library records on the kernel's own stream (kp_batch_profile), peak =
                8 TB/s HBM3E; traffic = PMC bytes of a committed offline collection (traffic_source names it) or null.
  dp            banded Smith-Waterman kernels: DP cells per second (integer VALU work; no HBM or MFMA roofline applies).
  cpu_baseline  the CPU oracle (oracle/kp_oracle.c rebuilt -O3 -march=native on this box + the numpy reduction) typing
                a bounded sample of the same assemblies: on every host core at once (value / cores) and on one core.
"""

from __future__ import annotations

import argparse
import hashlib
import json
import os
import sys
import time
from multiprocessing import get_context
from pathlib import Path

import numpy as np

os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")  # before torch / HIP initialise: see kaptive_amd/__init__.py

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E spec (MI355X_MICROARCH.md)

WORKLOADS = {
    # config 2/3 and config 4
    "kpsc": dict(main="kpsc_k", main_seed=100, also=("kpsc_o", 101), length=5.0e6, asm_kw={}),
    "ab_k": dict(main="ab_k", main_seed=102, also=None, length=4.0e6,
                 asm_kw=dict(median_contigs=1500, min_contig=200, force_split=True)),
}  # fmt: skip

_DBS: dict = {}
_WL: dict = {}


def _load_dbs(kind: str) -> None:
    from kaptive_amd.synth import make_db

    wl = WORKLOADS[kind]
    _WL.update(wl)
    _DBS["main"] = make_db(wl["main"], seed=wl["main_seed"])
    _DBS["also"] = make_db(wl["also"][0], seed=wl["also"][1]) if wl["also"] else None


def _make_one(job):
    seed, length = job
    from kaptive_amd.synth import make_assembly

    also = (_DBS["also"],) if _DBS["also"] is not None else ()
    kw = dict(_WL["asm_kw"])
    if _WL.get("mix") == "joins":
        # a workload that JOINS (kp_spec.h, kp-align v5): every fourth assembly carries three insertions / deletions of 64-300
        # bases inside genes of its locus, every fourth a storm of six 33-300 base events anywhere in the locus copy
        if seed % 4 == 0:
            kw.update(mid_indels=((64, "del"), (150, "ins"), (300, "del")), p_is=0)
        elif seed % 4 == 1:
            kw.update(indel_storm=(6, 33, 300), p_is=0)
    g = make_assembly(_DBS["main"], seed=seed, length=length, also=also, **kw)
    return g.id, g.packed()


def build_workload(n_asm: int, seed0: int, length: float, workers: int):
    """Synthetic databases and packed assemblies, generated before any GPU state exists (forked workers)."""
    jobs = [(seed0 + i, length) for i in range(n_asm)]
    if workers > 1 and n_asm > 4:
        pool = get_context("fork").Pool(workers)
        try:
            rows = pool.map(_make_one, jobs, chunksize=max(1, min(16, n_asm // (workers * 4))))
        finally:  # close + join, never terminate(): a SIGTERM to the workers wedges tools that hook signals (rocprofv3)
            pool.close()
            pool.join()
    else:
        rows = [_make_one(j) for j in jobs]
    return [r[0] for r in rows], [r[1] for r in rows]


# ---- CPU baseline (runs before any GPU state exists; workers are forked) ------------------------------------------------
_CPU: dict = {}


def _cpu_prepare(native_so) -> None:
    """Parent side, before the fork: the oracle library and one seed index per database, so that every worker shares
    the same pages instead of building (and thrashing the caches with) a private copy."""
    from kaptive_amd.core.pairwise import PairwiseAlignments
    from kaptive_amd.pack import pack_sequences_flat
    from kaptive_amd.serotyping.core import Serotyper
    from oracle import oracle as O

    if native_so:
        O.use_library(native_so)

    def oracle_proteins(q, t):
        return PairwiseAlignments.from_table(O.protein_align(q.seqs, q.offsets, q.lengths, t.seqs, t.offsets, t.lengths))

    stages = []
    for db in (d for d in (_DBS["main"], _DBS["also"]) if d is not None):
        odb = O.OracleDB(*pack_sequences_flat(db.genes))
        stages.append((db, odb, Serotyper(db, aligner=lambda g: None, protein_aligner=oracle_proteins)))
    _CPU["stages"] = stages


def _cpu_init(barrier) -> None:
    _CPU["barrier"] = barrier
    try:  # one thread per process: the numpy reduction must not start a BLAS/OpenMP pool per worker
        from threadpoolctl import threadpool_limits

        _CPU["limit"] = threadpool_limits(1)
    except Exception:
        pass
    try:
        import torch

        torch.set_num_threads(1)
    except Exception:
        pass


def _cpu_worker(job):
    """Types `seeds` on this core.  Inputs are generated first; the clock starts when every worker has its inputs
    (barrier), so nothing but typing runs on any core while any worker is being timed."""
    seeds, length = job
    from kaptive_amd.synth import make_assembly
    from tests.golden_util import hits_to_alignments

    also = (_DBS["also"],) if _DBS["also"] is not None else ()
    genomes = [make_assembly(_DBS["main"], seed=s, length=length, also=also, **_WL["asm_kw"]) for s in seeds]
    packed = [g.packed() for g in genomes]
    barrier = _CPU.get("barrier")
    if barrier is not None:
        barrier.wait(timeout=900)  # a dead worker must not hang the bench
    t0 = time.time()
    c0 = time.process_time()
    t_align = 0.0
    for g, pa in zip(genomes, packed):
        for db, odb, typer in _CPU["stages"]:
            ta = time.perf_counter()
            hits = odb.align(pa)
            t_align += time.perf_counter() - ta
            typer.reduce(g, hits_to_alignments(db, g, hits))
    return len(genomes), t0, time.time(), time.process_time() - c0, t_align


def ingest_rate(seed0: int, length: float) -> dict:
    """FASTA text -> contig table + 2-bit words + N runs (kp_fasta_ingest, what GenomeAssembly.from_file calls): MB/s of
    plain FASTA on one host core, and on all of them at once (a thread per core; the native call releases the GIL)."""
    from concurrent.futures import ThreadPoolExecutor

    from kaptive_amd import _native
    from kaptive_amd.synth import make_assembly

    also = (_DBS["also"],) if _DBS["also"] is not None else ()
    texts = [make_assembly(_DBS["main"], seed=seed0 + i, length=length, also=also, **_WL["asm_kw"]).contigs.to_fasta()
             for i in range(4)]
    nbytes = sum(len(t) for t in texts)
    _native.fasta_ingest(texts[0])  # first call: library load
    t = time.perf_counter()
    for x in texts:
        _native.fasta_ingest(x)
    one = nbytes / (time.perf_counter() - t) / 1e6
    t = time.perf_counter()
    for x in texts:
        _native.fasta_ingest(x, keep_text=False)  # what a TSV-only run of the CLI asks for
    one_packed_only = nbytes / (time.perf_counter() - t) / 1e6
    from kaptive_amd import usable_cpus

    cores = usable_cpus()  # (what the cgroup grants: 256 threads on 16 CPUs' worth of time measure the throttling, not the parser)
    chunk = [texts[i % len(texts)] for i in range(4 * cores)]  # a chunk is parsed, used and dropped, as a reader feeding a GPU
    rounds = 6                                             # does: the buffers of one chunk serve the next (block pool)
    with ThreadPoolExecutor(max_workers=cores) as pool:  # a Python thread per file
        list(pool.map(_native.fasta_ingest, chunk))
        t = time.perf_counter()
        for _ in range(rounds):
            list(pool.map(_native.fasta_ingest, chunk))
        box_py = rounds * sum(len(x) for x in chunk) / (time.perf_counter() - t) / 1e6
    _native.fasta_ingest_many(chunk, False, cores)  # (thread start-up, first touch of the block pool)
    t = time.perf_counter()
    for _ in range(rounds):
        _native.fasta_ingest_many(chunk, False, cores)  # one call, a library thread per granted CPU (GenomeAssembly.from_files)
    box = rounds * sum(len(x) for x in chunk) / (time.perf_counter() - t) / 1e6
    best = max(box, box_py)
    return {"MBps_per_core": round(one, 1), "MBps_per_core_without_text": round(one_packed_only, 1),
            "text_path": {0: "table look-ups", 1: "AVX2 + BMI2", 2: "AVX-512 (BW, VBMI2) + BMI2"}[_native.lib().kp_fasta_simd(-1)],
            "MBps_per_box": round(best, 1),
            "MBps_per_box_python_thread_per_file": round(box_py, 1), "MBps_per_box_one_native_call": round(box, 1), "cores": cores,
            "assemblies_per_s_per_box": round(best * 1e6 / (nbytes / len(texts)), 1),
            "note": "plain FASTA bytes through kp_fasta_ingest (sequence text kept, as GenomeAssembly.from_file needs it); "
                    "outside every timed leg above"}


def shard_ingest_rate(prep: dict, rounds: int = 12) -> dict:
    """The reader of `kaptive assembly` on its own: the CLI leg's files (tmpfs) through kp_fasta_ingest_shard +
    kp_shard_words_into as the CLI drives them -- three chunks in flight, each parsed by one native call on a third of the
    granted CPUs (a chunk of files to the tables of a batch and its words in one buffer, no Python per file).  This is the
    ingest bound the CLI leg is measured against."""
    import threading
    from concurrent.futures import ThreadPoolExecutor

    import numpy as np

    from kaptive_amd import _native, usable_cpus

    paths, in_flight = prep["paths"], 3
    threads = max(1, -(-usable_cpus() // in_flight))
    local = threading.local()

    def one(_):
        sh = _native.FastaShard(paths, [None] * len(paths), threads)
        if getattr(local, "dst", None) is None or len(local.dst) < sh.total_words:
            local.dst = np.zeros(sh.total_words + sh.total_words // 8, np.uint32)
        sh.words_into(local.dst, threads)
        sh.close()

    with ThreadPoolExecutor(in_flight) as pool:
        list(pool.map(one, range(in_flight)))  # (buffers touched, block pool filled)
        t = time.perf_counter()
        list(pool.map(one, range(rounds)))
        dt = (time.perf_counter() - t) / rounds
    return {"shard_threads": threads, "shard_calls_in_flight": in_flight, "shard_MBps_per_box": round(prep["nbytes"] / dt / 1e6, 1),
            "shard_assemblies_per_s_per_box": round(len(paths) / dt, 1),
            "shard_note": "files on tmpfs -> kp_fasta_ingest_shard + kp_shard_words_into (what the CLI's TSV-only reader calls per "
                          "chunk; three chunks in flight as in the CLI), sequence text not kept"}


def _write_fasta(job):
    seed, length, path = job
    from kaptive_amd.synth import make_assembly

    also = (_DBS["also"],) if _DBS["also"] is not None else ()
    data = make_assembly(_DBS["main"], seed=seed, length=length, also=also, name=Path(path).name.split(".")[0], **_WL["asm_kw"]).contigs.to_fasta()
    Path(path).write_bytes(data)
    return len(data)


def cli_prepare(seed0: int, length: float, workers: int, n_files: int = 192) -> dict:
    """FASTA files for the command-line leg, written to tmpfs before any GPU state exists (forked workers)."""
    import tempfile

    root = Path(tempfile.mkdtemp(prefix="kaptive_amd_cli_", dir="/dev/shm" if Path("/dev/shm").is_dir() else None))
    paths = [str(root / f"cli{i:04d}.fasta") for i in range(n_files)]
    jobs = [(seed0 + 50_000 + i, length, p) for i, p in enumerate(paths)]
    if workers > 1:
        pool = get_context("fork").Pool(workers)
        try:
            sizes = pool.map(_write_fasta, jobs, chunksize=2)
        finally:
            pool.close()
            pool.join()
    else:
        sizes = [_write_fasta(j) for j in jobs]
    return {"root": root, "paths": paths, "nbytes": sum(sizes), "db_path": _DBS["main"].save(root / "db.npz")}


def cli_from_fasta(prep: dict, repeats: int = 96, batch: int = 0) -> dict:
    """`python -m kaptive_amd assembly DB FILES... -o out.tsv` as a user runs it, in a process of its own: FASTA files on
    tmpfs -> reader threads -> pinned shards -> the batched typing -> TSV bytes (kaptive_amd/cli.py::_TypingPipeline).
    The distinct 5 Mbp assemblies of `prep` are listed `repeats` times (the page cache serves them, as it would a second
    pass over a directory), so the command runs for seconds while tmpfs holds a gigabyte.  Two rates: the whole command
    (interpreter start, database load, context creation and buffer sizing included) and the steady state between the
    fifth chunk's rows and the last one's."""
    import shutil
    import subprocess

    root, paths, n_files = prep["root"], prep["paths"], len(prep["paths"])

    def one(reps: int) -> dict:
        # The process before this one -- the other run of this leg, or this process's own page-locked shards given back a moment
        # ago -- left the kernel gigabytes of locked pages and a device context to take apart, which it does after the process
        # is gone as far as its parent can see; a command started into that waits for it at its first device work (first rows
        # after 1.5 s instead of 0.6 s, every time, for the 960-file run that follows the 18 432-file one).  Not the command's.
        time.sleep(float(os.environ.get("KAPTIVE_AMD_BENCH_CLI_SETTLE_S", "2.0")))
        out, timing = root / "out.tsv", root / "timing.json"
        env = dict(os.environ, KAPTIVE_AMD_CLI_TIMING=str(timing), PYTHONPATH=str(Path(__file__).resolve().parent))
        argv = [sys.executable, "-m", "kaptive_amd", "assembly", str(prep["db_path"]), *(paths * reps), "-o", str(out), *(["--batch-size", str(batch)] if batch else [])]
        t = time.perf_counter()
        r = subprocess.run(argv, env=env, capture_output=True, text=True, timeout=900)
        wall = time.perf_counter() - t
        if r.returncode != 0:
            return {"error": (r.stderr or r.stdout)[-400:]}
        tm = json.loads(timing.read_text())
        marks = tm["rows_written_at"]
        rows = out.read_bytes().count(b"\n") - 1
        steady = None
        if len(marks) >= 8:  # (from the fifth chunk's rows on: the chunks read ahead while the first pass sized its buffers are out)
            (n0, t0), (n1, t1) = marks[4], marks[-1]
            steady = (n1 - n0) / (t1 - t0)
        elif len(marks) >= 4:
            (n0, t0), (n1, t1) = marks[1], marks[-1]
            steady = (n1 - n0) / (t1 - t0)
        return {"assemblies": n_files * reps, "rows": rows, "wall_s": round(wall, 2),
                "assemblies_per_s_whole_command": round(n_files * reps / wall, 1),
                "assemblies_per_s_steady": None if steady is None else round(steady, 1), "first_rows_after_s": round(marks[0][1], 2),
                # where the whole command's time goes: phases_s and seconds_typing count from the start of the typing; process_s
                # from the creation of the process (start-up, imports and argument parsing come before run_type; what the kernel
                # does with the exiting process -- unlocking the page-locked shards, freeing the device context -- after
                # end_of_run_type: exit_s)
                "seconds_typing": round(tm["seconds"], 2), "phases_s": tm.get("phases_s"), "process_s": tm.get("process_s"),
                "exit_s": None if not tm.get("process_s") else round(wall - tm["process_s"]["end_of_run_type"], 2)}

    try:
        big = one(repeats)
        if "error" in big:
            return big
        small = one(max(1, round(1000 / n_files)))  # the reference's everyday use: about a thousand genomes
        return {**big, "distinct_files": n_files, "fasta_MB_per_assembly": round(prep["nbytes"] / n_files / 1e6, 2),
                "batch_size": batch or "the CLI's default: 64, 128, 256, then 512", "about_1000_files": small,
                "settle_s_before_each_run": float(os.environ.get("KAPTIVE_AMD_BENCH_CLI_SETTLE_S", "2.0")),
                "deferred_teardown": "untimed: the command ends with os._exit once its outputs are closed; the kernel unpins its page-locked shards and frees "
                                     "the device context after the parent has seen it exit (exit_s is what the parent waited), and a command started right "
                                     "behind it would wait for that at its first device call -- hence the settle time before each run, which no number here includes",
                "database": "K-locus only (the CLI types one database per run, as the reference's does)",
                "note": "files on tmpfs; steady = assemblies per second between the fifth chunk's rows and the last chunk's"}
    finally:
        shutil.rmtree(root, ignore_errors=True)


_GRANTED = {"cores": None}  # what the affinity mask and the cgroup quota grant, before the spin check below may lower it


def usable_cores() -> tuple[int, str]:
    """Cores this process may actually keep busy: the affinity mask, cut down to the cgroup's CPU quota when there is one
    (a container that sees 256 CPUs but is given 16 CPUs' worth of time runs 256 busy processes at 1/16 speed each)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    why = f"{n} in the affinity mask"
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = Path(path).read_text().split()
            if path.endswith("cpu.max"):
                quota, period = txt[0], float(txt[1])
            else:
                quota, period = txt[0], float(Path("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read_text())
            if quota not in ("max", "-1") and float(quota) > 0:
                q = max(1, int(float(quota) / period))
                if q < n:
                    n, why = q, f"{why}, cgroup quota {float(quota) / period:.1f} CPUs ({path})"
            break
        except (OSError, ValueError, IndexError):
            continue
    _GRANTED["cores"] = n
    if n > 1:  # ... and to what the box really gives: n spinning processes get sum(CPU seconds) / wall cores between them
        ctx = get_context("fork")
        with ctx.Pool(n, initializer=_cpu_init, initargs=(ctx.Barrier(n),)) as pool:
            parts = pool.map(_spin, [0.4] * n, chunksize=1)
        got = sum(p[0] for p in parts) / max(max(p[2] for p in parts) - min(p[1] for p in parts), 1e-9)
        if got < 0.8 * n:
            why = f"{why}; {n} spinning processes were given {got:.1f} cores' worth of CPU time"
            n = max(1, int(round(got)))
    return n, why


def _spin(seconds: float):
    barrier = _CPU.get("barrier")
    if barrier is not None:
        barrier.wait(timeout=120)
    t0, c0 = time.time(), time.process_time()
    x = 0
    while time.time() - t0 < seconds:
        for i in range(20000):
            x += i * i
    return time.process_time() - c0, t0, time.time()


def cpu_baseline(seed0: int, length: float, per_worker: int = 3) -> dict:
    """The CPU oracle (C aligner + C protein DP + numpy reduction) typing the first assemblies of the workload against
    every database: all host cores at once (one process per core, one thread each, one shared seed index, clock started
    on a barrier once every process holds its inputs), then one core alone."""
    from oracle import oracle as O

    native = O.build_native()  # -O3 -march=native for this box; None when no compiler is here (then the portable build)
    flags = "-O3 -march=native" if native else "-O2 (prebuilt; no compiler on this box)"
    cores, cores_why = usable_cores()
    n_dbs = 2 if _DBS["also"] is not None else 1
    _cpu_prepare(native)
    ctx = get_context("fork")
    jobs = [([seed0 + w * per_worker + i for i in range(per_worker)], length) for w in range(cores)]
    pool = ctx.Pool(cores, initializer=_cpu_init, initargs=(ctx.Barrier(cores),))
    try:
        parts = pool.map(_cpu_worker, jobs, chunksize=1)  # exactly one job per process: the barrier needs all of them
    finally:
        pool.close()
        pool.join()
    n_all = sum(p[0] for p in parts)
    wall = max(p[2] for p in parts) - min(p[1] for p in parts)
    cpu_s = sum(p[3] for p in parts)
    align_s = sum(p[4] for p in parts)
    rate_all = n_all / wall  # assemblies finished by all cores together per second of the common typing window
    _cpu_init(None)
    n1, t0, t1, c1, a1 = _cpu_worker(([seed0 + i for i in range(max(per_worker * 2, 6))], length))
    rate_one = n1 / (t1 - t0)
    what = "K then O" if n_dbs == 2 else "K"
    return {
        "value": rate_all, "unit": "assemblies/s", "cores": cores, "kind": "port",
        "sample": f"{cores} single-threaded processes x {per_worker} assemblies of the same workload each ({what}), started "
                  f"together on a barrier after input generation: CPU oracle (oracle/kp_oracle.c aligner + protein DP built "
                  f"{flags}, one seed index shared by all processes, numpy reduction); {n_all} assemblies in {wall:.1f} s",
        "cores_note": cores_why,
        # (a noisy box lowers `cores`, and with it the process count of this leg: the granted figure and the rate scaled to it
        # -- an extrapolation at the measured per-core rate -- stand beside the measured one)
        "cores_granted": _GRANTED["cores"],
        "value_scaled_to_granted_cores": None if not _GRANTED["cores"] else round(rate_all * _GRANTED["cores"] / cores, 2),
        "parallel_efficiency": round(rate_all / (cores * rate_one), 3),
        "cpu_seconds_per_assembly_all_cores": round(cpu_s / n_all, 3),
        "aligner_share_all_cores": round(align_s / max(sum(p[2] - p[1] for p in parts), 1e-9), 3),
        "single_core": {"value": rate_one, "unit": "assemblies/s", "cores": 1,
                        "cpu_seconds_per_assembly": round(c1 / n1, 3), "aligner_share": round(a1 / (t1 - t0), 3),
                        "sample": f"first {n1} assemblies, {t1 - t0:.1f} s on one of {cores} host cores"},
    }  # fmt: skip


SECONDARY = (  # (key, what, arguments): short legs of the default run, each the resident-batch number of a run of its own
    ("config4_ab_k", "BASELINE.json config 4: 5 000 fragmented 4 Mbp assemblies (~1 500 contigs), A. baumannii-shaped K database",
     ["--db", "ab_k", "--assemblies", "5000"]),
    ("paralog_background", "2 000 KpSC assemblies on the non-iid background (diverged relatives, IS-like repeats, a 7-copy operon), K+O",
     ["--background", "paralog", "--assemblies", "2000"]),
    ("join_heavy", "2 000 KpSC assemblies, K+O: every fourth with three 64-300 base insertions / deletions inside genes, every fourth with "
     "a storm of six 33-300 base events in its locus (kp_join.hip's kernels have work)", ["--mix", "joins", "--assemblies", "2000"]),
)


def secondary_legs() -> dict:
    """The short secondary legs of the default run: the same script, two timed steps over resident batches, for
    config 4, the non-iid background and a workload that joins; each in a process of its own beside this one (which keeps
    its context).  What comes back per leg: value, the digest of its rows, kernel milliseconds per step."""
    import subprocess

    out = {}
    for key, what, extra in SECONDARY:
        argv = [sys.executable, str(Path(__file__).resolve()), "--gpus", "1", "--steps", "2", "--warmup", "1", "--no-cpu-baseline",
                "--no-e2e", "--no-secondary", *extra]
        t = time.perf_counter()
        try:
            r = subprocess.run(argv, capture_output=True, text=True, timeout=600)
            rows = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
            if r.returncode != 0 or not rows:
                out[key] = {"workload": what, "error": (r.stderr or r.stdout)[-300:]}
                continue
            sub = json.loads(rows[-1])
            out[key] = {"workload": what, "value": sub["value"], "unit": sub["unit"], "ms_per_step": sub["ms_per_step"], "steps": sub["steps"],
                        "tsv_rows_sha1": sub["config"]["tsv_rows_sha1"], "typeable_in_last_step": sub["config"]["typeable_in_last_step"],
                        "kernel_ms_per_step": sub["kernel_ms_per_step"], "dp_cells_per_step": sub["dp"]["cells_per_step"],
                        "roofline_alone_frac": sub["roofline"]["alone"]["frac"], "wall_s": round(time.perf_counter() - t, 1)}
        except Exception as e:  # noqa: BLE001  (a secondary leg never costs the headline its line)
            out[key] = {"workload": what, "error": repr(e)[:300]}
    return out


def rank_seed0(rank: int, assemblies_per_rank: int) -> int:
    """First assembly seed of a rank: ranks hold disjoint, consecutive runs of the one seeded assembly series (weak
    scaling: every rank types `assemblies_per_rank` of its own; rank r of any world size holds the same assemblies)."""
    return 200 + rank * assemblies_per_rank


L2_GATHER_ROOF_G = 269.0  # G independent 8-byte reads per second out of a 2 MB table (profiles/l2_gather_r2.txt)
FILL_CYCLES_PER_WAVE_STEP = 2674 / 8  # tools/isa_cost.py on kp_sw_kernel's 8-step body (profiles/fill_isa_cost_r3.txt)
FILL_CLOCK_HZ = 2.26e9  # GRBM_GUI_ACTIVE per XCD / kernel duration (profiles/r2_pmc.txt)


def offline_pmc(args) -> dict | None:
    """PMC figures of the scan kernel cannot be read from inside the process; they come from a committed offline
    collection of this same command (profiles/scan_pmc_r6.json) and are reported only for the workload it ran."""
    path = ROOT / "profiles" / "scan_pmc_r6.json"
    try:
        pmc = json.loads(path.read_text())
    except OSError:
        return None
    if pmc.get("workload") != {"db": args.db, "batch": args.batch, "length": _WL["length"]}:
        return None
    return pmc


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--assemblies", type=int, default=10000, help="assemblies per GPU (one step types all of them)")
    ap.add_argument("--assemblies-total", type=int, default=0,
                    help="assemblies of the WHOLE job, sharded evenly over the ranks (overrides --assemblies): BASELINE.json "
                         "config 5 is `--gpus 8 --assemblies-total 100000`, i.e. 12 500 per rank")
    ap.add_argument("--as-rank", type=int, default=-1,
                    help="single process only: type the assemblies rank R of a multi-rank job would hold (same seeds), so "
                         "that a rank's rows can be checked against a one-process run")
    ap.add_argument("--batch", type=int, default=1000, help="assemblies per device batch")
    ap.add_argument("--db", choices=sorted(WORKLOADS), default="kpsc")
    ap.add_argument("--length", type=float, default=0.0, help="mean assembly length (0 = the workload's own)")
    ap.add_argument("--sub-rate", type=float, default=-1.0,
                    help="substitution rate of the planted loci (default: the generator's own, uniform 0-3 %% per assembly; 0 = "
                         "loci identical to the database, the common case in real collections)")
    ap.add_argument("--background", choices=("iid", "paralog"), default="iid",
                    help="what surrounds the planted loci: uniform random sequence (the headline workload, SURVEY.md 8d) or a "
                         "background that also holds diverged relatives of database genes, IS-like repeats and an rRNA-like "
                         "operon in seven copies (kaptive_amd.synth.make_assembly): what the throughput means on real genomes")
    ap.add_argument("--mix", choices=("plain", "joins"), default="plain",
                    help="joins: every fourth assembly with three 64-300 base insertions / deletions inside genes, every fourth with a "
                         "storm of six 33-300 base events in its locus (the kernels of kp_join.hip get work)")
    ap.add_argument("--no-secondary", action="store_true",
                    help="skip the short secondary legs (config 4, the non-iid background, the join-heavy mix), each a run of this script of its own")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-e2e", action="store_true", help="skip the two end-to-end legs (host shards -> rows)")
    ap.add_argument("--upload-ahead", type=int, default=2,
                    help="end-to-end legs: uploads enqueued this many shards beyond the alignment passes in flight (the copy "
                         "engine must never wait for the driving thread, which spends most of its time waiting for results)")
    ap.add_argument("--no-cli", action="store_true", help="skip the command-line leg (FASTA files on tmpfs -> `kaptive_amd assembly` -> TSV)")
    ap.add_argument("--e2e-steps", type=int, default=3, help="steps per end-to-end leg (back to back: the pipeline fills and drains once per leg)")
    ap.add_argument("--workers", type=int, default=0, help="processes for workload generation (0 = auto, 1 = inline)")
    ap.add_argument("--dist-backend", choices=("nccl", "gloo"), default="nccl",
                    help="torch.distributed backend of the barrier and the max-over-ranks of the elapsed time (nccl = RCCL)")
    ap.add_argument("--share-gpu", action="store_true",
                    help="ranks take device LOCAL_RANK modulo the visible devices (tests of the multi-process path on a "
                         "one-GPU box; use with --dist-backend gloo, RCCL refuses two ranks on one device)")
    ap.add_argument("--ahead", type=int, default=2, choices=(1, 2),
                    help="alignment passes in flight beside the shard being reduced (a context has KP_WORK_SLOTS = 3 work sets)")
    ap.add_argument("--ctx-option", action="append", default=[], metavar="NAME=VALUE",
                    help="tuning knob of every context (kp_ctx_set_option), e.g. trace_kb_per_asm=16384")
    ap.add_argument("--separate-passes", action="store_true",
                    help="one context and one alignment pass per database, as the reference runs them (default: the genes "
                         "of both databases share one seed index, so every assembly is scanned, chained and aligned once; "
                         "each database's reduction takes its own run of the gene-sorted hit table -- identical results, "
                         "tests/test_gpu_parity.py::test_one_alignment_pass_for_two_databases)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")

    if args.assemblies_total:
        from kaptive_amd.shard import shard_bounds as _bounds

        lo_r, hi_r = _bounds(args.assemblies_total, rank, world)  # the partitioning rule of the product (kaptive_amd/shard.py)
        if hi_r - lo_r != args.assemblies_total // world:
            raise SystemExit(f"--assemblies-total {args.assemblies_total} does not divide over {world} ranks")
        args.assemblies = hi_r - lo_r
    # every rank moves to its device's share of the granted CPUs on the device's NUMA node before it allocates anything large
    # (kaptive_amd/affinity.py; a one-node box keeps its whole mask; the devices' nodes are asked in a process of its own, so no
    # HIP state exists here when the workload workers are forked).
    from kaptive_amd import affinity

    placement = affinity.place(local_rank, list(range(world))) if world > 1 and not args.share_gpu else affinity.place(local_rank)
    workers = args.workers or max(1, min(64, len(os.sched_getaffinity(0)) if world > 1 else (os.cpu_count() or 1) // max(world, 1)))
    # torch (and with it the HIP runtime it bundles) has to be loaded before libkaptive_amd.so, which the packer below
    # already needs: in the other order the process ends up with two HIP runtimes and sees no device
    import torch  # noqa: F401
    _load_dbs(args.db)
    if args.sub_rate >= 0:
        _WL["asm_kw"] = dict(_WL["asm_kw"], sub_rate=args.sub_rate)
    if args.background != "iid":
        _WL["asm_kw"] = dict(_WL["asm_kw"], background=args.background)
    _WL["mix"] = args.mix
    length = args.length or _WL["length"]
    if args.as_rank >= 0 and world > 1:
        raise SystemExit("--as-rank is for single-process runs")
    seed0 = rank_seed0(args.as_rank if args.as_rank >= 0 else rank, args.assemblies)
    t_gen = time.perf_counter()
    ids, packed = build_workload(args.assemblies, seed0, length, workers)
    t_gen = time.perf_counter() - t_gen
    cpu = ingest = cli_prep = None
    if world == 1 and rank == 0 and not args.no_cpu_baseline:
        cpu = cpu_baseline(seed0, length)
        ingest = ingest_rate(seed0, length)
    if world == 1 and rank == 0 and not args.no_e2e and not args.no_cli:
        cli_prep = cli_prepare(seed0, length, workers)
        if ingest is not None:
            ingest.update(shard_ingest_rate(cli_prep))

    import torch
    import torch.distributed as dist

    from kaptive_amd import _native
    from kaptive_amd.engine import Engine
    from kaptive_amd.serotyping.core import Serotyper
    from kaptive_amd.shard import shard_bounds

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: no GPU visible (kaptive_amd has no CPU path)")
    if args.share_gpu:
        local_rank %= torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    # torch initialises its HIP state lazily, at the first call that needs it -- which would be the synchronize() that
    # opens the timed region, with a second or so of one-off work trailing into the first timed step: do it here instead
    torch.cuda.init()
    torch.zeros(1, device="cuda")
    torch.cuda.synchronize()
    # the box's own device-copy rate (read + write), quoted beside the 8 TB/s of the data sheet
    src_t = torch.empty(1 << 30, dtype=torch.uint8, device="cuda")
    dst_t = torch.empty_like(src_t)
    dst_t.copy_(src_t)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(8):
        dst_t.copy_(src_t)
    e1.record()
    torch.cuda.synchronize()
    copy_gbps = 8 * 2 * src_t.numel() / (e0.elapsed_time(e1) * 1e-3) / 1e9
    del src_t, dst_t
    torch.cuda.empty_cache()
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.dist_backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)

    def sync_all():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    dbs = [d for d in (_DBS["main"], _DBS["also"]) if d is not None]
    n_batches = max(1, (len(packed) + args.batch - 1) // args.batch)
    spans = [shard_bounds(len(packed), i, n_batches) for i in range(n_batches)]
    batch_ids = [ids[lo:hi] for lo, hi in spans]
    shared = len(dbs) > 1 and not args.separate_passes
    if shared:  # one context, one seed index over the genes of all databases; engines[k] = the engine as database k sees it
        both = Engine(dbs, device=local_rank)
        engines = [both.view(k) for k in range(len(dbs))]
    else:
        engines = [Engine(db, device=local_rank) for db in dbs]
    for ctx in {id(e.ctx): e.ctx for e in engines}.values():
        for opt in args.ctx_option:
            name, _, value = opt.partition("=")
            ctx.set_option(name, int(value))
    typers = [Serotyper(db, device=local_rank) for db in dbs]
    for eng, typer in zip(engines, typers):
        typer._engine = eng
    n_passes = 1 if shared else len(dbs)
    # results are collected smaller database first: with separate passes its pass ends long before the other's, with a
    # shared pass its reduction is the shorter chain
    collect_order = sorted(range(len(dbs)), key=lambda k: len(dbs[k].genes))

    def make_batches(i, pinned=None):
        """Device batches of shard i, one per alignment pass; further contexts adopt the first one's device words.
        Returns one batch per database (the same object for all of them when the pass is shared)."""
        lo, hi = spans[i]
        first = engines[0].ctx.batch(packed[lo:hi], pinned_words=pinned)
        if shared:
            return [first] * len(dbs)
        return [first] + [eng.ctx.batch(packed[lo:hi], device_words=first.device_words, after=first) for eng in engines[1:]]

    def distinct(bs):
        return bs[:n_passes] if not shared else bs[:1]

    prof: list[list[dict]] = [[] for _ in range(n_passes)]
    stats: list[list[dict]] = [[] for _ in range(n_passes)]

    debug = bool(os.environ.get("BENCH_DEBUG"))

    def mark(what, t_ref=[time.perf_counter()]):
        if debug:
            now = time.perf_counter()
            print(f"[bench] {what}: +{(now - t_ref[0]) * 1e3:.1f} ms", file=sys.stderr)
            t_ref[0] = now

    from concurrent.futures import ThreadPoolExecutor

    row_pool = ThreadPoolExecutor(max_workers=1)

    def run_pass(get_batches, release=None, rows_sink=None, record=False, count=None):
        """One step (or `count` shards in a row, numbered on through the steps): every shard through every database.  The
        alignment passes of the next --ahead shards are on the device (each on its work set's own stream) while this
        shard's reductions run and its results are collected."""
        n_batches = count or len(spans)
        out = []
        mark("step begins")
        live = {}

        def enqueue(j):
            live[j] = get_batches(j)
            for b in distinct(live[j]):
                b.align_async()

        for j in range(min(args.ahead, n_batches)):
            enqueue(j)
            mark(f"align {j} enqueued")
        for i in range(n_batches):
            if i + args.ahead < n_batches:
                enqueue(i + args.ahead)
                mark(f"align {i + args.ahead} enqueued")
            bs = live[i]
            if shared:  # all scores first (cheap), so that none queues up behind another database's reduction kernels
                scored = {k: engines[k].score_batches(typers[k], [bs[k]]) for k in reversed(collect_order)}
                staged = {k: engines[k].enqueue_reductions(typers[k], [bs[k]], scored[k]) for k in reversed(collect_order)}
            else:
                staged = {k: engines[k].reduce_batches(typers[k], [bs[k]], aligned=True) for k in collect_order}
            mark(f"reductions of {i} enqueued")
            for k in collect_order:
                bt = engines[k].collect_batches(typers[k], [bs[k]], [batch_ids[i % len(spans)]], staged[k])[0]
                if rows_sink is not None:  # formatted beside the main thread (the native formatter releases the GIL)
                    rows_sink.append(row_pool.submit(bt.tsv))
                out.append(bt)
                if record and (not shared or k == collect_order[-1]):
                    prof[k if not shared else 0].append(bs[k].profile())
                    stats[k if not shared else 0].append(bs[k].stats())
            mark(f"results of {i} collected")
            if release is not None:
                release(live.pop(i))
        return out

    # ---- leg 1: resident batches (the contract's number) -------------------------------------------------------------------
    def close_all(bs):
        for b in reversed(distinct(bs)):
            b.close()

    resident = [make_batches(i) for i in range(n_batches)]
    # (untimed) buffer sizing: the first pass of a context learns its work-buffer sizes and reruns until they fit; it also
    # takes every first-call cost of the measurement itself (event timing, statistics) out of the warm-up steps
    run_pass(lambda i: resident[i], record=True)
    for _ in range(args.warmup):
        run_pass(lambda i: resident[i], record=True)
    for lst in (*prof, *stats):
        lst.clear()
    sync_all()
    t0 = time.perf_counter()
    cpu0 = time.thread_time()  # CPU seconds of this (the driving) thread: what a rank needs of a host core
    proc0 = time.process_time()  # ... and of the whole process (driving thread + the HIP runtime's helper threads)
    allocs0 = _native.device_allocations()  # a device buffer that grows inside the timed steps stalls every pass in flight
    step_ms = []
    for _ in range(args.steps):
        t_step = time.perf_counter()
        res = run_pass(lambda i: resident[i], record=True)
        step_ms.append(round((time.perf_counter() - t_step) * 1e3, 2))  # host view, no synchronisation added
    sync_all()
    elapsed = time.perf_counter() - t0
    host_busy = (time.thread_time() - cpu0) / max(elapsed, 1e-9)
    import resource

    host_rank = {"rank": rank, "numa_node": placement["numa_node"], "cpus_pinned": len(placement["cpus"] or []) if placement["applied"] else None,
                 "driving_thread_cpu_s": round(time.thread_time() - cpu0, 3),
                 "process_cpu_s": round(time.process_time() - proc0, 3), "elapsed_s": round(elapsed, 3),
                 "max_rss_MB": round(resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 1024, 1),
                 "pinned_host_MB": round(_native.pinned_bytes() / 2**20, 1),
                 "buffer_growth_reruns_in_timed_steps": [sum(x["retries"] for x in slist) for slist in stats],
                 "device_reallocations_in_timed_steps": _native.device_allocations() - allocs0}
    # (untimed) the alignment kernels of one batch with nothing else on the device: what a launch takes on its own; in
    # the timed steps the passes of consecutive batches overlap and stretch each other's kernels
    alone = []
    for bs in resident[: min(2, n_batches)]:
        for b in distinct(bs)[:1]:
            b.align_async()
            b.wait()
            alone.append(b.profile())
    sync_all()
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda" if args.dist_backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        digests = [None] * world  # every rank's row digest, so that rank 0 can report what the whole job typed
        dist.all_gather_object(digests, hashlib.sha1(b"".join(sorted(r for bt in res for r in bt.tsv().splitlines(keepends=True)))).hexdigest())
        host_ranks = [None] * world  # what every rank took of the host while the clock ran
        dist.all_gather_object(host_ranks, host_rank)
    else:
        host_ranks = [host_rank]

    # ---- legs 2 and 3: from pinned host shards, without and with TSV bytes (one GPU only) -------------------------------------
    e2e = None
    if world == 1 and not args.no_e2e:
        for bs in resident:
            close_all(bs)
        resident = []
        pins = []
        t_pin = time.perf_counter()
        for lo, hi in spans:  # pre-packed shards in page-locked memory
            pb = _native.PinnedBuffer(sum(len(pa.words) for pa in packed[lo:hi]), np.uint32)
            at = 0
            for pa in packed[lo:hi]:
                pb.array[at : at + len(pa.words)] = pa.words
                at += len(pa.words)
            pins.append(pb)
        t_pin = time.perf_counter() - t_pin

        host_create = [0.0]

        def timed(with_rows: bool):
            ahead = {}
            host_create[0] = 0.0

            total = n_batches * args.e2e_steps  # the leg is one stream of shards: step s re-sends shard i as number s * n + i

            def get(i):  # uploads run --upload-ahead shards ahead of the alignment passes: the copy engine's queue is never
                # empty while the driving thread waits for scores and records (profiles/r4_h2d_timeline.md)
                t_c = time.perf_counter()
                for j in range(i, i + args.ahead + args.upload_ahead):
                    if j < total and j not in ahead:
                        ahead[j] = make_batches(j % n_batches, pins[j % n_batches].array)
                host_create[0] += time.perf_counter() - t_c
                return ahead.pop(i)

            sink = [] if with_rows else None
            sync_all()
            t1 = time.perf_counter()  # nothing of the stream is on the device yet: the first uploads are inside the clock
            run_pass(get, release=close_all, rows_sink=sink, count=total)
            if with_rows:
                sink = [f.result() for f in sink]  # every row is in memory before the clock stops
            sync_all()
            dt = time.perf_counter() - t1
            if debug:
                print(f"[bench] e2e leg: {dt * 1e3:.1f} ms, of which {host_create[0] * 1e3:.1f} ms in batch creation calls", file=sys.stderr)
            return args.assemblies * args.e2e_steps / dt, sum(len(x) for x in sink) if with_rows else 0

        timed(False)  # warm-up of the upload path (input buffers, pinned staging)
        v_shards, _ = timed(False)
        v_tsv, tsv_bytes = timed(True)
        # the copy alone: every shard uploaded (asynchronously, from the pinned buffers) and waited for, nothing typed
        sync_all()
        t_up = time.perf_counter()
        ups = [make_batches(j, pins[j].array) for j in range(n_batches)]
        for bs in ups:
            for b in distinct(bs)[:1]:
                b.upload_wait()
        t_up = time.perf_counter() - t_up
        up_bytes = sum(int(pb.array.nbytes) for pb in pins)
        for bs in ups:
            close_all(bs)
        e2e = {"from_host_shards": v_shards, "with_tsv": v_tsv, "unit": "assemblies/s", "steps": args.e2e_steps,
               "h2d_alone": {"GB_per_s": up_bytes / t_up / 1e9, "assemblies_per_s": args.assemblies / t_up,
                             "bytes": up_bytes, "note": "upper bound of any leg that starts from host memory"},
               "tsv_bytes_per_step": tsv_bytes // max(args.e2e_steps, 1), "pinning_s": round(t_pin, 1),
               "note": "a stream of steps x shards of --batch pre-packed assemblies in pinned host memory; H2D on a copy "
                       "stream two shards ahead of the alignment passes (cold start: every upload, the first ones included, is inside the clock); "
                       "with_tsv adds the KaptiveRow bytes of every assembly and database"}  # fmt: skip
        for pb in pins:
            pb.close()
        if cli_prep is not None:
            free_b, total_b = torch.cuda.mem_get_info(local_rank)
            e2e["cli_from_fasta"] = cli_from_fasta(cli_prep)
            # (the command runs beside this process, which keeps its resident batches and work sets: what it left of the device)
            e2e["cli_from_fasta"]["device_memory_free_GB_when_started"] = round(free_b / 2**30, 1)
            e2e["cli_from_fasta"]["device_memory_total_GB"] = round(total_b / 2**30, 1)

    if rank == 0:
        n_total = args.assemblies * world * args.steps
        scan_all = [p["scan"] for p in prof[0]]
        scan_ms = float(np.mean(scan_all))
        scan_bytes = float(np.mean([p["bytes_scanned"] for p in prof[0]]))
        achieved = scan_bytes / (scan_ms * 1e-3) / 1e9
        pmc = offline_pmc(args)
        if pmc and pmc.get("tcc_req_per_launch"):
            l2_req, l2_src = float(pmc["tcc_req_per_launch"]), "TCC_REQ_sum of the offline PMC collection (all of the kernel's L2 requests)"
        else:
            l2_req, l2_src = scan_bytes * 4.0 * 2.0 / 11.0, "estimate: one probe per seed, 2 / 11 of the bases"
        scan_alone_ms = float(np.mean([p["scan"] for p in alone]))
        fill_alone_ms = float(np.mean([p["sw16"] for p in alone]))
        # per database: kernel milliseconds per step (sum over the step's batches, mean over steps)
        mean_ms = [{k: float(np.sum([p[k] for p in plist])) / args.steps for k in plist[0] if k != "bytes_scanned"}
                   for plist in prof]
        sw_ms = sum(m["sw16"] + m["sw32"] + m["sw64"] + m["sw128"] for m in mean_ms)
        per_step = [{k: sum(s[k] for s in slist) // args.steps for k in slist[0]} for slist in stats]
        cells = sum(s["dp_cells"] for s in per_step)
        typed = int(sum(bt.typeable.sum() for bt in res))
        # a wave steps 2 x 64/P tasks of 4P diagonals each: 512 cells per wave-step whatever P
        wave_steps = cells / n_batches / 512.0  # (cells: per step, all batches)
        need = wave_steps * FILL_CYCLES_PER_WAVE_STEP / (256 * 4)
        have_cycles = fill_alone_ms * 1e-3 * FILL_CLOCK_HZ
        fill_model = {"cycles_per_wave_step": FILL_CYCLES_PER_WAVE_STEP, "wave_steps_per_launch": wave_steps,
                      "ms_per_launch_alone": fill_alone_ms, "clock_hz": FILL_CLOCK_HZ,
                      "simd_cycles_needed": need, "simd_cycles_available": have_cycles, "frac": need / have_cycles}
        t_rows = time.perf_counter()
        blobs = [bt.tsv() for bt in res]  # TSV bytes of the last step
        t_rows = time.perf_counter() - t_rows
        rows = [r for blob in blobs for r in blob.splitlines(keepends=True)]
        rows_digest = hashlib.sha1(b"".join(sorted(rows))).hexdigest()  # same workload -> same digest, whatever the schedule
        what = ("typed against the synthetic K-locus database, then the O-locus database" if len(dbs) == 2
                else "typed against the synthetic A. baumannii K-locus database (~1 500 contigs per assembly)")
        line = {
            "metric": "assemblies typed/sec (K+O)" if len(dbs) == 2 else "assemblies typed/sec (K)",
            "value": n_total / elapsed,
            "unit": "assemblies/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "ms_each_step": step_ms,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "uint16x2 (exact; int32-equivalent)",
            "data": "synthetic",
            "config": {
                "workload": (f"{args.assemblies_total} assemblies sharded over {world} ranks (BASELINE.json config 5 shape): "
                             if args.assemblies_total else "")
                            + f"{args.assemblies} synthetic {length / 1e6:g} Mbp {args.db} assemblies per GPU "
                            + (f"(planted loci at {100 * args.sub_rate:g} % substitutions) " if args.sub_rate >= 0 else "")
                            + ("(background with diverged relatives of database genes, IS-like repeats and a 7-copy rRNA-like operon) "
                               if args.background == "paralog" else "")
                            + f"{what}; one step "
                            f"= all of them, as {n_batches} batches of {args.batch} through context-owned work buffers; packed "
                            "assemblies resident in HBM before the timed region; "
                            + ("one alignment pass over the genes of both databases, one reduction per database" if shared
                               else "one context and alignment pass per database, one device copy of the assemblies"),
                "alignment_passes_per_batch": n_passes,
                "assemblies_per_gpu": args.assemblies,
                "assemblies_total": args.assemblies * world,
                "batch": args.batch,
                "databases": [f"{d.metadata.keyword}: {len(d.loci)} loci / {len(d.genes)} genes" for d in dbs],
                "parallelism": f"{world} x independent shard, no collective",
                "typeable_in_last_step": typed,
                "tsv_rows_per_s_host": round(len(rows) / max(t_rows, 1e-9), 1),
                "tsv_rows_sha1": rows_digest,
                "tsv_rows_sha1_per_rank": digests if world > 1 else None,
                "host_thread_busy_frac": round(host_busy, 3),
                "host_per_rank": host_ranks,
                "host_cores": len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else os.cpu_count(),
                "buffer_growth_reruns_in_timed_steps": [sum(s["retries"] for s in slist) for slist in stats],
                "workload_generation_s": round(t_gen, 1),
            },
            "e2e": e2e,
            "roofline": {
                "bound": "hbm", "kernel": "kp_scan_dense_kernel (" + ("pass over K and O genes" if shared else "K database pass") + ")", "achieved": achieved, "peak": HBM_PEAK_GBPS,
                "unit": "GB/s", "frac": achieved / HBM_PEAK_GBPS,
                "device_copy_GBps_measured": round(copy_gbps, 1),  # torch copy of 1 GiB, read + write bytes
                "traffic": pmc["traffic_bytes_per_launch"] if pmc else None,
                "traffic_source": f"offline: {pmc['source']}" if pmc else None,
                "bytes_per_launch": scan_bytes, "ms_per_launch": scan_ms, "launches_timed": len(scan_all),
                "note": "timed-region launches share the device with the other kernels of up to --ahead overlapping "
                        "passes; `alone` = the same launch with nothing else running (untimed, after the steps)",
                "alone": {"ms_per_launch": scan_alone_ms, "achieved": scan_bytes / (scan_alone_ms * 1e-3) / 1e9,
                          "frac": scan_bytes / (scan_alone_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS},
                # one 8-byte presence-filter probe per seed (minimizers: 2 / 11 of the bases; round 3's rule selected a
                # quarter): that request rate against the rate at which the L2s serve independent requests
                # (tools/microbench/l2_gather.hip).  Since round 4 the kernel is bound by vector issue first.
                "l2_requests": {"per_launch": l2_req, "per_launch_source": l2_src, "rate_alone_G_per_s": l2_req / (scan_alone_ms * 1e-3) / 1e9,
                                "measured_roof_G_per_s": L2_GATHER_ROOF_G, "frac_alone": l2_req / (scan_alone_ms * 1e-3) / 1e9 / L2_GATHER_ROOF_G,
                                "roof_source": "profiles/l2_gather_r2.txt (2 MB table, 8 loads in flight per lane)"},
            },
            "dp": {
                "kernel": "kp_sw_kernel", "cells_per_step": [s["dp_cells"] for s in per_step],
                "ms_per_step": sw_ms, "gcups": cells / (sw_ms * 1e-3) / 1e9 if sw_ms > 0 else None,
                "tasks_per_step": [s["tasks"] for s in per_step], "anchors_per_step": [s["anchors"] for s in per_step],
                # band tasks that end below the -s 80 cut (their direction bits were written for nothing): tasks without a hit.
                # (hits are counted after same-span duplicates were dropped, so this is an upper bound.)  Were it large, a
                # score-only first fill would pay; at a few per cent it does not.
                "hits_per_step": [s["hits"] for s in per_step],
                "tasks_without_hit_frac": [round(1.0 - s["hits"] / max(s["tasks"], 1), 4) for s in per_step],
                # fill kernel against its own roof, VALU issue: cycles the instruction mix of its 8-step body needs (tools/
                # isa_cost.py with the per-opcode costs measured by tools/microbench/valu_rate*.hip) over the SIMD cycles
                # the launch had, at the shader clock the PMC run saw (GRBM_GUI_ACTIVE / duration, profiles/)
                "fill_issue_model": fill_model,
            },
            "kernel_ms_per_step": {k: [round(m[k], 3) for m in mean_ms] for k in ("scan", "sort", "chain", "sw16", "sw32")},
        }  # fmt: skip
        line["kernel_ms_per_step"]["join_fill_and_walk"] = [round(m["sw64"], 3) for m in mean_ms]  # (kp_join.hip; its chaining kernel is part of "chain")
        if cpu is not None:
            line["cpu_baseline"] = cpu
            line["ingest"] = ingest
        if world == 1 and not args.no_secondary and not args.no_e2e:
            line["extra"] = secondary_legs()
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
