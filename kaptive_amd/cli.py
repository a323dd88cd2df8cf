"""Command line: ``python -m kaptive_amd type|assembly DATABASE GENOMES... -o out.tsv`` and ``convert``.

Same positionals and flags as the reference's ``kaptive type`` (alias ``assembly``) and ``kaptive convert``
(src/kaptive/serotyping/cli.py:118-267, shared output flags src/kaptive/cli.py:424-504), minus the plot output.  What
differs underneath: genomes are read and packed by a thread pool (``--threads`` is honoured; the reference parses and
ignores it) while earlier ones are being typed, in batches, on the GPU(s) (``--devices``: one process per
device; ``--batch-size``), and written in input order (``_TypingPipeline``).
DATABASE is a ``.npz`` blob written by ``Database.save`` or a GenBank file with its ``.toml`` next to it.
"""

from __future__ import annotations

import argparse
import json
import os
import sys
import time
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

import numpy as np

FILE_SUFFIX = "kaptive_results"
_KEEP: list = []  # FAST_EXIT: what must not be finalised one object at a time on the way out
FAST_EXIT = False  # set by __main__: the process ends right after main() returns, so nothing needs to be torn down in order
FAST_EXIT_ARMED = False  # set by run_type once its output handles are closed: only then may __main__ skip the interpreter's teardown


def result_to_json(result) -> bytes:
    """One JSON line per result, laid out as the reference's ``orjson.dumps(result.to_dict(), OPT_SERIALIZE_NUMPY |
    OPT_APPEND_NEWLINE)`` lays it out (src/kaptive/serotyping/cli.py:67-76): ``serotyping/jsonl.py``."""
    from kaptive_amd.serotyping.jsonl import dumps_line

    return dumps_line(result.to_dict())


class ResultExporter:
    """Fan-out of one result to every requested output (reference: src/kaptive/serotyping/cli.py:20-114)."""

    def __init__(self, args: argparse.Namespace) -> None:
        from kaptive_amd.serotyping.io import KaptiveRow, Pha4geRow

        self.handles, self.writers = [], []

        def stream(path):
            h = sys.stdout.buffer if str(path) in ("-", "stdout") else open(path, "wb")
            self.handles.append(h)
            return h

        if tsv := (getattr(args, "out", None) or getattr(args, "tsv", None)):
            h = stream(tsv)
            h.write(KaptiveRow.header())
            self.writers.append(lambda r, h=h: h.write(bytes(KaptiveRow.from_result(r))))
        if p := getattr(args, "pha4ge", None):
            h = stream(p)
            h.write(Pha4geRow.header())
            self.writers.append(lambda r, h=h: h.write(bytes(Pha4geRow.from_result(r))))
        if j := getattr(args, "json", None):
            h = stream(j)
            self.writers.append(lambda r, h=h: h.write(result_to_json(r)))
        for flag, attr, ext in (("loci", "locus_seqs", "fna"), ("genes", "gene_seqs", "ffn"), ("proteins", "translations", "faa")):
            if d := getattr(args, flag, None):
                d = Path(d)
                d.mkdir(parents=True, exist_ok=True)
                self.writers.append(
                    lambda r, d=d, attr=attr, ext=ext: (d / f"{r.genome}_{FILE_SUFFIX}.{ext}").write_bytes(
                        getattr(r, attr).to_fasta()
                    )
                )

    def __call__(self, result) -> None:
        for w in self.writers:
            w(result)

    def close(self) -> None:
        for h in self.handles:
            if h is not sys.stdout.buffer:
                h.close()
            else:
                h.flush()


def _add_outputs(p: argparse.ArgumentParser, tsv_flags, include_json: bool) -> None:
    """Defaults, ``nargs`` and ``const`` follow the reference (src/kaptive/cli.py:424-504): ``-o`` defaults to stdout,
    ``convert -t`` without a value means stdout, ``-l/-g/-p`` without a value mean the current directory."""
    g = p.add_argument_group("Outputs")
    if tsv_flags[0] == "-o":
        g.add_argument(*tsv_flags, metavar="FILE", default="stdout",
                       help="Write serotyping results as a TSV report to a file (default: %(default)s)")
    else:
        g.add_argument(*tsv_flags, metavar="FILE", nargs="?", const="stdout",
                       help="Write serotyping results as a TSV report to a file (default: %(const)s)")
    g.add_argument("-l", "--loci", metavar="DIR", nargs="?", const="./", type=Path,
                   help="Write locus nucleotide fasta files to a directory (default: %(const)s)")
    g.add_argument("-g", "--genes", metavar="DIR", nargs="?", const="./", type=Path,
                   help="Write gene nucleotide fasta files to a directory (default: %(const)s)")
    g.add_argument("-p", "--proteins", metavar="DIR", nargs="?", const="./", type=Path,
                   help="Write translation amino-acid fasta files to a directory (default: %(const)s)")
    if include_json:
        g.add_argument("-j", "--json", metavar="FILE", nargs="?", const="kaptive_results.jsonl",
                       help="Write serialised results to a newline-delimited JSON (default: %(const)s)")
    g.add_argument("--pha4ge", metavar="FILE", nargs="?", const="kaptive_results.pha4ge", type=Path,
                   help="Write PHA4GE-compliant serotyping report to a TSV file (default: %(const)s)")


def build_parser() -> argparse.ArgumentParser:
    ap = argparse.ArgumentParser(prog="kaptive_amd", description="MI355X-native locus typing (Kaptive-compatible)")
    sub = ap.add_subparsers(dest="command", required=True)
    t = sub.add_parser("type", aliases=["assembly"], help="In silico serotyping of assemblies")
    t.add_argument("database", help="Database blob (.npz) or GenBank file with its .toml")
    t.add_argument("genomes", nargs="+", help="Genome assemblies in fasta format; can be compressed")
    _add_outputs(t, ("-o", "--out"), include_json=True)
    c = t.add_argument_group("Confidence options")
    c.add_argument("--max-other-genes", type=int, default=1, metavar="", help="Typeable if <= other genes (default: 1)")
    c.add_argument("--min-completeness", type=float, default=0.5, metavar="", help="Typeable if >= completeness (default: 0.5)")
    c.add_argument("--below-threshold", action="store_true", help="Typeable if any genes in locus are below threshold")
    o = t.add_argument_group("Other options")
    o.add_argument("-t", "--threads", type=int, default=0, metavar="", help="Threads for reading/packing genomes (0 = the CPUs this process is granted)")
    o.add_argument("--partial-edge-tolerance", type=int, default=5, metavar="", help="Bases from contig edge to call a partial gene")
    o.add_argument("--devices", default="0", metavar="", help="Comma-separated GPU indices, or 'all' (default: 0)")
    o.add_argument("--batch-size", type=int, default=0, metavar="",
                   help="Assemblies per device submission (default: 64, 128, 256, then 512 -- first rows early, large batches later)")
    o.add_argument("-V", "--verbose", action="store_true")
    t.set_defaults(func=run_type)
    v = sub.add_parser("convert", help="Convert JSON-lines results to other formats")
    v.add_argument("jsonl", help="Serialised results in JSON-lines format (- for stdin)")
    _add_outputs(v, ("-t", "--tsv"), include_json=False)
    v.set_defaults(func=run_convert)
    return ap


def load_database(path: str):
    from kaptive_amd.db import Database

    p = Path(path)
    if p.suffix in (".gbk", ".gb", ".genbank"):
        from kaptive_amd.db.genbank import database_from_genbank

        return database_from_genbank(p)
    return Database.load(p)


class _TypingPipeline:
    """One device's share of ``kaptive assembly``: files -> reader pool -> pinned shard -> ``Engine.type_stream`` -> bytes.

    The stages run beside each other: while the GPU types chunk k, the reader threads parse and pack chunk k + 1 .. k + 2
    (``kp_fasta_ingest`` releases the interpreter lock), their packed words are copied into page-locked memory and
    uploaded on the copy stream, and chunk k - 1's rows are formatted.  When only the TSV and / or PHA4GE reports are asked for,
    no per-assembly object is ever built: the rows come from ``BatchTyping.tsv()`` (``kp_format_rows``, byte for byte what
    ``KaptiveRow.from_result`` gives) and ``BatchTyping.pha4ge()``, and the files' sequence text is not even kept.  The
    JSON lines (``-j``) come from ``BatchTyping.jsonl()`` (``kp_format_json``) and the records of the per-assembly fasta files
    (``-l``, ``-g``, ``-p``) from ``BatchTyping.fasta()`` (``kp_format_fasta``): they need the files' text but no objects either
    (the reference's writers work on one ``SerotypingResult`` at a time, src/kaptive/serotyping/cli.py:20-114).  The readers
    start with the process: their buffers are plain huge-page memory (``kp_host_reserve``) until a batch is made of them."""

    PREFETCH = 2  # chunks being read / uploaded ahead of the one whose alignment pass is enqueued next

    def __init__(self, args: argparse.Namespace, device: int, typer=None, chunks=None) -> None:
        """``typer``: a ``Serotyper`` the caller already holds (``Serotyper.tsv_from_files``); otherwise one is made from the
        command line's database and thresholds.  ``chunks``: what ``run`` will be given -- the first of them are handed to
        the reader threads before the database is loaded and the device context created (0.3 s in which nothing else
        would read a file)."""
        import threading

        from kaptive_amd import usable_cpus
        from kaptive_amd.serotyping.core import Serotyper

        self.args = args
        self.marks = {"pipeline_start": time.perf_counter()}  # (KAPTIVE_AMD_CLI_TIMING: where the time before the first rows goes)
        self.objects = any(getattr(args, f, None) for f in ("json", "loci", "genes", "proteins"))  # the files' text is kept
        self.fasta_outputs = any(getattr(args, f, None) for f in ("loci", "genes", "proteins"))  # ... and result objects are built
        self.threads = max(1, args.threads or usable_cpus())  # (the cgroup's quota, not the 256 CPUs a container may see)
        # PREFETCH + 1 chunks are being parsed at any time, each by one native call: the thread budget is shared out among them
        # (measured on the 16-CPU box, threads per call 4 / 6 / 8 / 12 / 16 / 24 / 32: 12.5 / 15.2 / 12.1 / 11.0 / 11.0 / 7.4 / 6.5 k
        # assemblies/s -- a cgroup throttles what oversubscribes its quota)
        self.shard_threads = int(os.environ.get("KAPTIVE_AMD_SHARD_THREADS", 0)) or max(1, -(-self.threads // (self.PREFETCH + 1)))
        self.readers = ThreadPoolExecutor(max_workers=self.threads)
        # whole chunks (TSV-only runs) are parsed PREFETCH + 1 at a time, each by shard_threads native threads, in input order
        self.shard_readers = ThreadPoolExecutor(max_workers=self.PREFETCH + 1)
        self.formatters = ThreadPoolExecutor(max_workers=2)  # a chunk's rows and JSON lines, beside the driving thread
        self.copiers = ThreadPoolExecutor(max_workers=min(4, self.threads))  # object mode: copies into pinned memory (not queued behind reads)
        self.janitor = ThreadPoolExecutor(max_workers=1)  # gives page-locked buffers back to the system while the run goes on
        self._pins: list = []  # recycled page-locked buffers
        self._pin_lock = threading.Lock()
        self._pin_need = 0  # the largest buffer asked for so far: smaller ones are not worth keeping
        self._unread = 0  # chunks that have not been handed to the readers yet
        self._early: list = []
        self._order: list = []
        if chunks is not None:
            chunks = list(chunks)
            # The readers start before the database is loaded and the device context exists (0.4-1 s), and they keep going for
            # as long as the read-ahead budget lasts: the words land in plain huge-page memory (kp_host_reserve: no device
            # runtime involved) that is page-locked when its batch is created.  A run is then as long as reading its files
            # takes, plus what the last chunk needs on the device.
            budget, ahead, sizes = _read_ahead_bytes() // max(1, getattr(args, "read_ahead_share", 1)), 0, {}
            for n, (k, paths) in enumerate(chunks):
                if n >= self.PREFETCH + 1:
                    if self.objects or any(str(p).endswith((".gz", ".bz2", ".xz")) for p in paths):
                        break  # (texts are kept / sizes unknown: no deeper than the steady state reads ahead)
                    try:
                        for p in paths:
                            if p not in sizes:
                                sizes[p] = os.stat(p).st_size
                    except OSError:
                        break
                    ahead += int(sum(sizes[p] for p in paths) * 0.3)  # 2 bits per base and the tables, with head-room
                    if ahead > budget:
                        break
                self._early.append((k, self.submit_read(paths)))
            self._unread = len(chunks) - len(self._early)
        self._own_typer = typer is None
        early_ctx = None
        if typer is None:
            # the HIP runtime and the device context come up on a thread of their own (0.2 s, interpreter lock released) while this
            # one reads and unpacks the database file (0.12 s): the context is waiting when the Serotyper asks for it
            from kaptive_amd import _native

            self._ctx_pool = ThreadPoolExecutor(max_workers=1)
            early_ctx = self._ctx_pool.submit(_native.Context, device)
        try:
            if typer is None:
                self.db = load_database(args.database)
                self.marks["database_loaded"] = time.perf_counter()
                typer = Serotyper(self.db, max_other_genes=args.max_other_genes, min_completeness=args.min_completeness,
                                  allow_below_threshold=args.below_threshold, partial_edge_tolerance=args.partial_edge_tolerance,
                                  device=device)  # fmt: skip
                typer._ctx_early = early_ctx
            self.typer = typer
            self.engine = self.typer.engine  # the context is created here, on the thread that will drive it
        except BaseException:
            # a bad database path, no device, no memory: the reads queued above must not be drained (gigabytes of FASTA) by
            # the interpreter's exit hook before the error is reported
            self._abandon_reads()
            if early_ctx is not None:
                self._ctx_pool.shutdown(wait=False, cancel_futures=True)
            raise
        self.marks["context_ready"] = time.perf_counter()
        self.want_tsv = bool(getattr(args, "out", None))

    def _abandon_reads(self) -> None:
        """Cancel every queued read, drop what finished reads hold (shards' page-locked blocks), keep no worker waiting."""
        for pool in (self.readers, self.shard_readers, self.copiers, self.formatters, self.janitor):
            pool.shutdown(wait=False, cancel_futures=True)
        for _, futures in self._early:
            for f in futures if isinstance(futures, list) else [futures]:
                if f.done() and not f.cancelled() and f.exception() is None:
                    got = f.result()
                    if isinstance(got, tuple) and isinstance(got[0], tuple):  # ((tables, words, pinned buffer), ids)
                        try:
                            got[0][2].close()
                        except Exception:
                            pass
        self._early = []
        for pb in self._pins:
            try:
                pb.close()
            except Exception:
                pass
        self._pins = []

    def release_pin(self, batch) -> None:
        """The batch's page-locked words back to the pool, once its upload has completed (waits for it if need be)."""
        pb = getattr(batch, "_pin", None)
        if pb is None:
            return
        batch._pin = None
        try:
            batch.upload_wait()
        finally:
            self._give_back(pb)

    def _give_back(self, pb) -> None:
        """A buffer whose upload is through: kept for one of the chunks still to be read (if it is large enough for
        them), otherwise unlocked and unmapped now, beside the run, rather than by the exiting process (30 ms per 0.8 GB
        either way, but the exit is on the command's clock)."""
        with self._pin_lock:
            if len(pb.array) >= self._pin_need and len(self._pins) < min(self.PREFETCH + 2, self._unread + self.PREFETCH + 1):
                self._pins.append(pb)
                return
        try:
            self.janitor.submit(pb.close)
        except RuntimeError:  # (shut down already: the process is ending)
            pass

    def submit_read(self, paths):
        """One chunk to the reader threads: a future of ``_load_shard`` (TSV / PHA4GE only) or a list of ``_load`` futures."""
        if self.objects:
            return [self.readers.submit(self._load, p) for p in paths]
        return self.shard_readers.submit(self._load_shard, paths)

    def close(self, fast: bool = False) -> None:
        """``fast``: the process is about to exit (the command line): reads are cancelled, nothing is waited for or freed --
        the operating system takes the page-locked memory and the device context back faster than the runtime unwinds them."""
        for pool in (self.readers, self.shard_readers, self.copiers, self.formatters, self.janitor, *([self._ctx_pool] if getattr(self, "_ctx_pool", None) else [])):
            pool.shutdown(wait=not fast, cancel_futures=True)
        if fast:
            return
        if self._own_typer and self.typer._engine is not None:
            self.typer._engine.close()
        for pb in self._pins:
            pb.close()

    # -- stage 1: files -> packed assemblies (reader threads) --------------------------------------------------------------------
    def _load(self, path):
        from kaptive_amd.core.genome import GenomeAssembly

        g = GenomeAssembly.from_file(path, keep_text=self.objects)
        g.packed()
        return g

    def _load_shard(self, paths):
        """A whole chunk in one native call (TSV-only runs: nothing but the packed form and the assembly names is needed):
        the library's threads map, parse and pack the files and lay out the batch's tables (kp_fasta_ingest_shard); the
        interpreter does nothing per file but derive the assembly's name from the path.  A chunk with a file the library
        could not take goes the per-file way, which knows the fall-backs and raises the reference's errors."""
        from kaptive_amd import _native
        from kaptive_amd.core.genome import _FASTA_NAME

        ids, comps = [], []
        for path in paths:
            name = os.path.basename(os.fspath(path))
            m = _FASTA_NAME.search(name)
            if not m:
                raise NotImplementedError(f"Unsupported format: {path}")
            ids.append(name.removesuffix(m.group()))
            comps.append(m.group("compression"))
        shard = _native.FastaShard(paths, comps, self.shard_threads)
        if shard.failed:
            shard.close()
            return [self._load(path) for path in paths]
        # the words go to page-locked memory here, on the reader's side of the pipeline (the driving thread only creates the
        # batch): 0.6 GB per chunk of 512 assemblies, copied by the library's threads
        pb = self._pinned(shard.total_words)
        shard.words_into(pb.array, self.shard_threads)
        tables = tuple(np.array(t) for t in shard.tables())
        total = shard.total_words
        shard.close()
        return (tables, total, pb), ids

    # -- stage 2: one chunk's packed words -> page-locked memory -> device (asynchronous upload) ---------------------------------
    def _pinned(self, n_words: int):
        from kaptive_amd import _native

        with self._pin_lock:  # (reader threads take buffers, the driving thread gives them back)
            self._pin_need = max(self._pin_need, n_words)
            for i, pb in enumerate(self._pins):
                if len(pb.array) >= n_words:
                    return self._pins.pop(i)
        return _native.PinnedBuffer(n_words + n_words // 8, np.uint32, lazy=True)  # page-locked in _make_batch, once it is full

    def _locked(self, pb, n_words: int):
        """``pb`` page-locked -- or, on a host that refuses to register it, a block of the runtime's own with the same words."""
        from kaptive_amd import _native

        try:
            pb.lock()
            return pb
        except _native.NativeError:
            eager = _native.PinnedBuffer(len(pb.array), np.uint32)
            eager.array[:n_words] = pb.array[:n_words]
            pb.close()
            return eager

    def _make_batch(self, genomes):
        if isinstance(genomes, tuple):  # ((tables, words, pinned buffer), ids) of _load_shard
            tables, total, pb = genomes[0]
            pb = self._locked(pb, total)
            batch = self.engine.ctx.batch(None, pinned_words=pb.array[:total], tables=tables)
            batch._pin = pb
            return batch
        packed = [g.packed() for g in genomes]
        sizes = [len(pa.words) for pa in packed]
        offs = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
        pb = self._pinned(int(offs[-1]))

        def copy(i):
            pb.array[offs[i] : offs[i + 1]] = packed[i].words

        list(self.copiers.map(copy, range(len(packed))))  # (numpy copies of this size run without the interpreter lock; an executor of their own: the readers' queue holds the next chunks' files)
        pb = self._locked(pb, int(offs[-1]))
        batch = self.engine.ctx.batch(packed, pinned_words=pb.array[: int(offs[-1])])
        batch._pin = pb
        return batch

    def _source(self, chunks):
        """(batch, ids, genomes) per chunk, as an iterator with ``ready()`` (``Engine.type_stream`` asks before it would
        wait): the files of the next PREFETCH chunks are being read while a chunk is handed on; their batches are created
        (uploads enqueued) as soon as the reads are through."""
        return _ChunkSource(self, chunks)

    # -- stage 3: records -> bytes ------------------------------------------------------------------------------------------------
    def run(self, chunks):
        """Yields ``(k, outputs)`` for every ``(k, paths)`` of ``chunks``, in order; ``outputs`` maps "tsv" / "pha4ge" /
        "json" to the bytes this chunk adds to that stream (per-assembly fasta files are written here)."""
        from collections import deque

        args = self.args
        self._order = []

        def render(bt) -> dict:
            out = {}
            if self.want_tsv:
                out["tsv"] = bt.tsv()
            if getattr(args, "pha4ge", None):
                out["pha4ge"] = bt.pha4ge()
            if getattr(args, "json", None):  # one native call per batch (kp_format_json): byte for byte result_to_json of every result
                out["json"] = bt.jsonl()
            if self.fasta_outputs:  # a file per assembly and kind, as the reference writes them; the records come per batch (kp_format_fasta)
                wanted = {flag: (Path(d), ext) for flag, ext in (("loci", "fna"), ("genes", "ffn"), ("proteins", "faa")) if (d := getattr(args, flag, None))}
                for flag, per_asm in bt.fasta(tuple(wanted)).items():
                    d, ext = wanted[flag]
                    d.mkdir(parents=True, exist_ok=True)
                    for genome, blob in zip(bt.ids, per_asm):
                        (d / f"{genome}_{FILE_SUFFIX}.{ext}").write_bytes(blob)
            return out

        # The rows of a chunk are rendered beside the driving thread (the native formatters release the interpreter lock: 35 MB
        # of JSON per 512 assemblies would otherwise stand between two submissions to the device); they leave in input order.
        rendering: deque = deque()
        done = 0
        for bt, batch in self.engine.type_stream(self.typer, self._source(chunks)):
            pb, batch._pin = getattr(batch, "_pin", None), None
            batch.close()  # (waits for whatever of the batch is still in flight: the pinned words are free after it)
            if pb is not None:
                self._give_back(pb)
            rendering.append(self.formatters.submit(render, bt))
            while rendering and (rendering[0].done() or len(rendering) > 2):
                yield self._order[done], rendering.popleft().result()
                done += 1
        while rendering:
            yield self._order[done], rendering.popleft().result()
            done += 1


class _ChunkSource:
    """Reads ahead of the device: PREFETCH + 1 chunks are with the reader threads at any time (started by
    ``_TypingPipeline.__init__`` -- as many as its read-ahead budget allows -- before the device context even exists)."""

    def __init__(self, pipe: "_TypingPipeline", chunks) -> None:
        from collections import deque

        self.pipe = pipe
        self.it = iter(chunks)
        self._prev = None
        self.reading: deque = deque(pipe._early)  # (k, future or list of futures)
        pipe._early = []
        for _ in self.reading:
            next(self.it)  # (those chunks are already being read)
        while len(self.reading) < pipe.PREFETCH + 1 and self._start_read():
            pass

    def _start_read(self) -> bool:
        try:
            k, paths = next(self.it)
        except StopIteration:
            return False
        self.pipe._unread -= 1
        self.reading.append((k, self.pipe.submit_read(paths)))
        return True

    def ready(self) -> bool:
        """Whether ``next()`` would return without waiting for a read."""
        if not self.reading:
            return True  # (nothing left: StopIteration comes at once)
        futures = self.reading[0][1]
        return all(f.done() for f in futures) if isinstance(futures, list) else futures.done()

    def __iter__(self):
        return self

    def __next__(self):
        if not self.reading:
            raise StopIteration
        k, futures = self.reading.popleft()
        genomes = [f.result() for f in futures] if isinstance(futures, list) else futures.result()
        self._start_read()
        pipe = self.pipe
        ids = genomes[1] if isinstance(genomes, tuple) else [g.id for g in genomes]
        pipe.marks.setdefault("first_chunk_parsed", time.perf_counter())
        batch = pipe._make_batch(genomes)
        pipe.marks.setdefault("first_batch_created", time.perf_counter())
        pipe._order.append(k)
        # the batch before this one has had its upload enqueued for at least a chunk's parse: its page-locked words go back to
        # the pool now, not when its rows are written (a run then page-locks 3 GB instead of 5: a second less to lock at the
        # start and to unlock at exit)
        prev, self._prev = self._prev, batch
        if prev is not None:
            pipe.release_pin(prev)
        return batch, ids, genomes if pipe.objects else None


def _read_ahead_bytes() -> int:
    """How much packed input the readers may hold before the device has taken any (KAPTIVE_AMD_READ_AHEAD_GB; default: a
    quarter of what the machine / the cgroup has available, at most 12 GB)."""
    if v := os.environ.get("KAPTIVE_AMD_READ_AHEAD_GB"):
        return int(float(v) * 2**30)
    avail = 64 * 2**30
    try:
        with open("/proc/meminfo") as f:
            for line in f:
                if line.startswith("MemAvailable:"):
                    avail = int(line.split()[1]) * 1024
                    break
        for name in ("/sys/fs/cgroup/memory.max", "/sys/fs/cgroup/memory/memory.limit_in_bytes"):
            if os.path.exists(name):
                with open(name) as f:
                    text = f.read().strip()
                if text.isdigit():
                    avail = min(avail, int(text))
                break
    except (OSError, ValueError):
        pass
    return min(12 * 2**30, avail // 4)


def _device_worker(args: argparse.Namespace, device: int, chunks: list, conn, devices=None, nodes=None) -> None:
    """One process per device (``--devices a,b,...``): types its chunks and sends ``(k, outputs)`` up the pipe; an
    exception travels the same way and ends the worker.  Before it reads a file the process moves to its device's share of
    the granted CPUs on the device's NUMA node (kaptive_amd/affinity.py): reader threads and page-locked shards follow."""
    import kaptive_amd
    from kaptive_amd import affinity

    kaptive_amd.tune_runtime()
    placed = affinity.place(device, devices, nodes)
    if os.environ.get("KAPTIVE_AMD_CLI_TIMING"):
        print(f"[kaptive_amd] device {device}: NUMA node {placed['numa_node']}, {len(placed['cpus'] or [])} CPUs, pinned={placed['applied']}", file=sys.stderr)
    pipe = None
    try:
        pipe = _TypingPipeline(args, device, chunks=chunks)
        for k, out in pipe.run(chunks):
            conn.send((k, out))
        conn.send(None)
    except BaseException as e:  # noqa: BLE001 - handed to the parent, which re-raises it
        try:
            conn.send(e)
        except Exception:  # noqa: BLE001 - an exception that does not pickle travels as its text
            conn.send(RuntimeError(f"{type(e).__name__}: {e}"))
    finally:
        if pipe is not None:
            pipe.close(fast=True)  # (a worker process: it ends here)
        conn.close()


def _since_process_start() -> float:
    """Seconds since the kernel created this process (Linux: /proc; elsewhere 0): interpreter start-up and imports included."""
    try:
        with open("/proc/self/stat") as f:
            start_ticks = float(f.read().rsplit(")", 1)[1].split()[19])
        with open("/proc/uptime") as f:
            return float(f.read().split()[0]) - start_ticks / os.sysconf("SC_CLK_TCK")
    except (OSError, ValueError, IndexError):
        return 0.0


def run_type(args: argparse.Namespace) -> int:
    entered = _since_process_start()
    if str(args.devices).strip().lower() == "all":
        from kaptive_amd import _native

        devices = list(range(max(1, _native.device_count())))
    else:
        devices = [int(d) for d in str(args.devices).split(",") if d != ""] or [0]
    # Chunks: of --batch-size when given; otherwise small ones first (the first rows are out after 0.6 s instead of 1.5 s:
    # page-locking and device buffers of a 512-assembly chunk take a second to set up) and 512 from the fourth chunk of a
    # device on, where the steady rate is highest.
    sizes = iter(()) if args.batch_size else iter([64] * len(devices) + [128] * len(devices) + [256] * len(devices))
    chunks, i = [], 0
    while i < len(args.genomes):
        n = next(sizes, args.batch_size or 512)
        chunks.append((len(chunks), args.genomes[i : i + n]))
        i += n
    t_check = time.perf_counter()
    seen: set = set()
    for p in args.genomes:  # the reference fails before it types anything (src/kaptive/cli.py:287-289)
        if p not in seen:  # (a path listed twice is looked at once; os.stat directly: Path objects cost more than the call)
            seen.add(p)
            try:
                import stat as _stat

                if not _stat.S_ISREG(os.stat(p).st_mode):
                    raise FileNotFoundError(f"{p} does not exist")
            except OSError:
                raise FileNotFoundError(f"{p} does not exist") from None
    t_check = time.perf_counter() - t_check
    from kaptive_amd.serotyping.io import KaptiveRow, Pha4geRow

    handles = {}

    def stream(path):
        return sys.stdout.buffer if str(path) in ("-", "stdout") else open(path, "wb")

    if tsv := getattr(args, "out", None):
        handles["tsv"] = stream(tsv)
        handles["tsv"].write(KaptiveRow.header())
    if p := getattr(args, "pha4ge", None):
        handles["pha4ge"] = stream(p)
        handles["pha4ge"].write(Pha4geRow.header())
    if j := getattr(args, "json", None):
        handles["json"] = stream(j)
    done = 0
    timing_path = os.environ.get("KAPTIVE_AMD_CLI_TIMING")  # bench.py: when each chunk's rows were written
    t_start, chunk_times, phases = time.perf_counter(), [], {}

    def write(out, n):
        nonlocal done
        for key, blob in out.items():
            handles[key].write(blob)
        done += n
        chunk_times.append((done, time.perf_counter() - t_start))
        if args.verbose:
            print(f"\r{done}/{len(args.genomes)}", end="", file=sys.stderr, flush=True)

    try:
        if len(devices) == 1:
            pipe = _TypingPipeline(args, devices[0], chunks=chunks)
            try:
                for k, out in pipe.run(chunks):
                    write(out, len(chunks[k][1]))
            finally:
                phases = {name: round(t - t_start, 3) for name, t in pipe.marks.items()}
                pipe.close(fast=FAST_EXIT)
                if FAST_EXIT:
                    _KEEP.append(pipe)  # (its buffers and its context leave with the process, not through their destructors)
        else:
            # chunk k goes to device k mod n; rows come back through pipes and are written in input order
            import multiprocessing as mp

            ctx = mp.get_context("spawn")
            conns, procs = [], []
            from kaptive_amd import usable_cpus

            # the device processes share the host: each gets its part of the reader-thread budget
            args.threads = max(1, (args.threads or usable_cpus()) // len(devices))
            args.read_ahead_share = len(devices)  # ... and of the read-ahead memory
            from kaptive_amd import affinity

            nodes = affinity.device_numa_nodes(list(devices))  # asked once, in a process of its own (0.3 s beside the first reads)
            for i, d in enumerate(devices):
                parent, child = ctx.Pipe(duplex=False)
                proc = ctx.Process(target=_device_worker, args=(args, d, chunks[i :: len(devices)], child, list(devices), nodes), daemon=True)
                proc.start()
                child.close()
                conns.append(parent)
                procs.append(proc)
            ok = False
            try:
                for k in range(len(chunks)):
                    dev = devices[k % len(devices)]
                    try:
                        msg = conns[k % len(devices)].recv()
                    except (EOFError, OSError) as e:  # the worker died without a word (killed, crashed in native code)
                        raise RuntimeError(f"the worker of device {dev} ended unexpectedly (exit code {procs[k % len(devices)].exitcode})") from e
                    if isinstance(msg, BaseException):
                        raise msg
                    if msg is None or msg[0] != k:
                        raise RuntimeError(f"device worker {dev} ended early or out of order")
                    write(msg[1], len(chunks[k][1]))
                ok = True
            finally:
                if not ok:  # the others are blocked on full pipes: nothing to wait for
                    for proc in procs:
                        if proc.is_alive():
                            proc.terminate()
                for c in conns:
                    c.close()
                for proc in procs:
                    proc.join(timeout=30 if ok else 5)
                    if proc.is_alive():
                        proc.terminate()
    finally:
        for h in handles.values():
            if h is not sys.stdout.buffer:
                h.close()
            else:
                h.flush()
    if FAST_EXIT:
        # every stream this command writes is closed or flushed: what is left (reader threads, page-locked shards, the device
        # context) may leave with the process.  Other subcommands, and a run that raised, end through the interpreter.
        global FAST_EXIT_ARMED
        FAST_EXIT_ARMED = True
    if args.verbose:
        print(file=sys.stderr)
    if timing_path:
        # process_s: seconds since the process was created at three points -- run_type entered (interpreter start, imports,
        # argument parsing), typing started (t_start: the origin of rows_written_at / phases_s / seconds), and now
        Path(timing_path).write_text(json.dumps({"assemblies": len(args.genomes), "seconds": time.perf_counter() - t_start,
                                                 "batch_size": args.batch_size, "devices": devices,
                                                 "rows_written_at": chunk_times, "phases_s": phases,
                                                 "process_s": {"run_type_entered": round(entered, 3), "file_check": round(t_check, 3),
                                                               "end_of_run_type": round(_since_process_start(), 3)}}) + "\n")  # fmt: skip
    return 0


def run_convert(args: argparse.Namespace) -> int:
    from kaptive_amd.serotyping.models import SerotypingResult

    exporter = ResultExporter(args)
    handle = sys.stdin.buffer if args.jsonl in ("-", "stdin") else open(args.jsonl, "rb")
    try:
        for line in handle:
            line = line.strip()
            if line:
                exporter(SerotypingResult.from_dict(json.loads(line)))
    finally:
        exporter.close()
        if handle is not sys.stdin.buffer:
            handle.close()
    return 0


def main(argv=None) -> int:
    import kaptive_amd

    kaptive_amd.tune_runtime()  # before the first HIP call of the process
    from kaptive_amd.db.models import DatabaseError

    args = build_parser().parse_args(argv)
    try:
        return args.func(args)
    except (DatabaseError, FileNotFoundError, PermissionError, NotImplementedError, ValueError) as e:  # (ValueError: an unreadable FASTA / compressed stream)
        print(f"error: {e}", file=sys.stderr)
        return 1
    except KeyboardInterrupt:
        return 1
    except BrokenPipeError:
        return 130


if __name__ == "__main__":
    sys.exit(main())
