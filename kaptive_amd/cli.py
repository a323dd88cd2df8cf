"""Command line: ``python -m kaptive_amd type|assembly DATABASE GENOMES... -o out.tsv`` and ``convert``.

Same positionals and flags as the reference's ``kaptive type`` (alias ``assembly``) and ``kaptive convert``
(src/kaptive/serotyping/cli.py:118-267, shared output flags src/kaptive/cli.py:424-504), minus the plot output.  What
differs underneath: genomes are read and packed by a thread pool (``--threads`` is honoured; the reference parses and
ignores it), typed in batches on the GPU(s) (``--devices``, ``--batch-size``), and written in input order.
DATABASE is a ``.npz`` blob written by ``Database.save`` or a GenBank file with its ``.toml`` next to it.
"""

from __future__ import annotations

import argparse
import json
import sys
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

import numpy as np

FILE_SUFFIX = "kaptive_results"


def _json_default(o):
    if isinstance(o, np.ndarray):
        return o.tolist()
    if isinstance(o, np.generic):
        return o.item()
    if isinstance(o, (bytes, np.bytes_)):
        return o.decode("utf-8", "replace")
    if isinstance(o, tuple):
        return list(o)
    raise TypeError(type(o))


def result_to_json(result) -> bytes:
    d = result.to_dict()
    d["problems"] = int(d["problems"])
    return (json.dumps(d, default=_json_default, allow_nan=True) + "\n").encode()


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
    o.add_argument("-t", "--threads", type=int, default=0, metavar="", help="Threads for reading/packing genomes, 0 = all")
    o.add_argument("--partial-edge-tolerance", type=int, default=5, metavar="", help="Bases from contig edge to call a partial gene")
    o.add_argument("--devices", default="0", metavar="", help="Comma-separated GPU indices (default: 0)")
    o.add_argument("--batch-size", type=int, default=256, metavar="", help="Assemblies per device submission (default: 256)")
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


def run_type(args: argparse.Namespace) -> int:
    import os

    from kaptive_amd.core.genome import GenomeAssembly
    from kaptive_amd.serotyping.core import Serotyper

    db = load_database(args.database)
    devices = [int(d) for d in str(args.devices).split(",") if d != ""]
    typers = [
        Serotyper(db, max_other_genes=args.max_other_genes, min_completeness=args.min_completeness,
                  allow_below_threshold=args.below_threshold, partial_edge_tolerance=args.partial_edge_tolerance, device=d)
        for d in devices
    ]  # fmt: skip
    exporter = ResultExporter(args)
    threads = args.threads or os.cpu_count() or 1
    chunks = [args.genomes[i : i + args.batch_size] for i in range(0, len(args.genomes), args.batch_size)]

    def load(path):
        g = GenomeAssembly.from_file(path)
        g.packed()
        return g

    for t in typers:
        _ = t.engine  # contexts are created here, on the main thread, before any worker runs

    # One worker thread per device, bound to it for the whole run: a context is only ever driven by its own thread
    # (include/kaptive_amd.h: calls on one context are serialised).  Workers take the next chunk off a shared queue;
    # results are written in input order.
    import queue
    import threading

    todo: "queue.Queue" = queue.Queue()
    for job in enumerate(chunks):
        todo.put(job)
    finished: dict = {}
    cond = threading.Condition()

    def worker(typer):
        with ThreadPoolExecutor(max_workers=max(1, threads // len(devices))) as readers:
            while True:
                try:
                    k, paths = todo.get_nowait()
                except queue.Empty:
                    return
                try:
                    out = typer.type_many(list(readers.map(load, paths)))
                except BaseException as e:  # handed to the main thread, which re-raises it in input order
                    out = e
                with cond:
                    finished[k] = out
                    cond.notify_all()

    workers = [threading.Thread(target=worker, args=(t,), daemon=True) for t in typers]
    done = 0
    try:
        for w in workers:
            w.start()
        for k in range(len(chunks)):
            with cond:
                cond.wait_for(lambda: k in finished)
                results = finished.pop(k)
            if isinstance(results, BaseException):
                raise results
            for r in results:
                exporter(r)
            done += len(results)
            if args.verbose:
                print(f"\r{done}/{len(args.genomes)}", end="", file=sys.stderr, flush=True)
    finally:
        while True:  # nothing more is started after a failure
            try:
                todo.get_nowait()
            except queue.Empty:
                break
        for w in workers:
            if w.is_alive() or w.ident is not None:
                w.join()
        exporter.close()
        for t in typers:
            if t._engine is not None:
                t._engine.close()
    if args.verbose:
        print(file=sys.stderr)
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
    except (DatabaseError, FileNotFoundError, PermissionError, NotImplementedError) as e:
        print(f"error: {e}", file=sys.stderr)
        return 1
    except KeyboardInterrupt:
        return 1
    except BrokenPipeError:
        return 130


if __name__ == "__main__":
    sys.exit(main())
