#!/usr/bin/env python3
"""`kaptive assembly` over FASTA files on tmpfs (plain or gzip-compressed), for several reader-thread counts and chunk sizes:
assemblies per second, steady state (GPU box).

    python tools/cli_probe.py [--gz] [--files 64] [--repeats 32] [--threads 16,32] [--batch 512] [--devices all]
"""
import argparse
import gzip
import json
import os
import shutil
import subprocess
import sys
import tempfile
import time
from multiprocessing import Pool
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))


def one(job):
    from kaptive_amd.synth import make_assembly, make_db

    i, root, gz = job
    g = make_assembly(make_db("kpsc_k", seed=100), seed=90_000 + i, name=f"asm{i:04d}")
    data = g.contigs.to_fasta()
    Path(root, f"asm{i:04d}.fasta" + (".gz" if gz else "")).write_bytes(gzip.compress(data, 6) if gz else data)
    return len(data)


def main() -> int:
    ap = argparse.ArgumentParser()
    ap.add_argument("--files", type=int, default=64)
    ap.add_argument("--repeats", type=int, default=32)
    ap.add_argument("--gz", action="store_true")
    ap.add_argument("--threads", default="0", help="comma-separated -t values (0 = the CLI's default)")
    ap.add_argument("--batch", default="512", help="comma-separated --batch-size values (0 = the CLI's default ramp)")
    ap.add_argument("--marks", action="store_true", help="also print when each chunk's rows were written")
    ap.add_argument("--devices", default="", help="passed on to `kaptive assembly --devices` (e.g. all, or 0,1,2,3)")
    args = ap.parse_args()
    from kaptive_amd.synth import make_db

    root = Path(tempfile.mkdtemp(prefix="kp_gz_", dir="/dev/shm"))
    try:
        db = make_db("kpsc_k", seed=100).save(root / "db.npz")
        with Pool(16) as pool:
            sizes = pool.map(one, [(i, str(root), args.gz) for i in range(args.files)])
        paths = sorted(str(p) for p in root.glob("asm*.fasta*"))
        timing = root / "timing.json"
        env = dict(os.environ, KAPTIVE_AMD_CLI_TIMING=str(timing), PYTHONPATH=str(ROOT))
        for threads in [int(x) for x in args.threads.split(",")]:
            for batch in [int(x) for x in args.batch.split(",")]:
                t = time.perf_counter()
                r = subprocess.run([sys.executable, "-m", "kaptive_amd", "assembly", str(db), *(paths * args.repeats), "-o", str(root / "out.tsv"),
                                    *(["--batch-size", str(batch)] if batch else []), *(["-t", str(threads)] if threads else []),
                                    *(["--devices", args.devices] if args.devices else [])], env=env, capture_output=True, text=True)
                wall = time.perf_counter() - t
                if r.returncode != 0:
                    print(r.stderr[-600:])
                    return 1
                tm = json.loads(timing.read_text())
                marks = tm["rows_written_at"]
                (n0, t0), (n1, t1) = marks[min(4, len(marks) - 2)], marks[-1]
                print(json.dumps({"files": len(paths) * args.repeats, "devices": args.devices or "0", "gz": args.gz, "threads": threads, "batch": batch,
                                  "text_MB_per_file": round(sum(sizes) / len(sizes) / 1e6, 2),
                                  "MB_per_file_on_disk": round(sum(os.path.getsize(p) for p in paths) / len(paths) / 1e6, 2), "wall_s": round(wall, 2),
                                  "first_rows_after_s": round(marks[0][1], 2), "phases_s": tm.get("phases_s"), "seconds_typing": round(tm["seconds"], 2),
                                  **({"rows_written_at": [(n, round(t, 3)) for n, t in marks[:12]] + ["..."] + [(n, round(t, 3)) for n, t in marks[-3:]]} if args.marks else {}),
                                  "process_s": tm.get("process_s"), "exit_s": None if not tm.get("process_s") else round(wall - tm["process_s"]["end_of_run_type"], 2), "assemblies_per_s_steady": round((n1 - n0) / (t1 - t0), 1)}), flush=True)
        return 0
    finally:
        shutil.rmtree(root, ignore_errors=True)


if __name__ == "__main__":
    raise SystemExit(main())
