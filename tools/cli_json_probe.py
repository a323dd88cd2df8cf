#!/usr/bin/env python3
"""`kaptive assembly ... -o out.tsv -j out.jsonl` (the per-result path: SerotypingResult objects, JSON lines) on the GPU box."""
import json, os, shutil, subprocess, sys, tempfile, time
from multiprocessing import Pool
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
def one(job):
    from kaptive_amd.synth import make_assembly, make_db
    i, root = job
    g = make_assembly(make_db("kpsc_k", seed=100), seed=90_000 + i, name=f"asm{i:04d}")
    Path(root, f"asm{i:04d}.fasta").write_bytes(g.contigs.to_fasta())
def main():
    from kaptive_amd.synth import make_db
    root = Path(tempfile.mkdtemp(prefix="kp_js_", dir="/dev/shm"))
    try:
        db = make_db("kpsc_k", seed=100).save(root / "db.npz")
        with Pool(16) as pool: pool.map(one, [(i, str(root)) for i in range(96)])
        paths = sorted(str(p) for p in root.glob("asm*.fasta")) * 48
        env = dict(os.environ, PYTHONPATH=str(ROOT), KAPTIVE_AMD_CLI_TIMING=str(root / "t.json"))
        for extra in (["-j", str(root / "o.jsonl")], ["--pha4ge", str(root / "o.pha4ge")], ["-g", str(root / "genes"), "-p", str(root / "prot"), "-l", str(root / "loci")]):
            t = time.perf_counter()
            r = subprocess.run([sys.executable, "-m", "kaptive_amd", "assembly", str(db), *paths, "-o", str(root / "o.tsv"), *extra], env=env, capture_output=True, text=True)
            wall = time.perf_counter() - t
            if r.returncode: print(r.stderr[-500:]); return 1
            marks = json.loads((root / "t.json").read_text())["rows_written_at"]
            (n0, t0), (n1, t1) = marks[min(4, len(marks) - 2)], marks[-1]
            print(json.dumps({"files": len(paths), "extra": extra[0], "wall_s": round(wall, 2), "per_s": round(len(paths) / wall, 1),
                              "per_s_steady": round((n1 - n0) / (t1 - t0), 1),
                              "jsonl_MB": round(os.path.getsize(root / "o.jsonl") / 1e6, 1) if extra[0] == "-j" else None}), flush=True)
        import cProfile, pstats
    finally:
        shutil.rmtree(root, ignore_errors=True)
if __name__ == "__main__":
    raise SystemExit(main())
