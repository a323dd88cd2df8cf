#!/usr/bin/env python3
"""Where the whole-command time of `kaptive assembly ... -j` goes (GPU box): phases, when each chunk's lines were written, exit."""
import json, os, shutil, subprocess, sys, tempfile, time
from multiprocessing import Pool
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT))
from tools.cli_json_probe import one
from kaptive_amd.synth import make_db
root = Path(tempfile.mkdtemp(prefix="kp_jt_", dir="/dev/shm"))
try:
    db = make_db("kpsc_k", seed=100).save(root / "db.npz")
    with Pool(16) as pool: pool.map(one, [(i, str(root)) for i in range(96)])
    paths = sorted(str(p) for p in root.glob("asm*.fasta")) * 48
    env = dict(os.environ, PYTHONPATH=str(ROOT), KAPTIVE_AMD_CLI_TIMING=str(root / "t.json"))
    for extra in (["-j", str(root / "o.jsonl")], [], ["-j", str(root / "o.jsonl")]):
        t = time.perf_counter()
        r = subprocess.run([sys.executable, "-m", "kaptive_amd", "assembly", str(db), *paths, "-o", str(root / "o.tsv"), *extra], env=env, capture_output=True, text=True)
        wall = time.perf_counter() - t
        tm = json.loads((root / "t.json").read_text())
        m = tm["rows_written_at"]
        print(extra[:1], "wall", round(wall, 2), "phases", tm["phases_s"], "process", tm["process_s"], "exit", round(wall - tm["process_s"]["end_of_run_type"], 2),
              "marks", [(n, round(t, 2)) for n, t in m[:6]], "...", [(n, round(t, 2)) for n, t in m[-2:]], flush=True)
finally:
    shutil.rmtree(root, ignore_errors=True)
