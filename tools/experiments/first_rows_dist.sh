# distribution of the first-rows latency and the whole-command time of a 960-file run, for several reader-thread settings
python - "$@" <<'PY'
import json, os, subprocess, sys, time, tempfile, shutil
from multiprocessing import Pool
from pathlib import Path
sys.path.insert(0, ".")
from tools.cli_probe import one
from kaptive_amd.synth import make_db
root = Path(tempfile.mkdtemp(prefix="kp_fr_", dir="/dev/shm"))
variants = [("default", {})] + [(a, dict(b.split("=") for b in v.split(","))) for a, v in (x.split(":") for x in sys.argv[1:])]
try:
    db = make_db("kpsc_k", seed=100).save(root / "db.npz")
    with Pool(16) as pool:
        pool.map(one, [(i, str(root), False) for i in range(192)])
    paths = sorted(str(p) for p in root.glob("asm*.fasta"))
    time.sleep(1.0)
    res = {tag: [] for tag, _ in variants}
    for k in range(10):
        for tag, extra in variants:
            timing = root / "timing.json"
            env = dict(os.environ, KAPTIVE_AMD_CLI_TIMING=str(timing), PYTHONPATH=".", **extra)
            t0 = time.perf_counter()
            subprocess.run([sys.executable, "-m", "kaptive_amd", "assembly", str(db), *(paths * 5), "-o", str(root / "out.tsv")], env=env, capture_output=True, text=True)
            wall = time.perf_counter() - t0
            tm = json.loads(timing.read_text())
            res[tag].append((round(wall, 2), round(tm["rows_written_at"][0][1], 2), round(tm["phases_s"]["context_ready"], 2)))
    for tag, rows in res.items():
        print(f"{tag:10s} wall {sorted(r[0] for r in rows)}  first rows {sorted(r[1] for r in rows)}  context {sorted(r[2] for r in rows)}")
finally:
    shutil.rmtree(root, ignore_errors=True)
PY
