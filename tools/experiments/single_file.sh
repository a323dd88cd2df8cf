# one genome per command, ten commands in a row: what a shell loop over a directory pays per genome
python - <<'PY'
import os, subprocess, sys, time, tempfile, shutil
from pathlib import Path
sys.path.insert(0, ".")
from tools.cli_probe import one
from kaptive_amd.synth import make_db
root = Path(tempfile.mkdtemp(prefix="kp_one_", dir="/dev/shm"))
try:
    db = make_db("kpsc_k", seed=100).save(root / "db.npz")
    for i in range(10): one((i, str(root), False))
    paths = sorted(str(p) for p in root.glob("asm*.fasta"))
    env = dict(os.environ, PYTHONPATH=".")
    walls = []
    for p in paths:
        t = time.perf_counter()
        r = subprocess.run([sys.executable, "-m", "kaptive_amd", "assembly", str(db), p, "-o", str(root / "o.tsv")], env=env, capture_output=True, text=True)
        walls.append(round(time.perf_counter() - t, 3))
        assert r.returncode == 0, r.stderr[-300:]
    print("one genome per command:", walls)
finally:
    shutil.rmtree(root, ignore_errors=True)
PY
