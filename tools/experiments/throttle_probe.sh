# How many reader threads does the command line want inside a cgroup CPU quota?  (wall time per -t, cpu.stat around each run)
python - <<'PY'
import json, os, subprocess, sys, time, tempfile, shutil
from multiprocessing import Pool
from pathlib import Path
sys.path.insert(0, ".")
from tools.cli_probe import one
from kaptive_amd.synth import make_db
root = Path(tempfile.mkdtemp(prefix="kp_thr_", dir="/dev/shm"))
def stat():
    d = dict(l.split() for l in open("/sys/fs/cgroup/cpu.stat"))
    return int(d.get("nr_throttled", 0)), int(d.get("throttled_usec", 0))
try:
    db = make_db("kpsc_k", seed=100).save(root / "db.npz")
    with Pool(16) as pool:
        pool.map(one, [(i, str(root), False) for i in range(192)])
    paths = sorted(str(p) for p in root.glob("asm*.fasta"))
    time.sleep(1.0)
    for reps in (96, 5):
        for t in (16, 14, 13, 12, 10, 8, 16, 12):
            timing = root / "timing.json"
            env = dict(os.environ, KAPTIVE_AMD_CLI_TIMING=str(timing), PYTHONPATH=".")
            s0 = stat(); t0 = time.perf_counter()
            r = subprocess.run([sys.executable, "-m", "kaptive_amd", "assembly", str(db), *(paths * reps), "-o", str(root / "out.tsv"), "-t", str(t)], env=env, capture_output=True, text=True)
            wall = time.perf_counter() - t0; s1 = stat()
            tm = json.loads(timing.read_text())
            print(f"files {len(paths) * reps:6d}  -t {t:2d}: wall {wall:.2f} s, first rows {tm['rows_written_at'][0][1]:.2f}, context {tm['phases_s']['context_ready']:.2f}, typing {tm['seconds']:.2f}, "
                  f"throttled periods {s1[0] - s0[0]}, throttled thread-seconds {(s1[1] - s0[1]) / 1e6:.1f}", flush=True)
finally:
    shutil.rmtree(root, ignore_errors=True)
PY
