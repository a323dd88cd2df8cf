"""Experiment: what does the end of the CLI process cost, piece by piece?  Runs the pipeline in-process, then frees things one by one."""
import os, sys, time, json, subprocess, tempfile, shutil
from pathlib import Path
ROOT = Path("/root/repo"); sys.path.insert(0, str(ROOT))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    from kaptive_amd import cli
    mode = sys.argv[2]
    import argparse
    argv = sys.argv[3:]
    t0 = time.perf_counter()
    cli.FAST_EXIT = mode != "slow"
    # monkeypatch close to time steps
    orig_close = cli._TypingPipeline.close
    def timed_close(self, fast=False):
        t = time.perf_counter()
        if mode == "free_pins":
            with self._pin_lock:
                pins, self._pins = self._pins, []
            for pb in pins: pb.close()
            print(f"[child] closed {len(pins)} pooled buffers in {time.perf_counter()-t:.3f}s", file=sys.stderr)
        if mode == "free_all":
            orig_close(self, fast=False)
            print(f"[child] full close in {time.perf_counter()-t:.3f}s", file=sys.stderr)
            return
        orig_close(self, fast=fast)
    cli._TypingPipeline.close = timed_close
    rc = cli.main(argv)
    from kaptive_amd import _native
    print(f"[child] pinned bytes at exit {_native.lib().kp_host_pinned_bytes()/2**30:.2f} GB, threads {len(os.listdir('/proc/self/task'))}, "
          f"maps {sum(1 for _ in open('/proc/self/maps'))}, rss {int(open('/proc/self/statm').read().split()[1])*4096/2**30:.2f} GB", file=sys.stderr)
    sys.stderr.flush()
    Path(os.environ["EXIT_STAMP"]).write_text(str(time.time()))
    os._exit(0)
from tools.cli_probe import one
from multiprocessing import Pool
from kaptive_amd.synth import make_db
root = Path(tempfile.mkdtemp(prefix="kp_exit_", dir="/dev/shm"))
try:
    db = make_db("kpsc_k", seed=100).save(root / "db.npz")
    with Pool(16) as pool:
        pool.map(one, [(i, str(root), False) for i in range(64)])
    paths = sorted(str(p) for p in root.glob("asm*.fasta"))
    for mode in ("fast", "free_pins", "free_all", "fast"):
        stamp = root / "stamp"
        env = dict(os.environ, PYTHONPATH=str(ROOT), EXIT_STAMP=str(stamp))
        r = subprocess.run([sys.executable, __file__, "child", mode, "assembly", str(db), *(paths * 288), "-o", str(root / "out.tsv")], env=env, capture_output=True, text=True)
        t_end = time.time()
        print(mode, "exit took", round(t_end - float(stamp.read_text()), 3), "s;", r.stderr.strip().replace("\n", " | ")[-400:])
finally:
    shutil.rmtree(root, ignore_errors=True)
