python - <<'PY'
import sys
sys.path.insert(0, ".")
from multiprocessing import Pool
from tools.cli_probe import one
import os
os.makedirs("/dev/shm/rp", exist_ok=True)
with Pool(16) as pool:
    pool.map(one, [(i, "/dev/shm/rp", False) for i in range(64)])
PY
tools/microbench/read_pieces.bin /dev/shm/rp 16 1.5
tools/microbench/read_pieces.bin /dev/shm/rp 6 1.5 | head -4
rm -rf /dev/shm/rp
