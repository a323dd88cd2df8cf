"""Builds libkaptive_amd.so (HIP kernels + C ABI) for gfx950 with hipcc, in-tree.

    python -m kaptive_amd.build [--force]

hipcc cross-compiles without a GPU, so this runs in the build container; the .so is git-ignored but travels with the
tree to the GPU box.
"""

from __future__ import annotations

import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

PKG = Path(__file__).resolve().parent
CSRC = PKG / "csrc"
INCLUDE = PKG.parent / "include"
LIB = PKG / "libkaptive_amd.so"
SOURCES = ("kp_capi.hip", "kp_scan.hip", "kp_sort.hip", "kp_bsort.hip", "kp_chain.hip", "kp_join.hip", "kp_sw.hip", "kp_prot.hip", "kp_reduce.hip",
           "kp_fasta.cpp", "kp_rows.cpp", "kp_json.cpp", "kp_kmers.cpp")
ARCH = "gfx950"
FLAGS = ["-O3", "-std=c++17", "-fPIC", f"--offload-arch={ARCH}", "-fvisibility=hidden", "-Wall", "-Wno-unused-function",
         *os.environ.get("KAPTIVE_AMD_EXTRA_FLAGS", "").split()]  # (experiments: -DKP_SW_WAVES=6 and the like)


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if cand and (Path(cand).exists() or cand == "hipcc"):
            return cand
    raise RuntimeError("hipcc not found")


def _stale(target: Path, deps: list[Path]) -> bool:
    return not target.exists() or target.stat().st_mtime < max(d.stat().st_mtime for d in deps)


def build(force: bool = False, verbose: bool = False) -> Path:
    headers = [CSRC / "kp_internal.h", CSRC / "kp_sketch.h", CSRC / "kp_reduce_core.h", INCLUDE / "kaptive_amd.h", INCLUDE / "kp_spec.h", INCLUDE / "kp_mapq.h"]
    objdir = CSRC / "build"
    objdir.mkdir(exist_ok=True)
    hipcc = _hipcc()

    def compile_one(name: str) -> Path:
        src, obj = CSRC / name, objdir / (name + ".o")
        if force or _stale(obj, [src, *headers]):
            # host-only sources are compiled as plain C++ (hipcc would run a device pass over them as well)
            flags = FLAGS if name.endswith(".hip") else [*(f for f in FLAGS if not f.startswith("--offload-arch")), "-x", "c++"]
            cmd = [hipcc, *flags, "-c", str(src), "-o", str(obj)]
            if verbose:
                print(" ".join(cmd), flush=True)
            r = subprocess.run(cmd, capture_output=True, text=True)
            if r.returncode != 0:
                raise RuntimeError(f"hipcc failed on {name}:\n{r.stdout}\n{r.stderr}")
            if verbose and r.stderr.strip():
                print(r.stderr, file=sys.stderr)
        return obj

    with ThreadPoolExecutor(max_workers=min(4, len(SOURCES))) as pool:
        objs = list(pool.map(compile_one, SOURCES))
    if force or _stale(LIB, objs):
        cmd = [hipcc, "-shared", "-fPIC", f"--offload-arch={ARCH}", "-o", str(LIB), *map(str, objs), "-lz"]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
