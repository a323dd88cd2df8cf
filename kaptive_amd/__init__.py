"""kaptive_amd: MI355X-native locus typing behind Kaptive's Serotyper / Database / SerotypingResult API.

Only the `kaptive assembly` hot path lives here: contig-vs-locus-gene alignment and the
per-locus reduction run as HIP kernels (kaptive_amd/csrc) behind a C-ABI (include/kaptive_amd.h); this package is
the Python host side that mirrors the reference's operator interface for that path.
"""

import os as _os


def tune_runtime(hw_queues: int = 16) -> bool:
    """Ask the HIP runtime for `hw_queues` hardware queues (GPU_MAX_HW_QUEUES) unless the environment already says.

    A context drives about a dozen HIP streams (a copy stream, three alignment passes, the reductions of every typing
    group); the runtime multiplexes them onto 4 hardware queues by default, and a queue whose head waits -- a 20 ms
    upload, a long kernel -- then holds up unrelated streams behind it (measured: 26 k instead of 32 k assemblies/s when
    shards stream in from the host).  The variable is read when the HIP runtime initialises, so this is for ENTRY
    POINTS to call first thing (`kaptive_amd.cli`, `bench.py`, `__graft_entry__` do); importing the package does not
    touch the process environment.  Returns False when the native library (and with it HIP) is already loaded in this
    process, i.e. when the call came too late to matter."""
    from kaptive_amd import _native

    _os.environ.setdefault("GPU_MAX_HW_QUEUES", str(hw_queues))
    return not _native.is_loaded()


def usable_cpus() -> int:
    """CPUs this process may keep busy: the affinity mask cut down to the cgroup's CPU quota (a container that sees 256
    CPUs and is granted 16 CPUs' worth of time runs 256 reader threads no faster than 16, and slower for the switching)."""
    n = len(_os.sched_getaffinity(0)) if hasattr(_os, "sched_getaffinity") else (_os.cpu_count() or 1)
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            with open(path) as f:
                txt = f.read().split()
            if path.endswith("cpu.max"):
                quota, period = txt[0], float(txt[1])
            else:
                quota = txt[0]
                with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as f:
                    period = float(f.read())
            if quota not in ("max", "-1") and float(quota) > 0:
                n = min(n, max(1, int(float(quota) / period + 0.5)))
            break
        except (OSError, ValueError, IndexError):
            continue
    return max(1, n)


__version__ = "0.1.0"
# Version string written into TSV/JSON rows where the reference writes kaptive.__version__
# (reference: src/kaptive/serotyping/core.py:463).
KAPTIVE_COMPAT_VERSION = "3.3.2"
