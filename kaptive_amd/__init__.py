"""kaptive_amd: MI355X-native locus typing behind Kaptive's Serotyper / Database / SerotypingResult API.

Only the `kaptive assembly` hot path lives here: contig-vs-locus-gene alignment and the
per-locus reduction run as HIP kernels (kaptive_amd/csrc) behind a C-ABI (include/kaptive_amd.h); this package is
the Python host side that mirrors the reference's operator interface for that path.
"""

import os as _os

# A context drives about a dozen HIP streams (a copy stream, three alignment passes, the reductions of every typing group);
# the runtime multiplexes them onto 4 hardware queues by default, and a queue whose head waits -- a 20 ms upload, a long
# kernel -- then holds up unrelated streams behind it (measured: 26 k instead of 32 k assemblies/s when shards stream in
# from the host).  Takes effect when set before the HIP runtime initialises; an explicit setting wins.
_os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")

__version__ = "0.1.0"
# Version string written into TSV/JSON rows where the reference writes kaptive.__version__
# (reference: src/kaptive/serotyping/core.py:463).
KAPTIVE_COMPAT_VERSION = "3.3.2"
