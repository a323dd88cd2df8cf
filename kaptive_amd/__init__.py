"""kaptive_amd: MI355X-native locus typing behind Kaptive's Serotyper / Database / SerotypingResult API.

Only the `kaptive assembly` hot path lives here: contig-vs-locus-gene alignment and the
per-locus reduction run as HIP kernels (kaptive_amd/csrc) behind a C-ABI (include/kaptive_amd.h); this package is
the Python host side that mirrors the reference's operator interface for that path.
"""

__version__ = "0.1.0"
# Version string written into TSV/JSON rows where the reference writes kaptive.__version__
# (reference: src/kaptive/serotyping/core.py:463).
KAPTIVE_COMPAT_VERSION = "3.3.2"
