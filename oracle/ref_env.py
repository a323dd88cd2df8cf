"""Makes the reference importable in THIS container (Python 3.10, no numba/rammappy/gb_io) -- SURVEY.md Appendix C.

Used only by oracle/make_golden.py.  /root/reference does not exist on the GPU box; nothing under tests/, bench.py or
__graft_entry__.py imports this module.
"""

import os
import sys
import tempfile
import typing
from pathlib import Path

REFERENCE = Path("/root/reference")


def activate() -> None:
    if not REFERENCE.is_dir():
        raise RuntimeError("the reference checkout is only present in the build container")
    os.environ["PYTHONDONTWRITEBYTECODE"] = "1"
    sys.dont_write_bytecode = True
    os.environ.setdefault("KAPTIVE_DB_DIR", tempfile.mkdtemp(prefix="kaptive_db_"))
    import tomli
    import typing_extensions

    if not hasattr(typing, "Self"):
        typing.Self = typing_extensions.Self
    sys.modules.setdefault("tomllib", tomli)
    shim = str(Path(__file__).resolve().parent / "refshim")
    for p in (str(REFERENCE / "src"), shim):
        if p not in sys.path:
            sys.path.insert(0, p)
