"""CPU-side checks of the C-ABI library: it loads, exports every symbol include/kaptive_amd.h declares, and refuses
to run without a GPU instead of falling back to anything."""

import re
from pathlib import Path

import pytest

from kaptive_amd import _native

ROOT = Path(__file__).resolve().parent.parent


def declared_symbols() -> list[str]:
    text = (ROOT / "include" / "kaptive_amd.h").read_text()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(kp_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    lib = _native.lib()
    names = declared_symbols()
    assert len(names) >= 15
    for name in names:
        assert hasattr(lib, name), f"{name} declared in include/kaptive_amd.h but not exported"
    assert sorted(_native.EXPORTS) == names


def test_hit_record_layout_matches_header():
    assert _native.HIT_DTYPE.itemsize == 40
    assert _native.HIT_DTYPE.fields["strand"][1] == 36 and _native.HIT_DTYPE.fields["mapq"][1] == 37


def test_binding_constants_match_header():
    text = (ROOT / "include" / "kaptive_amd.h").read_text()
    assert int(re.search(r"#define\s+KP_WORK_SLOTS\s+(\d+)", text).group(1)) == _native.WORK_SLOTS
    spec = (ROOT / "include" / "kp_spec.h").read_text()
    assert int(re.search(r"#define\s+KP_MAX_GENE_LEN\s+(\d+)", spec).group(1)) == _native.MAX_GENE_LEN
    assert int(re.search(r"#define\s+KP_FILL16_MAX_GENE_LEN\s+(\d+)", spec).group(1)) == _native.FILL16_MAX_GENE_LEN


def test_no_gpu_means_loud_failure():
    import torch

    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(_native.NativeError):
        _native.Context(0)
    from kaptive_amd.serotyping.core import Serotyper
    from tests.golden_util import load_case, load_db

    key, genome, *_ = load_case("k_plain1")
    with pytest.raises(_native.NativeError):
        Serotyper(load_db(key))(genome)


def test_host_reserve_needs_no_device():
    """kp_host_reserve hands out 2 MB-aligned anonymous memory without touching the device runtime (the command line's
    readers fill such blocks before a context exists); kp_host_free takes a block that was never locked."""
    import numpy as np

    from kaptive_amd import _native

    pb = _native.PinnedBuffer(1 << 20, np.uint32, lazy=True)
    assert not pb.locked and pb.array.ctypes.data % (2 << 20) == 0 and len(pb.array) == 1 << 20
    pb.array[:] = np.arange(1 << 20, dtype=np.uint32)
    assert int(pb.array[-1]) == (1 << 20) - 1
    assert _native.pinned_bytes() == 0  # nothing is page-locked yet
    pb.close()
    pb.close()
