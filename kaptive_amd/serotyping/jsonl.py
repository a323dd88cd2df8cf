"""JSON lines the way the reference writes them.

The reference serialises ``SerotypingResult.to_dict()`` with ``orjson.dumps(..., OPT_SERIALIZE_NUMPY | OPT_APPEND_NEWLINE)``
(src/kaptive/serotyping/cli.py:67-76).  The orjson wheel is not in the build image, so its bytes cannot be produced here;
this module restates its published output conventions instead, so that a line written here is the line orjson writes
for the same dictionary:

* no whitespace between tokens, keys in insertion order, UTF-8 text as it is (only ``"``, ``\\`` and control characters
  escaped);
* numpy arrays and scalars natively, each in its own precision: a float32 is written with the shortest digits that read
  back as that float32 (0.1, not 0.10000000149011612), integers and booleans as JSON integers and true / false;
* floating-point layout of the Ryu printer orjson uses (ryu::raw::format64 / format32): plain decimals while the decimal
  point lies within 16 (float32: 13) digits to the right or 5 (6) zeros to the left of the digits -- always with a
  fractional part, ``100.0`` --, otherwise ``1e16``, ``1.5e-7`` (no ``+``, no padded exponent);
* NaN and the infinities as ``null``; enum members by their value.
"""

from __future__ import annotations

import enum
import json
import math
from decimal import Decimal
from typing import Any

import numpy as np

_str = json.JSONEncoder(ensure_ascii=False).encode  # JSON string escapes, non-ASCII left alone


def _ryu_layout(shortest: str, point_max: int, zeros_max: int) -> str:
    """Digits of a shortest round-trip representation (any notation) laid out as Ryu's ``format`` does."""
    sign, digits, k = Decimal(shortest).as_tuple()
    while len(digits) > 1 and digits[-1] == 0:
        digits, k = digits[:-1], k + 1
    ds = "".join(map(str, digits))
    kk = len(ds) + k  # position of the decimal point, counted from the first digit
    if 0 <= k and kk <= point_max:
        body = ds + "0" * k + ".0"
    elif 0 < kk <= point_max:
        body = ds[:kk] + "." + ds[kk:]
    elif -zeros_max < kk <= 0:
        body = "0." + "0" * (-kk) + ds
    elif len(ds) == 1:
        body = f"{ds}e{kk - 1}"
    else:
        body = f"{ds[0]}.{ds[1:]}e{kk - 1}"
    return ("-" if sign else "") + body


def format_f64(x: float) -> str:
    if x != x or x in (math.inf, -math.inf):
        return "null"
    if x == 0.0:
        return "-0.0" if math.copysign(1.0, x) < 0 else "0.0"
    return _ryu_layout(repr(float(x)), 16, 5)


def format_f32(x) -> str:
    x = np.float32(x)
    if not np.isfinite(x):
        return "null"
    if x == 0.0:
        return "-0.0" if np.signbit(x) else "0.0"
    return _ryu_layout(np.format_float_scientific(x, unique=True, trim="-"), 13, 6)


def _array(a: np.ndarray) -> str:
    if a.ndim > 1:
        return "[" + ",".join(_array(row) for row in a) + "]"
    kind = a.dtype.kind
    if kind == "b":
        return "[" + ",".join("true" if v else "false" for v in a.tolist()) + "]"
    if kind in "iu":
        return "[" + ",".join(map(str, a.tolist())) + "]"
    if kind == "f":
        fmt = format_f64 if a.dtype.itemsize == 8 else format_f32
        return "[" + ",".join(fmt(v) for v in (a.tolist() if a.dtype.itemsize == 8 else a.astype(np.float32))) + "]"
    if kind in "SU":
        return "[" + ",".join(_str(v.decode("utf-8", "replace") if isinstance(v, bytes) else v) for v in a.tolist()) + "]"
    raise TypeError(f"array of dtype {a.dtype} is not JSON serialisable")


def _value(o: Any) -> str:
    if o is None:
        return "null"
    if o is True or o is False or isinstance(o, np.bool_):
        return "true" if o else "false"
    if isinstance(o, str):
        return _str(o)
    if isinstance(o, enum.Enum):
        return _value(o.value)
    if isinstance(o, (int, np.integer)):
        return str(int(o))
    if isinstance(o, np.floating):
        return format_f64(float(o)) if o.dtype.itemsize == 8 else format_f32(o)
    if isinstance(o, float):
        return format_f64(o)
    if isinstance(o, np.ndarray):
        return _array(o)
    if isinstance(o, dict):
        return "{" + ",".join(f"{_str(str(k))}:{_value(v)}" for k, v in o.items()) + "}"
    if isinstance(o, (list, tuple)):
        return "[" + ",".join(_value(v) for v in o) + "]"
    if isinstance(o, (bytes, np.bytes_)):
        return _str(bytes(o).decode("utf-8", "replace"))
    raise TypeError(f"{type(o).__name__} is not JSON serialisable")


def dumps_line(obj: Any) -> bytes:
    """``orjson.dumps(obj, option=OPT_SERIALIZE_NUMPY | OPT_APPEND_NEWLINE)`` restated."""
    return (_value(obj) + "\n").encode("utf-8")
