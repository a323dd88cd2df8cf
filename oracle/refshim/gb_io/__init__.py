"""Stand-in for the ``gb-io`` wheel (absent from the build image) with the surface the reference's
``Database.from_genbank`` reads (src/kaptive/db/core.py:320-404): ``iter(handle)`` yields records with ``name``,
``sequence`` (bytes) and ``features``; a feature has ``kind``, ``qualifiers`` (``key`` / ``value``) and ``location``
(``start``, ``end`` 0-based half-open, ``strand``).

TEST INFRASTRUCTURE.  The flat-file text is split by this repository's own reader (kaptive_amd/db/genbank.py), so what
the reference's compile step pins through this shim is everything AFTER record parsing: id and cluster rules, gene
positions, vocabularies, translations, phenotype masks.  The reader itself is pinned by the hand-written fixture whose
contents the test also states literally (tests/test_genbank_fixture.py).
"""

from __future__ import annotations

import re
import tempfile
from dataclasses import dataclass
from pathlib import Path

_NUM = re.compile(r"\d+")


@dataclass
class Qualifier:
    key: str
    value: str


@dataclass
class Location:
    start: int
    end: int
    strand: int


@dataclass
class Feature:
    kind: str
    location: Location
    qualifiers: list


@dataclass
class Record:
    name: str
    sequence: bytes
    features: list


def iter(handle):  # noqa: A001
    from kaptive_amd.db.genbank import read_genbank

    with tempfile.NamedTemporaryFile(suffix=".gbk", delete=False) as tmp:
        tmp.write(handle.read())
    try:
        records = read_genbank(tmp.name)
    finally:
        Path(tmp.name).unlink()
    for rec in records:
        feats = []
        for f in rec["features"]:
            nums = [int(x) for x in _NUM.findall(f["location"])]
            loc = Location(min(nums) - 1, max(nums), -1 if "complement" in f["location"] else 1) if len(nums) >= 2 else Location(0, 0, 1)
            feats.append(Feature(f["kind"], loc, [Qualifier(q["key"], q["value"]) for q in f["quals"]]))
        yield Record(rec["name"], rec["seq"], feats)
