"""GenBank + TOML -> Database.

The reference compiles its databases with the ``gb-io`` Rust wheel (src/kaptive/db/core.py:289-507).  This is a small
pure-Python reader for the subset those files use: one record per locus, a ``source`` feature whose ``/note``
qualifiers name the locus ("K locus: KL1"), its type ("K type: K1") or mark an "Extra genes: ..." record, and ``CDS``
features with ``/gene`` and ``/product``.  Locations: ``a..b``, ``complement(a..b)``, ``<``/``>`` markers and
``join(...)`` (outer bounds are used).  The naming rules and derived arrays are ``Database.from_parts``'s.
"""

from __future__ import annotations

import re
from pathlib import Path

from kaptive_amd.db.core import Database
from kaptive_amd.db.models import DatabaseError

_LOCUS = re.compile(r"locus:\s?(.*)$")
_TYPE = re.compile(r"type:\s?(.*)$")
_EXTRA = re.compile(r"Extra genes:\s?(.*)$")
_NUM = re.compile(r"\d+")
_QUAL = re.compile(r'^/([A-Za-z_]+)(?:=(.*))?$')


def _parse_features(lines: list[str]) -> list[dict]:
    feats: list[dict] = []
    cur = None
    qual = None
    for raw in lines:
        key, body = raw[5:21].strip(), raw[21:].rstrip()
        if key:  # new feature
            cur = {"kind": key, "location": body.strip(), "quals": []}
            feats.append(cur)
            qual = None
            continue
        if cur is None:
            continue
        text = body.strip()
        m = _QUAL.match(text) if text.startswith("/") else None
        if m and (qual is None or qual["closed"]):
            value = m.group(2) or ""
            quoted = value.startswith('"')
            qual = {"key": m.group(1), "value": value.strip('"') if quoted else value,
                    "closed": not quoted or (len(value) > 1 and value.endswith('"'))}  # fmt: skip
            cur["quals"].append(qual)
        elif qual is not None and not qual["closed"]:  # continuation of a quoted value
            closed = text.endswith('"')
            joiner = "" if qual["key"] == "translation" else " "
            qual["value"] += joiner + (text[:-1] if closed else text)
            qual["closed"] = closed
        elif not cur["quals"]:  # continuation of a long location
            cur["location"] += text
    return feats


def read_genbank(path: str | Path) -> list[dict]:
    """Records as dicts: name, seq (bytes, upper case), features (kind, location, quals)."""
    records = []
    with open(path, "r", encoding="utf-8", errors="replace") as fh:
        name, feat_lines, seq, mode = None, [], [], None
        for line in fh:
            if line.startswith("LOCUS"):
                name, feat_lines, seq, mode = line.split()[1], [], [], None
            elif line.startswith("FEATURES"):
                mode = "features"
            elif line.startswith("ORIGIN"):
                mode = "origin"
            elif line.startswith("//"):
                if name is not None:
                    records.append({"name": name, "seq": "".join(seq).upper().encode(), "features": _parse_features(feat_lines)})
                name, mode = None, None
            elif mode == "features":
                if line[:5].strip():  # a new top-level section ends the table
                    mode = None
                else:
                    feat_lines.append(line.rstrip("\n"))
            elif mode == "origin":
                seq.append("".join(ch for ch in line if ch.isalpha()))
    return records


def database_from_genbank(path: str | Path) -> Database:
    path = Path(path)
    if not (path.is_file() and path.stat().st_size > 0):
        raise FileNotFoundError(path)
    toml_path = path.with_suffix(".toml")
    if not toml_path.is_file():
        raise DatabaseError("Missing required TOML metadata file alongside Genbank file.")
    import tomli

    with toml_path.open("rb") as fp:
        metadata = tomli.load(fp)
    loci = []
    for rec in read_genbank(path):
        feats = rec["features"]
        notes = [q["value"] for q in feats[0]["quals"] if q["key"] == "note"] if feats else []
        if not notes:
            raise DatabaseError(f'Locus has no "note" qualifiers: {rec["name"]}')
        locus_name, serotype, extra = None, None, False
        for note in notes:
            if m := _EXTRA.search(note):
                extra, locus_name = True, m.group(1)
                break
            if not locus_name and (m := _LOCUS.search(note)):
                locus_name = m.group(1)
            if not serotype and (m := _TYPE.search(note)):
                serotype = m.group(1)
        if not locus_name:
            raise DatabaseError(f'Locus has no valid "locus" qualifiers: {rec["name"]}')
        genes = []
        for f in feats[1:]:
            if f["kind"] != "CDS":
                continue
            nums = [int(x) for x in _NUM.findall(f["location"])]
            if len(nums) < 2:
                raise DatabaseError(f"Cannot read CDS location {f['location']!r} in {rec['name']}")
            gene = next((q["value"] for q in f["quals"] if q["key"] == "gene"), "")
            product = next((q["value"] for q in f["quals"] if q["key"] == "product"), "")
            genes.append(dict(start=min(nums) - 1, end=max(nums), strand=-1 if "complement" in f["location"] else 1,
                              gene=gene, product=product))  # fmt: skip
        loci.append(dict(name=locus_name, type=serotype or "", extra=extra, seq=rec["seq"], genes=genes))
    return Database.from_parts(metadata, loci)


def write_genbank(db: Database, path: str | Path, antigen_word: str = "K") -> Path:
    """Writes a database back out as GenBank + TOML (used by the tests for the round trip; also handy for moving a
    synthetic database into the reference)."""
    path = Path(path)
    out = []
    for li, name in enumerate(db.loci.ids):
        o, n = int(db.loci.offsets[li]), int(db.loci.lengths[li])
        seq = db.loci.seqs[o : o + n].tobytes().decode().lower()
        g0, gl = int(db.locus_gene_offsets[li]), int(db.locus_gene_lengths[li])
        extra = bool(db.extra_genes[g0])
        out.append(f"LOCUS       {name.replace(' ', '_'):<16} {n} bp    DNA     linear   BCT 01-JAN-2000")
        out.append("FEATURES             Location/Qualifiers")
        out.append(f"     source          1..{n}")
        if extra:
            out.append(f'                     /note="Extra genes: {name}"')
        else:
            out.append(f'                     /note="{antigen_word} locus: {name}"')
            if db.serotypes[li]:
                out.append(f'                     /note="{antigen_word} type: {db.serotypes[li]}"')
        for g in range(g0, g0 + gl):
            s, e = int(db.gene_intervals.starts[g]) + 1, int(db.gene_intervals.ends[g])
            loc = f"{s}..{e}" if db.gene_intervals.strands[g] > 0 else f"complement({s}..{e})"
            out.append(f"     CDS             {loc}")
            out.append(f'                     /gene="{db.cluster_keys[db.gene_cluster_ids[g]]}"')
            out.append(f'                     /product="{db.description_keys[db.gene_description_ids[g]]}"')
        out.append("ORIGIN")
        for i in range(0, n, 60):
            chunk = seq[i : i + 60]
            out.append(f"{i + 1:>9} " + " ".join(chunk[j : j + 10] for j in range(0, len(chunk), 10)))
        out.append("//")
    path.write_text("\n".join(out) + "\n")
    meta = db.metadata.to_dict()
    lines = []
    for k, v in meta.items():
        if k in ("contact", "phenotype_logic", "antigenic_units"):
            continue
        lines.append(f"{k} = {json_like(v)}")
    lines.append("[contact]")
    for rule, spec in meta["phenotype_logic"].items():
        lines.append(f'[phenotype_logic."{rule}"]')
        for k, v in spec.items():
            lines.append(f"{k} = {json_like(v)}")
    path.with_suffix(".toml").write_text("\n".join(lines) + "\n")
    return path


def json_like(v) -> str:
    import json

    return json.dumps(v)
