"""Row-level concordance of kp-align with an independent model of the published minimap2 pipeline.

BUILD CONTAINER ONLY (imports the reference from /root/reference through oracle/ref_env.py; test infrastructure, never
part of the product).  For every synthetic assembly:

  hits_kp  = oracle.kpo_align            the CPU statement of include/kp_spec.h, bit-equal to the HIP path (tests/test_gpu_parity.py)
  hits_mm2 = oracle.mm2 (mm2_model.c)    minimizers, chaining DP, anchor-to-anchor fill + z-drop extension, mm_set_mapq

are replayed through the REFERENCE's own Serotyper (src/kaptive/serotyping/core.py:124-486) by the mechanism of
oracle/make_golden.py and the two KaptiveRows are compared.  Output: profiles/concordance_r3.md (+ .json).

    python -m tools.concordance [--scale 1.0] [--procs 8] [--full-size 6]

The workload (about 520 assemblies at scale 1): the config-2/3/4 generators of SURVEY.md section 8d (KpSC K; K and O on one
assembly; 240-locus A. baumannii-shaped database with every locus split over >= 2 of ~1500 contigs) and a divergence
sweep on the K database (0-20 % substitutions, indel rates up to 1 %, IS insertions, tandem gene copies, a second
partial locus, N runs).  Backgrounds are shortened to 400 kb (the iid background only contributes chance seeds) except for
`--full-size` assemblies per config, which keep their 5 / 4 Mbp.
"""

from __future__ import annotations

import argparse
import json
import os
import sys
import time
from collections import Counter, defaultdict
from multiprocessing import get_context
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
FIELDS = ("Best match locus", "Best match type", "Match confidence", "Problems")

MID_INDEL_SIZES = (33, 48, 64, 100, 150, 300, 450)
END_INDEL_DISTS = (40, 70, 100)
END_INDEL_EVENTS = (("del", 45), ("ins", 61), ("del", 150), ("ins", 200))
INTERLEAVES = (("-40+50@400", (("del", 40, 200), ("ins", 50, 640))), ("+60-45@250", (("ins", 60, 200), ("del", 45, 450))),
               ("-33+36@150", (("del", 33, 200), ("ins", 36, 383))))  # (the second offset counts on the unedited gene: after a deletion of n, n more)
_STATE: dict = {}


def _dbs():
    if "dbs" not in _STATE:
        from oracle import make_golden as MG  # activates the reference environment
        from oracle import oracle as O
        from kaptive_amd.pack import pack_sequences_flat
        from kaptive_amd.synth import make_db

        dbs = {"k": make_db("kpsc_k", seed=100), "o": make_db("kpsc_o", seed=101), "ab": make_db("ab_k", seed=102)}
        _STATE["dbs"] = dbs
        _STATE["ref"] = {k: MG.to_ref_db(d) for k, d in dbs.items()}
        _STATE["typer"] = {k: MG.RefSerotyper(r) for k, r in _STATE["ref"].items()}
        _STATE["odb"] = {k: O.OracleDB(*pack_sequences_flat(d.genes)) for k, d in dbs.items()}
        _STATE["header"] = bytes(MG.RefKaptiveRow.header()).decode().rstrip("\n").split("\t")
        _STATE["MG"] = MG
    return _STATE


def cases(scale: float, full_size: int):
    """(tag, [db keys to type against], make_assembly kwargs).  Seeds are fixed per case."""
    out = []
    small = dict(length=400_000, median_contigs=10)
    n2, n3, n4 = int(120 * scale), int(100 * scale), int(100 * scale)
    for i in range(n2):
        kw = dict(seed=20_000 + i, **(dict() if i < full_size else small))
        out.append(("config2", ["k"], "k", kw, ()))
    for i in range(n3):
        kw = dict(seed=30_000 + i, **(dict() if i < full_size else small))
        out.append(("config3", ["k", "o"], "k", kw, ("o",)))
    for i in range(n4):
        big = i < full_size
        kw = dict(seed=40_000 + i, force_split=True, min_contig=200,
                  **(dict(length=4.0e6, median_contigs=1500) if big else dict(length=400_000, median_contigs=150)))  # fmt: skip
        out.append(("config4", ["ab"], "ab", kw, ()))
    # divergence sweep on the K database
    sub_rates = (0.0, 0.01, 0.03, 0.05, 0.08, 0.10, 0.12, 0.15, 0.18, 0.20)
    per = max(1, int(12 * scale))
    for si, sr in enumerate(sub_rates):
        for i in range(per):
            kw = dict(seed=50_000 + 100 * si + i, sub_rate=sr, p_is=0, p_stop=0, **small)
            out.append((f"sub{int(round(sr * 100)):02d}", ["k"], "k", kw, ()))
    for ii, ir in enumerate((0.001, 0.003, 0.01)):
        for i in range(per):
            kw = dict(seed=60_000 + 100 * ii + i, sub_rate=0.03, indel_rate=ir, p_is=0, p_stop=0, **small)
            out.append((f"indel{ir:g}", ["k"], "k", kw, ()))
    for i in range(per * 2):
        out.append(("is_insertion", ["k"], "k", dict(seed=61_000 + i, p_is=1.0, p_break=0, **small), ()))
    for i in range(per * 2):
        out.append(("tandem_copies", ["k"], "k", dict(seed=62_000 + i, tandem_gene=1 + i % 3, **small), ()))
    for i in range(per):
        out.append(("second_locus", ["k"], "k", dict(seed=63_000 + i, second_locus=(7 * i + 3) % 163, **small), ()))
    for i in range(per):
        out.append(("n_run", ["k"], "k", dict(seed=64_000 + i, n_run=40, p_is=0, **small), ()))
    for i in range(per):
        out.append(("no_locus", ["k"], "k", dict(seed=65_000 + i, locus=-1, **small), ()))
    # insertions and deletions of 33-450 bases inside genes (minimap2 chains across them, bw = 500): three genes per
    # assembly get one each -- the size itself, size + 1 (the other reading-frame class) and, of the opposite kind, size + 2
    small_ab = dict(length=400_000, median_contigs=150, force_split=True, min_contig=200)
    for di, (key, small_kw) in enumerate((("k", small), ("ab", small_ab))):
        for zi, size in enumerate(MID_INDEL_SIZES):
            for ki, kind in enumerate(("del", "ins")):
                other = "ins" if kind == "del" else "del"
                for i in range(max(1, per // 2)):
                    kw = dict(seed=70_000 + 10_000 * di + 500 * zi + 100 * ki + i, p_is=0, p_stop=0,
                              mid_indels=((size, kind), (size + 1, kind), (size + 2, other)), **small_kw)
                    out.append((f"midindel_{size}_{kind}", [key], key, kw, ()))
    # round 6 (the review's probes): events at an explicit distance from a gene's start or end -- the piece beyond the event is
    # too short to be a chain of its own --, pairs of events that nearly cancel (the pieces' query ranges interleave by diagonal),
    # storms of events anywhere in the locus, and the non-iid background
    for di, dist in enumerate(END_INDEL_DISTS):
        for wi, where in enumerate(("start", "end")):
            for ei, (kind, size) in enumerate(END_INDEL_EVENTS):
                for i in range(max(1, per // 3)):
                    off = dist if where == "start" else -dist
                    entry = ((kind, size, off),)
                    kw = dict(seed=95_000 + 1000 * di + 500 * wi + 100 * ei + i, p_is=0, p_stop=0, sub_rate=0.01,
                              placed_indels=(entry, entry, entry), **small)
                    out.append((f"endindel_{dist}_{where}_{kind}{size}", ["k"], "k", kw, ()))
    for ii, (name, entry) in enumerate(INTERLEAVES):
        for i in range(max(1, per // 2)):
            kw = dict(seed=98_000 + 100 * ii + i, p_is=0, p_stop=0, sub_rate=0.01, placed_indels=(entry, entry, entry), **small)
            out.append((f"interleave_{name}", ["k"], "k", kw, ()))
    for si, storm in enumerate(((6, 33, 300), (14, 33, 120))):
        for i in range(max(1, per)):
            kw = dict(seed=99_000 + 100 * si + i, p_is=0, p_stop=0, indel_storm=storm, **small)
            out.append((f"storm_{storm[0]}x{storm[1]}_{storm[2]}", ["k"], "k", kw, ()))
    for i in range(max(1, per)):
        kw = dict(seed=99_500 + i, background="paralog", **(dict() if i < min(2, full_size) else dict(length=1_000_000, median_contigs=25)))
        out.append(("paralog", ["k"], "k", kw, ()))
    # the occurrence cut (minimap2 -f 2e-4 with min_mid_occ 10; kp_spec.h KP_MID_OCC): a 150-300 base stretch of a gene of the
    # locus planted 12 / 20 / 40 more times -- at FULL size, where the 2e-4 quantile of the index stays at its floor of 10 -- and
    # what the floor cannot follow: 40 copies of an IS-like element lift the model's cut to ~40 (reported, not hidden)
    for ri, (seg, copies) in enumerate(((220, 12), (150, 20), (300, 40), (220, 9))):
        for i in range(max(1, per // 4)):
            out.append((f"repeat_{seg}x{copies}", ["k"], "k", dict(seed=90_000 + 100 * ri + i, p_is=0, p_stop=0, repeat_segment=(seg, copies)), ()))
    for i in range(max(1, per // 4)):
        out.append(("is_x40", ["k"], "k", dict(seed=91_000 + i, p_is=0, p_stop=0, is_copies=(40, 1200)), ()))
        out.append(("is_x40_repeat_220x12", ["k"], "k", dict(seed=91_100 + i, p_is=0, p_stop=0, is_copies=(40, 1200), repeat_segment=(220, 12)), ()))
    return out


def _row_fields(st, row: bytes) -> dict:
    vals = row.decode().rstrip("\n").split("\t")
    return dict(zip(st["header"], vals))


def _gene_table(exp: dict) -> dict:
    """gene index -> list of (contig, start, end, strand, gene start, gene end, state, inside) of the final gene hits."""
    g = defaultdict(list)
    gi = exp["gene_hits.gene_indices"]
    for i in range(len(gi)):
        g[int(gi[i])].append((int(exp["gene_hits.t_indices"][i]), int(exp["gene_hits.q_starts"][i]), int(exp["gene_hits.q_ends"][i]),
                              int(exp["gene_hits.strands"][i]), int(exp["gene_hits.t_starts"][i]), int(exp["gene_hits.t_ends"][i]),
                              int(exp["gene_states"][i]), bool(exp["gene_hits.is_inside"][i])))  # fmt: skip
    return g


def run_case(case):
    tag, type_with, gen_db, kw, also = case
    st = _dbs()
    from oracle import mm2
    from kaptive_amd.synth import make_assembly

    MG = st["MG"]
    t0 = time.time()
    genome = make_assembly(st["dbs"][gen_db], also=tuple(st["dbs"][a] for a in also), **kw)
    packed = genome.packed()
    idx = mm2.Mm2Index.from_contigs(genome.contigs)
    out = []
    for key in type_with:
        db = st["dbs"][key]
        t1 = time.time()
        hk = st["odb"][key].align(packed)
        t2 = time.time()
        hm = idx.map(db.genes)
        t3 = time.time()
        ek = MG.run_reference_typing(st["ref"][key], st["typer"][key], genome, hk)
        em = MG.run_reference_typing(st["ref"][key], st["typer"][key], genome, hm)
        t4 = time.time()
        rk, rm = bytes(ek["kaptive_row"]), bytes(em["kaptive_row"])
        fk, fm = _row_fields(st, rk), _row_fields(st, rm)
        gk, gm = _gene_table(ek), _gene_table(em)
        genes_both = sorted(set(gk) & set(gm))
        coord_same = sum(1 for g in genes_both if sorted(x[:6] for x in gk[g]) == sorted(x[:6] for x in gm[g]))
        state_same = sum(1 for g in genes_both if sorted(x[6] for x in gk[g]) == sorted(x[6] for x in gm[g]))
        max_delta = 0
        for g in genes_both:
            if len(gk[g]) == 1 and len(gm[g]) == 1 and gk[g][0][0] == gm[g][0][0]:
                a, b = gk[g][0], gm[g][0]
                max_delta = max(max_delta, abs(a[1] - b[1]), abs(a[2] - b[2]), abs(a[4] - b[4]), abs(a[5] - b[5]))
        spans_k = {(int(h["gene"]), int(h["contig"]), int(h["strand"]), int(h["q_start"]), int(h["q_end"]), int(h["t_start"]), int(h["t_end"])): h for h in hk}  # fmt: skip
        spans_m = {(int(h["gene"]), int(h["contig"]), int(h["strand"]), int(h["q_start"]), int(h["q_end"]), int(h["t_start"]), int(h["t_end"])): h for h in hm}  # fmt: skip
        both = set(spans_k) & set(spans_m)
        out.append(dict(
            tag=tag, db=key, seed=kw["seed"], genome=genome.id, n_contigs=len(genome.contigs),
            row_identical=rk == rm, fields_k={f: fk[f] for f in FIELDS}, fields_m={f: fm[f] for f in FIELDS},
            diff_columns=[c for c in st["header"] if fk[c] != fm[c]],
            row_k=rk.decode(), row_m=rm.decode(),
            n_hits_k=len(hk), n_hits_m=len(hm), n_spans_both=len(both), n_spans_k_only=len(set(spans_k) - both),
            n_spans_m_only=len(set(spans_m) - both),
            score_equal_on_shared=sum(1 for s in both if int(spans_k[s]["score"]) == int(spans_m[s]["score"])),
            mapq_equal_on_shared=sum(1 for s in both if int(spans_k[s]["mapq"]) == int(spans_m[s]["mapq"])),
            seeds_equal_on_shared=sum(1 for s in both if int(spans_k[s]["n_seeds"]) == int(spans_m[s]["n_seeds"])),
            n_final_genes_k=len(gk), n_final_genes_m=len(gm), n_genes_both=len(genes_both), coord_same=coord_same,
            state_same=state_same, max_coord_delta=max_delta,
            genes_k_only=sorted(set(gk) - set(gm)), genes_m_only=sorted(set(gm) - set(gk)),
            t_kp=t2 - t1, t_mm2=t3 - t2, t_ref=t4 - t3,
        ))  # fmt: skip
    out[0]["t_total"] = time.time() - t0
    return out


def summarise(records: list[dict]) -> str:
    by_tag: dict[str, list[dict]] = defaultdict(list)
    for r in records:
        by_tag[r["tag"] + ("/" + r["db"] if r["tag"] == "config3" or r["tag"].startswith("midindel") else "")].append(r)
    for r in records:  # the round-6 classes once more, pooled (24 + 3 + 2 classes of a few rows each)
        for prefix in ("endindel_40", "endindel_70", "endindel_100", "endindel", "interleave", "storm"):
            if r["tag"].startswith(prefix + "_"):
                by_tag["(pooled) " + prefix].append(r)
    lines = []
    lines.append("| workload | rows | byte-identical rows | locus/type/confidence/problems identical | locus+type+confidence identical | final genes identical in coordinates | in state | raw hit spans shared / kp-only / mm2-only | scores equal on shared spans | mapq equal | chain anchors equal |")
    lines.append("|---|---|---|---|---|---|---|---|---|---|---|")

    def line(name, rs):
        n = len(rs)
        ident = sum(r["row_identical"] for r in rs)
        f4 = sum(r["fields_k"] == r["fields_m"] for r in rs)
        f3 = sum(all(r["fields_k"][f] == r["fields_m"][f] for f in FIELDS[:3]) for r in rs)
        gb = sum(r["n_genes_both"] for r in rs)
        gu = sum(r["n_genes_both"] + len(r["genes_k_only"]) + len(r["genes_m_only"]) for r in rs)
        cs, ss = sum(r["coord_same"] for r in rs), sum(r["state_same"] for r in rs)
        sb, sk, sm = sum(r["n_spans_both"] for r in rs), sum(r["n_spans_k_only"] for r in rs), sum(r["n_spans_m_only"] for r in rs)
        se = sum(r["score_equal_on_shared"] for r in rs)
        me, ne = sum(r["mapq_equal_on_shared"] for r in rs), sum(r["seeds_equal_on_shared"] for r in rs)
        return (f"| {name} | {n} | {ident} ({100 * ident / n:.1f} %) | {f4} ({100 * f4 / n:.1f} %) | {f3} ({100 * f3 / n:.1f} %) | "
                f"{cs}/{gu} ({100 * cs / max(gu, 1):.2f} %) | {ss}/{gb} ({100 * ss / max(gb, 1):.2f} %) | {sb} / {sk} / {sm} | "
                f"{se}/{sb} ({100 * se / max(sb, 1):.2f} %) | {100 * me / max(sb, 1):.2f} % | {100 * ne / max(sb, 1):.2f} % |")

    for name in sorted(by_tag):
        lines.append(line(name, by_tag[name]))
    lines.append(line("**all**", records))
    return "\n".join(lines)


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--scale", type=float, default=1.0)
    ap.add_argument("--procs", type=int, default=min(8, os.cpu_count() or 1))
    ap.add_argument("--full-size", type=int, default=6)
    ap.add_argument("--out", default=str(ROOT / "profiles" / "concordance_r6"))
    ap.add_argument("--only", default="")
    a = ap.parse_args()
    cs = cases(a.scale, a.full_size)
    if a.only:
        cs = [c for c in cs if c[0].startswith(a.only)]
    # full-size cases first (they take longest)
    cs.sort(key=lambda c: -float(c[3].get("length", 5.0e6)))
    t0 = time.time()
    records: list[dict] = []
    if a.procs > 1:
        with get_context("fork").Pool(a.procs) as pool:
            for i, rs in enumerate(pool.imap_unordered(run_case, cs, chunksize=1)):
                records.extend(rs)
                if (i + 1) % 20 == 0:
                    print(f"{i + 1}/{len(cs)} assemblies, {time.time() - t0:.0f} s", file=sys.stderr, flush=True)
    else:
        for c in cs:
            records.extend(run_case(c))
    records.sort(key=lambda r: (r["tag"], r["seed"], r["db"]))
    table = summarise(records)
    n_asm = len(cs)
    diff_cols = Counter(c for r in records for c in r["diff_columns"])
    Path(a.out + ".json").write_text(json.dumps(dict(n_assemblies=n_asm, n_rows=len(records), seconds=time.time() - t0,
                                                     diff_columns=diff_cols, records=records), indent=1) + "\n")  # fmt: skip
    print(table)
    print("\ncolumns that differ (rows):", dict(diff_cols.most_common()))
    print(f"{n_asm} assemblies, {len(records)} rows, {time.time() - t0:.0f} s")


if __name__ == "__main__":
    main()
