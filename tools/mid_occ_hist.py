#!/usr/bin/env python3
"""How often is minimap2's occurrence cut above its floor?  (build container; test infrastructure: imports oracle/)

minimap2 drops a query seed that occurs more than mid_occ times in the index, mid_occ = max(10, the (1 - 2e-4) quantile of
the occurrence counts of the index's distinct minimizers); kp_spec.h's KP_MID_OCC restates the floor (10), which is exact
whenever the quantile is <= 10.  This prints, for synthetic assemblies of every background the generator knows, the mid_occ
the independent model (oracle/mm2_model.c: mm_idx_cal_max_occ) derives -- i.e. how often and by how much the floor is too low.

    python tools/mid_occ_hist.py [--n 16] [--length 5e6]
"""
import argparse
import collections
import json
import sys
from multiprocessing import get_context
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))

KINDS = {
    "iid": dict(),
    "paralog (2 IS-like families x 5-30 copies, 7-copy operon, 40 diverged gene relatives)": dict(background="paralog"),
    "iid + one IS-like element x 12": dict(is_copies=(12, 1200)),
    "iid + one IS-like element x 40": dict(is_copies=(40, 1200)),
}


def one(job):
    kind, seed, length = job
    from kaptive_amd.synth import make_assembly, make_db
    from oracle import mm2

    db = make_db("kpsc_k", seed=100)
    g = make_assembly(db, seed=seed, length=length, **KINDS[kind])
    return kind, int(mm2.Mm2Index.from_contigs(g.contigs).mid_occ)


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=16)
    ap.add_argument("--length", type=float, default=5.0e6)
    ap.add_argument("--procs", type=int, default=6)
    a = ap.parse_args()
    jobs = [(k, 77_000 + i, a.length) for k in KINDS for i in range(a.n)]
    with get_context("fork").Pool(a.procs) as pool:
        rows = pool.map(one, jobs, chunksize=1)
    out = {}
    for kind in KINDS:
        vals = [v for k, v in rows if k == kind]
        out[kind] = {"assemblies": len(vals), "mid_occ_histogram": dict(sorted(collections.Counter(vals).items())),
                     "above_floor": sum(v > 10 for v in vals)}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
