#!/usr/bin/env python3
"""Per-kernel sums of rocprofv3 --pmc counters (csv output): python tools/pmc_summary.py DIR [substring ...]"""
import collections
import csv
import glob
import sys


def main(d, keys):
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    disp = collections.defaultdict(set)
    for f in glob.glob(f"{d}/*/*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            k = k[k.find("kp_"):][:40] if "kp_" in k else k[:40]
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
            disp[k].add(r["Dispatch_Id"])
    for k, v in sorted(agg.items()):
        if keys and not any(x in k for x in keys):
            continue
        print(f"{k:<42} dispatches {len(disp[k]):>3}  " + "  ".join(f"{c}={x:.4g}" for c, x in sorted(v.items())))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2:])
