#!/usr/bin/env python3
"""Readable per-kernel table of a rocprofv3 --kernel-trace --stats csv: python tools/kstats.py gpurun_out/<dir>"""
import csv, glob, sys
f = sorted(glob.glob(f"{sys.argv[1]}/*/*kernel_stats.csv"))[-1]
for r in csv.DictReader(open(f)):
    n = r["Name"]; n = n[n.find("kp_"):][:30] if "kp_" in n else n[:30]
    if float(r["AverageNs"]) > 50000:
        print(f"{n:32s} calls {r['Calls']:>4} avg {float(r['AverageNs'])/1e3:9.1f} us  max {float(r['MaxNs'])/1e3:9.1f}")
