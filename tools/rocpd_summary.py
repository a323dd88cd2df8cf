#!/usr/bin/env python3
"""Summarise a rocprofv3 (ROCm 7.2, rocpd sqlite) kernel trace as the per-kernel table `--stats` prints in CSV mode.

    python tools/rocpd_summary.py gpurun_out/prof_x/x_results.db > profiles/r1_x_kernel_stats.txt
"""
import sqlite3
import sys


def main(path: str) -> None:
    con = sqlite3.connect(path)
    cols = [r[1] for r in con.execute("pragma table_info(kernels)")]
    name = "name" if "name" in cols else "kernel_name"
    rows = con.execute(
        f"select {name}, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) from kernels "
        f"group by {name} order by sum(end-start) desc"
    ).fetchall()
    total = sum(r[2] for r in rows) or 1
    print(f"# source: {path}")
    print(f"{'kernel':<90} {'calls':>7} {'total_ms':>10} {'avg_us':>10} {'min_us':>10} {'max_us':>10} {'pct':>6}")
    for n, c, tot, avg, mn, mx in rows:
        short = n if len(n) <= 88 else n[:85] + "..."
        print(f"{short:<90} {c:>7} {tot / 1e6:>10.3f} {avg / 1e3:>10.2f} {mn / 1e3:>10.2f} {mx / 1e3:>10.2f} {100 * tot / total:>6.2f}")


if __name__ == "__main__":
    main(sys.argv[1])
